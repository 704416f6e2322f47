"""GPU-box helper: time the raster at the BASELINE configs (C2 at 1024^2 and rendered at 2048^2, C5) and print one line each."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, rasterizer as RZ, _capi

dev = torch.device("cuda:0")
for name, res, n, rres in (("C2 1024->1024", 1024, 600000, 1024), ("C2 1024->2048 (use_hr_img)", 1024, 600000, 2048), ("C5 2048->2048", 2048, 2400000, 2048)):
    g = S.make_scene(res, n, render_res=rres)
    t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                          torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                          torch.from_numpy(g["campos"]).to(dev), False, False)
    rast = RZ.GaussianRasterizer(rs)
    gout = torch.randn(3, g["H"], g["W"], device=dev)
    out = {"config": name, "P": n}
    for mode in ("fwd", "fwd_bwd"):
        RZ.set_stage_timing(mode == "fwd_bwd")
        for it in range(5 + 20):
            if it == 5:
                torch.cuda.synchronize(); _capi.timing_read(); t0 = time.perf_counter()
            if mode == "fwd":
                with torch.no_grad():
                    rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
            else:
                img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
                img.backward(gout)
        torch.cuda.synchronize()
        out[mode + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    st = _capi.timing_read()
    RZ.set_stage_timing(False)
    out["R"] = RZ.last_stats(dev)["last_R"]
    out["stages_us"] = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in st.items() if v[1]}
    print(json.dumps(out))
