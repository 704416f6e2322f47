import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, rasterizer as RZ, _capi
dev = torch.device("cuda:0")
g = S.make_scene(1024, 600000)
t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
m2 = torch.zeros_like(t["means3D"], requires_grad=True)
rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                      torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                      torch.from_numpy(g["campos"]).to(dev), False, False)
rast = RZ.GaussianRasterizer(rs)
gout = torch.randn(3, g["H"], g["W"], device=dev)
for dbg in [int(x) for x in (sys.argv[1:] or ["0", "1", "2", "4"])]:
    RZ._extra_flags = _capi.GSR_FLAG_TIMING | (dbg << 8)
    for it in range(8):
        img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        img.backward(gout)
        if it == 2: torch.cuda.synchronize(); _capi.timing_read()
    st = _capi.timing_read()
    print("dbg", dbg, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in st.items() if v[1]})
