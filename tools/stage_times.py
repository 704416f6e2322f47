"""GPU box: per-kernel hipEvent times of the rasteriser step (BASELINE config 2 by default) for one or both compositing kernel
families, plus the end-to-end step rate.  python tools/stage_times.py [--families mfma,valu] [--res 1024] [--gaussians 600000]"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import _capi, synthetic as S, rasterizer as RZ

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None, help="development: load this build of libgpsgs_hip.so instead (tools/build_ablations.sh: GSR_ABL_* probe variants)")
ap.add_argument("--families", default="tiles,valu")
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--render-res", type=int, default=None)
ap.add_argument("--gaussians", type=int, default=600000)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--no-color-grad", action="store_true", help="the gradient set stage 2 differentiates (the colours are input pixels): k_composite_bwd_tiles<false>")
ap.add_argument("--seed-offset", type=int, default=0, help="added to the synthetic scene's seed (bench.py's `configs` legs use 77)")
ap.add_argument("--attributes", default="trained", help="'untrained': scales at the 0.01 m clamp, opacity ~0.5 (what random network weights give: configs 3 / 4)")
a = ap.parse_args()
if a.lib:
    _capi.LIB_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda:0")
rr = a.render_res or a.res
smp = S.make_stereo_sample(a.res, a.gaussians, seed=S.SEED + a.seed_offset, render_res=rr, attributes=a.attributes)
g = S.compact_sample(smp); cam = smp["novel_view"]
names = ("means3D", "colors", "opacities", "scales", "rotations")
t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(not (a.no_color_grad and k == "colors")) for k in names}
m2 = torch.zeros_like(t["means3D"], requires_grad=True)
rs = RZ.GaussianRasterizationSettings(rr, rr, math.tan(float(cam["FovX"]) * .5), math.tan(float(cam["FovY"]) * .5), torch.zeros(3, device=dev), 1.0,
                                      torch.from_numpy(cam["world_view_transform"]).to(dev), torch.from_numpy(cam["full_proj_transform"]).to(dev), 3,
                                      torch.from_numpy(cam["camera_center"]).to(dev), False, False)
rast = RZ.GaussianRasterizer(rs)
gout = torch.randn(3, rr, rr, device=dev)


def step():
    for v in t.values(): v.grad = None
    m2.grad = None
    img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
    if not a.fwd_only:
        img.backward(gout)
    return img


ref = None
for fam in a.families.split(","):
    os.environ["GPSGS_COMPOSITE"] = fam
    for _ in range(5): step()
    torch.cuda.synchronize()
    RZ.set_stage_timing(True); _capi.timing_read()
    for _ in range(a.steps): step()
    st = _capi.timing_read(); RZ.set_stage_timing(False)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    img = step().detach(); torch.cuda.synchronize()
    gr = {k: t[k].grad.clone() for k in names if t[k].grad is not None} if not a.fwd_only else {}
    out = {"family": fam, "P": int(t["means3D"].shape[0]), "W": rr, "H": rr, "ms_per_step": round(dt * 1e3, 4), "views_per_s": round(1 / dt, 1), "R": RZ.last_stats(dev).get("last_R"),
           "stages_us": {k: round(v[0] / v[1] * 1e3, 2) for k, v in st.items() if v[1]}, "sum_us": round(sum(v[0] / v[1] for v in st.values() if v[1]) * 1e3, 1)}
    if ref is None:
        ref = (img, gr)
    else:
        out["img_maxdiff_vs_first"] = float((img - ref[0]).abs().max())
        out["grad_reldiff_vs_first"] = {k: float(((gr[k] - ref[1][k]).abs().max() / (ref[1][k].abs().max() + 1e-30))) for k in gr}
    print(json.dumps(out), flush=True)
