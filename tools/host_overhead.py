"""GPU box: where the HOST time of one rasteriser step (sync check mode, through the drop-in autograd function) goes."""
import math, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import _capi, synthetic as S, rasterizer as RZ

dev = torch.device("cuda:0")
res, P = 1024, 600000
smp = S.make_stereo_sample(res, P, seed=S.SEED, render_res=res)
g = S.compact_sample(smp); cam = smp["novel_view"]
names = ("means3D", "colors", "opacities", "scales", "rotations")
t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in names}
m2 = torch.zeros_like(t["means3D"], requires_grad=True)
rs = RZ.GaussianRasterizationSettings(res, res, math.tan(float(cam["FovX"]) * .5), math.tan(float(cam["FovY"]) * .5), torch.zeros(3, device=dev), 1.0,
                                      torch.from_numpy(cam["world_view_transform"]).to(dev), torch.from_numpy(cam["full_proj_transform"]).to(dev), 3,
                                      torch.from_numpy(cam["camera_center"]).to(dev), False, False)
rast = RZ.GaussianRasterizer(rs)
gout = torch.randn(3, res, res, device=dev)
acc = {"wait": 0.0, "fwd": 0.0, "bwd": 0.0, "zero": 0.0}
real_wait = RZ._wait_notify
def timed_wait(*a):
    t0 = time.perf_counter(); real_wait(*a); acc["wait"] += time.perf_counter() - t0
RZ._wait_notify = timed_wait

def step():
    t0 = time.perf_counter()
    for v in t.values(): v.grad = None
    m2.grad = None
    t1 = time.perf_counter()
    img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
    t2 = time.perf_counter()
    img.backward(gout)
    t3 = time.perf_counter()
    acc["zero"] += t1 - t0; acc["fwd"] += t2 - t1; acc["bwd"] += t3 - t2

for _ in range(20): step()
torch.cuda.synchronize()
for mode in ("gpu_busy", "gpu_idle"):
    for k in acc: acc[k] = 0.0
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        step()
        if mode == "gpu_idle": torch.cuda.synchronize()   # every step starts on an idle GPU: fwd/bwd python time without queueing effects
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(json.dumps({"mode": mode, "us_per_step": round(tot / n * 1e6, 1), "fwd_total_us": round(acc["fwd"] / n * 1e6, 1), "of_which_notify_wait_us": round(acc["wait"] / n * 1e6, 1),
                      "bwd_call_us": round(acc["bwd"] / n * 1e6, 1), "zero_grad_us": round(acc["zero"] / n * 1e6, 1)}))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
