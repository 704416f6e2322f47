"""GPU box: HOST time of one step of the reference's call shape -- GaussianRasterizer(rs)(...) + image.backward() at BASELINE config 2 -- for both host
paths of the drop-in module: the compiled one (csrc/host_shim.cpp) and the Python one (GPSGS_FAST=0).  Every step starts on an idle GPU (synchronize
between steps), so a call's wall time is host work plus, in the forward, the wait for the capacity notification, which is measured separately and
subtracted: host total = zero_grad + (forward call - notification wait) + backward call.  (<= 120 us asked; round 5's Python path
measured ~240 us with tools/host_timeline.py.)   python tools/host_time.py [--steps 200] [--pin]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, rasterizer as RZ

n = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 200
if "--pin" in sys.argv:
    from gps_gaussian_amd import dist as D
    print("pinned to", sorted(D.pin_near_gpu(0) or []))
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
gout = torch.randn(3, res, res, device=dev)
wait_py = [0.0]
real_wait = RZ._wait_notify


def timed_wait(*a):
    t0 = time.perf_counter(); real_wait(*a); wait_py[0] += time.perf_counter() - t0


RZ._wait_notify = timed_wait
now = time.perf_counter
for path in ("compiled", "python"):
    os.environ["GPSGS_FAST"] = "1" if path == "compiled" else "0"
    acc = dict(zero=0.0, fwd=0.0, wait=0.0, bwd=0.0)
    for it in range(20 + n):
        if it == 20:
            acc = dict(zero=0.0, fwd=0.0, wait=0.0, bwd=0.0)
        torch.cuda.synchronize()
        t0 = now()
        for v in t.values(): v.grad = None
        m2.grad = None
        t1 = now()
        wait_py[0] = 0.0
        img, _ = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        t2 = now()
        w = RZ._fast_module().last_header()[6] * 1e-6 if path == "compiled" else wait_py[0]
        img.backward(gout)
        t3 = now()
        acc["zero"] += t1 - t0; acc["fwd"] += t2 - t1; acc["wait"] += w; acc["bwd"] += t3 - t2
    us = {k: v / n * 1e6 for k, v in acc.items()}
    print(json.dumps({"host_path": path, "steps": n, "zero_grad_us": round(us["zero"], 1), "forward_call_us": round(us["fwd"], 1), "of_which_notification_wait_us": round(us["wait"], 1),
                      "backward_call_us": round(us["bwd"], 1), "host_total_us": round(us["zero"] + us["fwd"] - us["wait"] + us["bwd"], 1),
                      "grad_fn": img.grad_fn.name()}))
