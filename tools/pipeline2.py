"""GPU box: does rendering TWO independent views concurrently (two RasterSessions on two streams) raise views/s?  The compositing
kernels have only ~5 one-wave work items per SIMD (5131 busy bins / 1024 SIMDs at config 2), so a second view in flight doubles the
waves the SIMDs can interleave."""
import math, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, rasterizer as RZ
from gps_gaussian_amd.session import RasterSession
dev = torch.device("cuda:0")
res, P = 1024, 600000
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
views = []
for i in range(depth):
    smp = S.make_stereo_sample(res, P, seed=S.SEED + i, render_res=res)
    g = S.compact_sample(smp); cam = smp["novel_view"]
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    views.append(dict(t=t, opa=t["opacities"].reshape(-1), view=torch.from_numpy(cam["world_view_transform"]).to(dev), proj=torch.from_numpy(cam["full_proj_transform"]).to(dev),
                      bg=torch.zeros(3, device=dev), tx=math.tan(float(cam["FovX"]) * .5), ty=math.tan(float(cam["FovY"]) * .5), sess=RasterSession(P, res, res, dev),
                      stream=torch.cuda.Stream(), gout=torch.randn(3, res, res, device=dev)))
def step(v):
    with torch.cuda.stream(v["stream"]):
        t = v["t"]
        v["sess"].forward(t["means3D"], t["colors"], v["opa"], t["scales"], t["rotations"], v["view"], v["proj"], v["bg"], v["tx"], v["ty"])
        v["sess"].backward(v["gout"])
def begin(v):
    with torch.cuda.stream(v["stream"]):
        t = v["t"]
        v["sess"].forward_begin(t["means3D"], t["colors"], v["opa"], t["scales"], t["rotations"], v["view"], v["proj"], v["bg"], v["tx"], v["ty"])
def end(v):
    with torch.cuda.stream(v["stream"]):
        v["sess"].forward_end()
        v["sess"].backward(v["gout"])
torch.cuda.synchronize()
for n_inflight in range(1, depth + 1):
    act = views[:n_inflight]
    for _ in range(10):
        for v in act: step(v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 60
    for _ in range(N):
        for v in act: step(v)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for _ in range(10):
        for v in act: begin(v)
        for v in act: end(v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        for v in act: begin(v)
        for v in act: end(v)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(json.dumps({"views_in_flight": n_inflight, "round_robin_views_per_s": round(N * n_inflight / dt, 1), "begin_all_then_end_views_per_s": round(N * n_inflight / dt2, 1),
                      "ms_per_view": round(dt2 / (N * n_inflight) * 1e3, 4)}))
