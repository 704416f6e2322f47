"""GPU box: forward-only throughput with F views in flight (see tools/pipeline2.py)."""
import math, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, rasterizer as RZ
from gps_gaussian_amd.session import RasterSession
dev = torch.device("cuda:0")
res, P = 1024, 600000
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
views = []
for i in range(depth):
    smp = S.make_stereo_sample(res, P, seed=S.SEED + i, render_res=res)
    g = S.compact_sample(smp); cam = smp["novel_view"]
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    views.append(dict(t=t, opa=t["opacities"].reshape(-1), view=torch.from_numpy(cam["world_view_transform"]).to(dev), proj=torch.from_numpy(cam["full_proj_transform"]).to(dev),
                      bg=torch.zeros(3, device=dev), tx=math.tan(float(cam["FovX"]) * .5), ty=math.tan(float(cam["FovY"]) * .5), sess=RasterSession(P, res, res, dev, training=False),
                      stream=torch.cuda.Stream()))
def begin(v):
    with torch.cuda.stream(v["stream"]):
        t = v["t"]
        v["sess"].forward_begin(t["means3D"], t["colors"], v["opa"], t["scales"], t["rotations"], v["view"], v["proj"], v["bg"], v["tx"], v["ty"])
def end(v):
    with torch.cuda.stream(v["stream"]):
        v["sess"].forward_end()
torch.cuda.synchronize()
for n in range(1, depth + 1):
    act = views[:n]
    for _ in range(10):
        for v in act: begin(v)
        for v in act: end(v)
    torch.cuda.synchronize(); t0 = time.perf_counter(); N = 100
    for _ in range(N):
        for v in act: begin(v)
        for v in act: end(v)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"views_in_flight": n, "forward_only_views_per_s": round(N * n / dt, 1), "ms_per_view": round(dt / (N * n) * 1e3, 4)}))
