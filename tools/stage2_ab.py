"""GPU box: the stage-2 hot-path leg of bench.py (batch of 4 pairs: pack + 4 raster forwards + L1/SSIM + backward), A/B over the
compositing kernel family and over rendering the batch's samples concurrently (one stream each) or one after the other."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, render_api, loss as L
dev = torch.device("cuda:0")
B, res = 4, 1024
smp = S.make_stereo_sample(res, 600000, seed=S.SEED, render_res=res)
cam = smp["novel_view"]
data = {}
for v in ("lmain", "rmain"):
    d = smp[v]
    rep = lambda a, *r: torch.from_numpy(a).to(dev)[None].repeat(B, *r)
    data[v] = dict(img=rep(d["img"], 1, 1, 1), xyz=rep(d["xyz"], 1, 1).requires_grad_(True), pts_valid=rep(d["pts_valid"], 1), rot_maps=rep(d["rot_maps"], 1, 1, 1).requires_grad_(True),
                   scale_maps=rep(d["scale_maps"], 1, 1, 1).requires_grad_(True), opacity_maps=rep(d["opacity_maps"], 1, 1, 1).requires_grad_(True))
data["novel_view"] = dict(FovX=torch.tensor([float(cam["FovX"])] * B), FovY=torch.tensor([float(cam["FovY"])] * B), width=torch.tensor([res] * B), height=torch.tensor([res] * B),
                          world_view_transform=torch.from_numpy(cam["world_view_transform"])[None].repeat(B, 1, 1).pin_memory(),
                          full_proj_transform=torch.from_numpy(cam["full_proj_transform"])[None].repeat(B, 1, 1).pin_memory(), camera_center=torch.from_numpy(cam["camera_center"])[None].repeat(B, 1))
gt = torch.rand(B, 3, res, res, device=dev)
def step():
    for v in ("lmain", "rmain"):
        for k in ("xyz", "rot_maps", "scale_maps", "opacity_maps"):
            data[v][k].grad = None
    img = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"]
    L.stage2_photometric_loss(img, gt).backward()
for fam in ("tiles",):
    for mode, streams in (("batch", "0"), ("loop", "0"), ("loop", "1")):
        os.environ["GPSGS_COMPOSITE"] = fam; os.environ["GPSGS_PTS2RENDER"] = mode; os.environ["GPSGS_PTS2RENDER_STREAMS"] = streams
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        print(json.dumps({"family": fam, "pts2render": mode, "samples_concurrent": mode == "batch" or streams == "1",
                          "ms_per_iter": round((time.perf_counter() - t0) / 20 * 1e3, 3)}))
# where the time of the default form goes: the same loop under the profiler would perturb it; time the parts instead
os.environ["GPSGS_PTS2RENDER"] = "batch"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)
def fwd_only():
    with torch.no_grad():
        render_api.pts2render(data, [0, 0, 0])
img0 = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"].detach().requires_grad_(True)
def loss_only():
    img0.grad = None
    L.stage2_photometric_loss(img0, gt).backward()
print(json.dumps({"pts2render_forward_only_ms": timeit(fwd_only), "loss_fwd_bwd_ms": timeit(loss_only)}))
if os.environ.get("STAGE2_CPROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    for _ in range(3): step()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20): step()
    torch.cuda.synchronize()
    pr.disable()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(45)
    print(sio.getvalue()[:9000])
