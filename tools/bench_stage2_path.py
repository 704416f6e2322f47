"""GPU-box helper: the non-network part of one stage-2 training step (train_stage2.py:65-72,83), per batch:
pack (lib/GaussianRender.py:15-34) -> render x B at 2048^2 -> 0.8*L1 + 0.2*(1-SSIM) -> backward to the per-pixel maps.
Compares the fused path of this repo with the eager pieces the reference would run around the same rasteriser."""
import os, sys, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, render_api, loss as L

dev = torch.device("cuda:0")
B, res, rres = int(os.environ.get("B", 2)), 1024, 2048
s = S.make_stereo_sample(res, 600000, render_res=rres)
cam = s["novel_view"]
def mk():
    data = {}
    for v in ("lmain", "rmain"):
        d = s[v]
        data[v] = dict(img=torch.from_numpy(d["img"]).to(dev)[None].repeat(B, 1, 1, 1), xyz=torch.from_numpy(d["xyz"]).to(dev)[None].repeat(B, 1, 1).requires_grad_(True),
                       pts_valid=torch.from_numpy(d["pts_valid"]).to(dev)[None].repeat(B, 1),
                       rot_maps=torch.from_numpy(d["rot_maps"]).to(dev)[None].repeat(B, 1, 1, 1).requires_grad_(True),
                       scale_maps=torch.from_numpy(d["scale_maps"]).to(dev)[None].repeat(B, 1, 1, 1).requires_grad_(True),
                       opacity_maps=torch.from_numpy(d["opacity_maps"]).to(dev)[None].repeat(B, 1, 1, 1).requires_grad_(True))
    data["novel_view"] = dict(FovX=torch.tensor([float(cam["FovX"])] * B), FovY=torch.tensor([float(cam["FovY"])] * B),
                              width=torch.tensor([rres] * B), height=torch.tensor([rres] * B),
                              world_view_transform=torch.from_numpy(cam["world_view_transform"])[None].repeat(B, 1, 1).pin_memory(),
                              full_proj_transform=torch.from_numpy(cam["full_proj_transform"])[None].repeat(B, 1, 1).pin_memory(),
                              camera_center=torch.from_numpy(cam["camera_center"])[None].repeat(B, 1))
    return data
data = mk()
gt = torch.rand(B, 3, rres, rres, device=dev)
gw = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32); gw = (gw / gw.sum()).unsqueeze(1)
win = gw.mm(gw.t()).float()[None, None].expand(3, 1, 11, 11).contiguous().to(dev)
def eager_loss(a, b):
    mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
    s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return 0.8 * (a - b).abs().mean() + 0.2 * (1 - m.mean())
def step(fused):
    for v in ("lmain", "rmain"):
        for k in ("xyz", "rot_maps", "scale_maps", "opacity_maps"):
            data[v][k].grad = None
    img = (render_api.pts2render if fused else render_api.pts2render_unfused)(data, [0, 0, 0])["novel_view"]["img_pred"]
    loss = L.stage2_photometric_loss(img, gt) if fused else eager_loss(img, gt)
    loss.backward()
out = {"B": B, "render": "%dx%d" % (rres, rres)}
for name, fused in (("eager_neighbours", False), ("fused_path", True)):
    for it in range(3 + 10):
        if it == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        step(fused)
    torch.cuda.synchronize(); out[name + "_ms_per_step"] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
out["speedup"] = round(out["eager_neighbours_ms_per_step"] / out["fused_path_ms_per_step"], 1)
print(json.dumps(out))
