"""GPU-box helper: fused pack vs the literal torch mask-gather path of lib/GaussianRender.py:15-34 (1024^2, B=2)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S
from gps_gaussian_amd.pack import pack_views

dev = torch.device("cuda:0")
B, res = 2, 1024
s = S.make_stereo_sample(res, 600000)
data = {}
for v in ("lmain", "rmain"):
    d = s[v]
    data[v] = dict(img=torch.from_numpy(d["img"]).to(dev)[None].repeat(B, 1, 1, 1), xyz=torch.from_numpy(d["xyz"]).to(dev)[None].repeat(B, 1, 1),
                   pts_valid=torch.from_numpy(d["pts_valid"]).to(dev)[None].repeat(B, 1), rot_maps=torch.from_numpy(d["rot_maps"]).to(dev)[None].repeat(B, 1, 1, 1),
                   scale_maps=torch.from_numpy(d["scale_maps"]).to(dev)[None].repeat(B, 1, 1, 1), opacity_maps=torch.from_numpy(d["opacity_maps"]).to(dev)[None].repeat(B, 1, 1, 1))

def unfused():
    outs = []
    for i in range(B):
        parts = [[], [], [], [], []]
        for v in ("lmain", "rmain"):
            d = data[v]; valid = d["pts_valid"][i]
            maps = (d["xyz"][i], d["img"][i].permute(1, 2, 0).reshape(-1, 3), d["rot_maps"][i].permute(1, 2, 0).reshape(-1, 4),
                    d["scale_maps"][i].permute(1, 2, 0).reshape(-1, 3), d["opacity_maps"][i].permute(1, 2, 0).reshape(-1, 1))
            for lst, m in zip(parts, maps): lst.append(m[valid])
        x = [torch.cat(p, 0) for p in parts]; x[1] = x[1] * 0.5 + 0.5
        outs.append(x)
    return outs

def fused():
    o = pack_views(data); return o[5].tolist()

res_ = {}
for name, fn in (("torch_mask_gather", unfused), ("fused_pack", fused)):
    for it in range(3 + 20):
        if it == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
    torch.cuda.synchronize(); res_[name + "_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
npix = B * 2 * res * res; nvalid = B * 600000
byts = npix * 61 + nvalid * 56 + npix * 4
res_["algorithmic_MB"] = round(byts / 1e6, 1); res_["fused_GBps"] = round(byts / res_["fused_pack_us"] / 1e3, 1)
print(json.dumps(res_))
