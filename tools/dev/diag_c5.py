import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import hip_render, oracle_render
import gps_gaussian_amd
from gps_gaussian_amd import synthetic as S, rasterizer as RZ
res, n = int(sys.argv[1]), int(sys.argv[2])
g = S.make_scene(res, n)
H, W = g["H"], g["W"]
dpix = np.random.default_rng(5).standard_normal((3, H, W)).astype(np.float32)
try:
    img, radii, grads, info = hip_render(g, dpix, debug=True)
except Exception as e:
    print("debug render raised:", e)
    img, radii, grads, info = hip_render(g, dpix, debug=False)
o, oimg, oradii = oracle_render(g, "f32")
err = np.abs(img - oimg).max(0)
print("max err", err.max(), "pixels > 1e-3:", int((err > 1e-3).sum()))
st = RZ.export_state(info["ws"], g["means3D"].shape[0], W, H, info["cap"])
ranges = st["ranges"].cpu().numpy(); plist = st["point_list"].cpu().numpy().astype(np.int64)
depth = o.geom()["depth"].astype(np.float32).view(np.uint32).astype(np.int64)
bxp = st["bx"]
ys, xs = np.nonzero(err > 1e-3)
bins = sorted(set((int(y) // 8) * bxp + int(x) // 8 for y, x in zip(ys, xs)))
print("bad bins:", len(bins), bins[:10])
nbad = 0
lens = ranges[:, 1] - ranges[:, 0]
for b in range(ranges.shape[0]):
    a_, b_ = ranges[b]
    if b_ <= a_: continue
    ids = plist[a_:b_]
    key = depth[ids] * (1 << 32) + ids
    if not (np.diff(key) > 0).all():
        nbad += 1
        if nbad <= 8:
            d = np.diff(key); k = int(np.nonzero(d <= 0)[0][0])
            print("bin", b, "n", len(ids), "first disorder at", k, "ids", ids[max(0,k-2):k+3], "depth bits", depth[ids[max(0,k-2):k+3]], "dup ids:", len(ids) - len(set(ids.tolist())), "in bad-pixel bins:", b in bins)
print("bins out of order:", nbad, "of busy", int((lens > 0).sum()), "max len", int(lens.max()))
