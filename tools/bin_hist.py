"""GPU-box helper: per-bin list length statistics of the bench scene (how evenly the one-wave-per-bin work items are sized),
and how deep the backward walks each list (max n_contrib over the bin's pixels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import hip_render
from gps_gaussian_amd import synthetic as S, rasterizer as RZ

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
g = S.make_scene(res, n)
img, radii, _, info = hip_render(g, np.ones((3, g["H"], g["W"]), np.float32))
st = RZ.export_state(info["ws"], n, g["W"], g["H"], info["cap"], info.get("bin_cap", 0))
r = st["ranges"].cpu().numpy()
cnt = (r[:, 1] - r[:, 0]).astype(np.int64)
busy = cnt[cnt > 0]
print("bins", cnt.size, "busy", busy.size, "R", int(cnt.sum()), "mean(busy)", busy.mean(), "max", busy.max())
print("percentiles 10/50/90/99/99.9:", np.percentile(busy, [10, 50, 90, 99, 99.9]))
print("hist:", np.histogram(busy, bins=[1, 32, 64, 128, 256, 384, 512, 768, 1024, 1 << 20])[0])
nc = st["n_contrib"].cpu().numpy()
H, W = nc.shape
depth = nc[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).astype(np.int64)
print("backward walk depth (max n_contrib per bin): sum", depth.sum(), "= %.3f of R" % (depth.sum() / cnt.sum()), "max", depth.max())
# ideal vs. static round-robin makespan over 1024 SIMDs (in list entries)
w = np.sort(depth[depth > 0].ravel())[::-1]
print("work items", w.size, "ideal per SIMD", w.sum() / 1024, "longest item", w[0])
