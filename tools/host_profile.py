"""GPU box: fine-grained host timing of the Python rasteriser path (monkeypatched timers, GPU kept idle between steps)."""
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
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
lib = _capi.lib()
class LibProxy:
    def __init__(self, l): self._l = l
    def __getattr__(self, n):
        f = getattr(self._l, n)
        def w(*a):
            t0 = time.perf_counter(); r = f(*a); T["C:" + n] = T.get("C:" + n, 0.0) + time.perf_counter() - t0; return r
        return w
_capi._lib = LibProxy(lib)
wrap(RZ, "_wait_notify", "wait")
wrap(RZ._RasterizeGaussians, "forward", "Function.forward"); RZ._RasterizeGaussians.forward = staticmethod(RZ._RasterizeGaussians.forward)
wrap(RZ._RasterizeGaussians, "backward", "Function.backward"); RZ._RasterizeGaussians.backward = staticmethod(RZ._RasterizeGaussians.backward)
def step():
    for v in t.values(): v.grad = None
    m2.grad = None
    t0 = time.perf_counter()
    img, _ = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
    t1 = time.perf_counter()
    img.backward(gout)
    t2 = time.perf_counter()
    T["call fwd"] = T.get("call fwd", 0.0) + t1 - t0; T["call bwd"] = T.get("call bwd", 0.0) + t2 - t1
for _ in range(20): step(); torch.cuda.synchronize()
T.clear(); n = 200
for _ in range(n): step(); torch.cuda.synchronize()
print(json.dumps({k: round(v / n * 1e6, 1) for k, v in sorted(T.items())}))
