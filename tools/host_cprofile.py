"""GPU box: cProfile of the drop-in autograd path (GaussianRasterizer forward + image.backward), the GPU drained after every step so that only HOST time
is measured.  python tools/host_cprofile.py [--check sync|deferred]"""
import cProfile, io, math, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, rasterizer as RZ
if "--check" in sys.argv:
    os.environ["GPSGS_CHECK"] = sys.argv[sys.argv.index("--check") + 1]
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


def step():
    for v in t.values(): v.grad = None
    m2.grad = None
    img, _ = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
    img.backward(gout)


for _ in range(30): step()
torch.cuda.synchronize()
n = 300
# (1) back to back, as bench.py times it
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print("back to back: %.1f us per step (%.0f views/s)" % ((time.perf_counter() - t0) / n * 1e6, n / (time.perf_counter() - t0)))
# (2) host only: GPU drained after every step
host = 0.0
for _ in range(n):
    a = time.perf_counter(); step(); host += time.perf_counter() - a
    torch.cuda.synchronize()
print("host time per step with an idle GPU: %.1f us" % (host / n * 1e6))
pr = cProfile.Profile()
for _ in range(n):
    pr.enable(); step(); pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
