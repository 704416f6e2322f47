"""GPU box: where the HOST spends one step of the drop-in autograd path (GaussianRasterizer forward + image.backward) in the steady state, back to
back, as bench.py's `value` runs it.  The stretch that decides whether the GPU stays fed is `notified -> backward launched`: it must fit into what is
left of the forward on the GPU when the capacity notification arrives (scatter + sort + compositing, ~78 us at BASELINE config 2).
python tools/host_timeline.py [--steps 300] [--pin]   (--pin: dist.pin_near_gpu first, what bench.py and INTEGRATION.md do)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gps_gaussian_amd  # noqa
from gps_gaussian_amd import synthetic as S, rasterizer as RZ, _capi

n = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 300
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
now = time.perf_counter_ns
T = {}


def mark(k):
    T.setdefault(k, []).append(now())


def wrap(obj, name, k):
    f = getattr(obj, name)

    def w(*a, **kw):
        mark(k + "0")
        try:
            return f(*a, **kw)
        finally:
            mark(k + "1")
    setattr(obj, name, w)


lib = _capi.lib()
wrap(lib, "gsr_forward_ex", "flaunch")
wrap(lib, "gsr_backward_ex", "blaunch")
wrap(RZ, "_wait_notify", "wait")
wrap(RZ, "_backward_impl", "bimpl")


def step():
    for v in t.values(): v.grad = None
    m2.grad = None
    mark("step0")
    img, _ = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
    mark("fwd_ret")
    img.backward(gout)
    mark("bwd_ret")


for _ in range(50): step()
torch.cuda.synchronize()
T.clear()
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("back to back: %.1f us per step (%.0f views/s)" % (el / n * 1e6, n / el))


def seg(a, b, shift=0):
    x, y = T[a], T[b]
    m = min(len(x), len(y)) - shift
    d = sorted((y[i + shift] - x[i]) * 1e-3 for i in range(m))
    return "%6.1f (median %6.1f, p90 %6.1f)" % (sum(d) / len(d), d[len(d) // 2], d[int(len(d) * .9)])


print("step entry -> forward launch begins        ", seg("step0", "flaunch0"))
print("gsr_forward_ex (the launches)              ", seg("flaunch0", "flaunch1"))
print("launches done -> spin begins               ", seg("flaunch1", "wait0"))
print("spin for the capacity notification         ", seg("wait0", "wait1"))
print("notified -> forward returned to the caller ", seg("wait1", "fwd_ret"))
print("backward() called -> _backward_impl entered", seg("fwd_ret", "bimpl0"))
print("_backward_impl entered -> launch begins    ", seg("bimpl0", "blaunch0"))
print("gsr_backward_ex (the launches)             ", seg("blaunch0", "blaunch1"))
print("launches done -> backward() returned       ", seg("blaunch1", "bwd_ret"))
print("backward() returned -> next step entry     ", seg("bwd_ret", "step0", 1))
print("NOTIFIED -> BACKWARD LAUNCH BEGINS         ", seg("wait1", "blaunch0"))
print("BACKWARD LAUNCHED -> NEXT FORWARD LAUNCHED ", seg("blaunch1", "flaunch1", 1))
