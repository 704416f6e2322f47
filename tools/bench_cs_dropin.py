"""GPU-box helper: the DROP-IN correlation sampler path exactly as the reference's unmodified scripts hit it
(/root/reference/core/corr.py:38-61: CorrBlockFast1D.__call__ = one corr_sampler call per pyramid level, 4 levels, radius 4,
per GRU iteration), through gps-gaussian_amd/dropin/corr_sampler, at the stage-2 shapes (batch of B stereo pairs at 1024^2 ->
fmap 128x128, N = 2B).  Prints per-call times; run under rocprofv3 by tools/prof_cs.sh for the kernel rows and the HBM traffic."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gps-gaussian_amd", "dropin"))
import torch
import torch.nn.functional as F
import corr_sampler  # the drop-in module: forward(volume, coords, radius) -> (out,), backward(volume, coords, grad, radius) -> (grad_volume,)

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.float32
N, H, W, iters, r = 2 * B, 128, 128, 3, 4
torch.manual_seed(0)
corr = torch.randn(N * H * W, 1, 1, W, device=dev, dtype=dt)
pyr = []
for i in range(4):
    pyr.append(corr.view(N, H, W, W // 2 ** i).contiguous())
    corr = F.avg_pool2d(corr.float(), [1, 2], stride=[1, 2]).to(dt)
coords = [torch.rand(N, 1, H, W, device=dev) * W for _ in range(iters)]
grads = [torch.randn(N, 2 * r + 1, H, W, device=dev, dtype=dt) for _ in range(4)]


def fwd():
    return [[corr_sampler.forward(pyr[i], (c / 2 ** i).contiguous(), r)[0] for i in range(4)] for c in coords]


def fwd_bwd():
    fwd()
    return [[corr_sampler.backward(pyr[i], (c / 2 ** i).contiguous(), grads[i], r)[0] for i in range(4)] for c in coords]


def timeit(fn, n=30):
    for it in range(5 + n):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"shape": "volume [%d,%d,%d,W2] W2=128,64,32,16, radius 4, %d GRU iterations, %s" % (N, H, W, iters, str(dt).split(".")[-1]),
       "ms_12_forward_calls": round(timeit(fwd), 4), "ms_12_forward_plus_12_backward_calls": round(timeit(fwd_bwd), 4)}
# device time per call of the largest level, events around 50 back-to-back launches
c0 = coords[0].contiguous()
for lvl in (0, 3):
    cc = (c0 / 2 ** lvl).contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        corr_sampler.forward(pyr[lvl], cc, r)
    e0.record()
    for _ in range(50):
        corr_sampler.forward(pyr[lvl], cc, r)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    es = pyr[lvl].element_size()
    useful = N * H * W * ((2 * r + 2) * es + 4 + (2 * r + 1) * es)  # taps read + coord + outputs written
    res["level%d_forward_us_back_to_back" % lvl] = round(us, 2)
    res["level%d_algorithmic_GBps" % lvl] = round(useful / us / 1e3, 1)
print(json.dumps(res))
