"""GPU-box helper: time the fused correlation pyramid / multi-level lookup / convex upsampling against
the same steps written the reference's way in eager PyTorch on the same GPU (einsum + divide + avg_pool2d; 4 sampler calls +
cat; softmax + unfold + sum).  Sizes = the reference's stage-2 shapes for a batch of B stereo pairs at 1024^2:
fmap [2B,192,128,128], 4 levels, radius 4, 3 GRU iterations, upsampling factor 8."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import gps_gaussian_amd
from gps_gaussian_amd import corr as K

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.float32
N, D, H, W, f, iters = 2 * B, 192, 128, 128, 8, 3
torch.manual_seed(0)
f1 = torch.randn(N, D, H, W, device=dev, dtype=dt, requires_grad=True)
f2 = torch.randn(N, D, H, W, device=dev, dtype=dt, requires_grad=True)
coords = [torch.rand(N, 2, H, W, device=dev) * W for _ in range(iters)]
mask = torch.randn(N, 9 * f * f, H, W, device=dev, requires_grad=True)
flow = torch.randn(N, 2, H, W, device=dev, requires_grad=True)


def eager_block():
    corr = torch.einsum('aijk,aijh->ajkh', f1, f2).reshape(N, H, W, 1, W).contiguous() / math.sqrt(D)
    corr = corr.reshape(N * H * W, 1, 1, W)
    pyr = []
    for i in range(4):
        pyr.append(corr.view(N, H, W, -1, W // 2 ** i))
        corr = F.avg_pool2d(corr, [1, 2], stride=[1, 2])
    outs = []
    for c in coords:
        lv = [K.CorrSampler.apply(pyr[i].squeeze(3), c[:, [0]] / 2 ** i, 4) for i in range(4)]   # the row-a14 drop-in, one call per level
        outs.append(torch.cat(lv, dim=1))
    return outs


def fused_block():
    blk = K.CorrBlockFast1D(f1, f2, num_levels=4, radius=4)
    return [blk(c) for c in coords]


def eager_up():
    m = torch.softmax(mask.view(N, 1, 9, f, f, H, W), dim=2)
    up = F.unfold(f * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, f * H, f * W)


def fused_up():
    return K.upsample_flow(flow, mask, f)


def timeit(fn, bwd, n=20):
    for it in range(3 + n):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        if bwd:
            outs = out if isinstance(out, list) else [out]
            sum(o.float().sum() for o in outs).backward()
            f1.grad = f2.grad = mask.grad = flow.grad = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"B": B, "dtype": str(dt), "shape": [N, D, H, W]}
for name, a, b in (("corr_block", eager_block, fused_block), ("upsample", eager_up, fused_up)):
    for bwd in (False, True):
        k = name + ("_fwd_bwd" if bwd else "_fwd")
        res[k] = {"eager_ms": round(timeit(a, bwd), 3), "fused_ms": round(timeit(b, bwd), 3)}
# volume kernel alone: 2 * N*H * W*W*D flops
for it in range(23):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        K._BuildPyramid.apply(f1, f2, 4)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
flops = 2.0 * N * H * W * W * D
byts = (2 * N * D * H * W + N * H * W * W * 1.875) * f1.element_size()
res["cv_build_fwd"] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1), "algorithmic_GBs": round(byts / ms / 1e6, 1)}
print(json.dumps(res))
