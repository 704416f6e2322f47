"""GPU-box helper: fused L1+SSIM vs the eager restatement of lib/loss.py at the stage-2 training size [2,3,2048,2048]."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import gps_gaussian_amd
from gps_gaussian_amd import loss as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
pred = torch.rand(2, 3, 2048, 2048, device=dev, requires_grad=True)
gt = (pred.detach() + 0.1 * torch.randn_like(pred)).clamp(0, 1)
gw = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32); gw = (gw / gw.sum()).unsqueeze(1)
win = gw.mm(gw.t()).float()[None, None].expand(3, 1, 11, 11).contiguous().to(dev)
def eager():
    a, b = pred, gt
    mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
    s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return 0.8 * (a - b).abs().mean() + 0.2 * (1 - m.mean())
def fused():
    return L.stage2_photometric_loss(pred, gt)
out = {}
for name, fn in (("eager_torch", eager), ("fused", fused)):
    for it in range(3 + 10):
        if it == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        pred.grad = None
        fn().backward()
    torch.cuda.synchronize(); out[name + "_fwd_bwd_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
n = pred.numel()
out["fused_algorithmic_MB"] = round(n * (8 + 12 + 20 + 4) / 1e6, 1)
out["fused_GBps"] = round(n * 44 / (out["fused_fwd_bwd_ms"] * 1e-3) / 1e9, 1)
print(json.dumps(out))
