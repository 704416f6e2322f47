"""GPU (-m gpu): fused L1 + SSIM against values AND gradients produced by the reference's own
lib/loss.py (tests/golden/loss_golden.npz), the fp64 oracle, and torch's eager restatement at training size."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_matches_reference_golden_values_and_gradients():
    import torch
    from gps_gaussian_amd import loss as L
    g = np.load(os.path.join(GOLDEN, "loss_golden.npz"))
    dev = torch.device("cuda:0")
    pred = torch.from_numpy(g["pred"]).to(dev).requires_grad_(True)
    gt = torch.from_numpy(g["gt"]).to(dev)
    l1, ss = L.l1_and_ssim(pred, gt)
    assert abs(float(l1.detach()) - float(g["l1"])) <= 1e-6 and abs(float(ss.detach()) - float(g["ssim"])) <= 2e-6
    g_l1, = torch.autograd.grad(l1, pred, retain_graph=True)
    g_ss, = torch.autograd.grad(ss, pred)
    np.testing.assert_allclose(g_l1.cpu().numpy(), g["grad_l1"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(g_ss.cpu().numpy(), g["grad_ssim"], rtol=0, atol=2e-4 * np.abs(g["grad_ssim"]).max())


@pytest.mark.parametrize("shape", [(1, 3, 16, 16), (2, 3, 53, 70), (1, 1, 5, 7), (2, 3, 128, 96)])
def test_matches_fp64_oracle_on_ragged_sizes(shape):
    import torch
    from gps_gaussian_amd import loss as L
    from oracle import loss_oracle as O
    rng = np.random.default_rng(3)
    pred = rng.uniform(0, 1, shape).astype(np.float32)
    gt = np.clip(pred + 0.2 * rng.standard_normal(shape), 0, 1).astype(np.float32)
    gt[..., : shape[-1] // 3] = pred[..., : shape[-1] // 3]          # a region with |pred - gt| = 0 (sign(0) = 0 in the L1 backward)
    dev = torch.device("cuda:0")
    p = torch.from_numpy(pred).to(dev).requires_grad_(True)
    l1, ss = L.l1_and_ssim(p, torch.from_numpy(gt).to(dev))
    s_ref, g_ref = O.ssim(pred, gt, with_grad=True)
    assert abs(float(l1) - O.l1(pred, gt)) <= 1e-6 and abs(float(ss) - s_ref) <= 5e-6
    (0.8 * l1 + 0.2 * (1 - ss)).backward()
    g_l1 = np.sign(pred.astype(np.float64) - gt) / pred.size
    want = 0.8 * g_l1 - 0.2 * g_ref
    np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=0, atol=2e-4 * np.abs(want).max())


def test_training_size_against_eager_torch_and_stage2_combination():
    """[2,3,1024,1024] (a batch-2 stage-2 step renders 2048^2; 1024^2 keeps the eager reference cheap): eager torch
    restatement of lib/loss.py:62-83 on the GPU vs the fused op, values and gradient."""
    import torch
    import torch.nn.functional as F
    from gps_gaussian_amd import loss as L
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    pred = torch.rand(2, 3, 1024, 1024, device=dev, requires_grad=True)
    gt = (pred.detach() + 0.1 * torch.randn_like(pred)).clamp(0, 1)
    gw = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    gw = (gw / gw.sum()).unsqueeze(1)
    win = gw.mm(gw.t()).float()[None, None].expand(3, 1, 11, 11).contiguous().to(dev)

    def eager(a, b):
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        return 0.8 * (a - b).abs().mean() + 0.2 * (1 - m.mean())

    ref = eager(pred, gt)
    g_ref, = torch.autograd.grad(ref, pred)
    fused = L.stage2_photometric_loss(pred, gt)
    g_fused, = torch.autograd.grad(fused, pred)
    assert abs(float(fused) - float(ref)) <= 2e-6
    assert float((g_fused - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    # deterministic reduction: bit-identical on a re-run
    assert float(L.stage2_photometric_loss(pred, gt)) == float(fused)
