"""GPU (-m gpu): the HIP correlation sampler against the reference-generated golden vectors and the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_forward_backward_match_reference_golden_fp32():
    import torch
    from gps_gaussian_amd import corr
    g = np.load(os.path.join(GOLDEN, "corr_sampler_golden.npz"))
    dev = torch.device("cuda:0")
    r = int(g["radius"]); rd = 2 * r + 1
    coords = torch.from_numpy(g["coords"]).to(dev)
    for lvl in range(4):
        vol = torch.from_numpy(g["volume%d" % lvl]).to(dev)
        out, = corr.forward(vol, coords / 2 ** lvl, r)
        ref = g["out"][:, lvl * rd:(lvl + 1) * rd]
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-5 * float(vol.abs().max()))
        go = torch.from_numpy(np.ascontiguousarray(g["grad_out"][:, lvl * rd:(lvl + 1) * rd])).to(dev)
        gv, = corr.backward(vol, coords / 2 ** lvl, go, r)
        np.testing.assert_allclose(gv.cpu().numpy(), g["grad_volume%d" % lvl], rtol=0, atol=2e-5 * float(go.abs().max()))


@pytest.mark.parametrize("shape", [(4, 128, 128, 128), (4, 128, 128, 16), (1, 3, 5, 7), (2, 9, 33, 64), (2, 5, 9, 12), (1, 2, 3, 4)])
@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_bit_exact_against_oracle_at_model_sizes(shape, dtype):
    """Shapes the model uses at 1024^2 (features 128x128, pyramid widths 128..16).  fp32: bit-exact vs the CPU oracle
    (same operation order, no FMA contraction); fp16: fp32 arithmetic rounded once to half."""
    import torch
    from gps_gaussian_amd import corr
    from oracle import gsr_oracle
    gsr_oracle.build()
    aux = C.CDLL(os.path.join(os.path.dirname(gsr_oracle.__file__), "_build", "libaux_oracle.so"))
    N, H1, W1, W2 = shape
    rng = np.random.default_rng(1)
    vol = rng.standard_normal(shape).astype(np.float32)
    if dtype == "float16":
        vol = vol.astype(np.float16).astype(np.float32)
    coords = (rng.uniform(-6, W2 + 6, (N, 1, H1, W1))).astype(np.float32)
    r = 4; rd = 9
    out = np.zeros((N, rd, H1, W1), np.float32)
    c0 = np.ascontiguousarray(coords[:, 0])
    aux.cs_oracle_forward(_p(vol), _p(c0), _p(out), N, H1, W1, W2, r)
    go = rng.standard_normal((N, rd, H1, W1)).astype(np.float32)
    if dtype == "float16":
        go = go.astype(np.float16).astype(np.float32)
    gv = np.zeros_like(vol)
    aux.cs_oracle_backward(_p(c0), _p(go), _p(gv), N, H1, W1, W2, r)
    dev = torch.device("cuda:0")
    td = getattr(torch, dtype)
    o, = corr.forward(torch.from_numpy(vol).to(dev, td), torch.from_numpy(coords).to(dev), r)
    g, = corr.backward(torch.from_numpy(vol).to(dev, td), torch.from_numpy(coords).to(dev), torch.from_numpy(go).to(dev, td), r)
    assert o.dtype == td and g.dtype == td
    if dtype == "float32":
        np.testing.assert_array_equal(o.cpu().numpy(), out)
        np.testing.assert_array_equal(g.cpu().numpy(), gv)
    else:
        np.testing.assert_array_equal(o.cpu().numpy(), out.astype(np.float16))
        np.testing.assert_array_equal(g.cpu().numpy(), gv.astype(np.float16))


def test_autograd_function_as_the_reference_wraps_it():
    """core/corr.py:17-29 wraps forward/backward in an autograd.Function; the same wrapper over our module must agree with
    torch's grid_sample path (CorrBlock1D semantics) including gradients."""
    import torch
    import torch.nn.functional as F
    from gps_gaussian_amd import corr as cs

    class CorrSampler(torch.autograd.Function):
        @staticmethod
        def forward(ctx, volume, coords, radius):
            ctx.save_for_backward(volume, coords)
            ctx.radius = radius
            corr, = cs.forward(volume, coords, radius)
            return corr

        @staticmethod
        def backward(ctx, grad_output):
            volume, coords = ctx.saved_tensors
            grad_volume, = cs.backward(volume, coords, grad_output.contiguous(), ctx.radius)
            return grad_volume, None, None

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, H1, W1, W2, r = 2, 6, 20, 20, 4
    vol = torch.randn(N, H1, W1, W2, device=dev, requires_grad=True)
    coords = torch.rand(N, 1, H1, W1, device=dev) * (W2 + 4) - 2
    out = CorrSampler.apply(vol, coords, r)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g1 = vol.grad.clone()
    # torch reference: grid_sample on a height-1 image, align_corners=True (core/utils/utils.py:59-74)
    vol2 = vol.detach().clone().requires_grad_(True)
    dx = torch.linspace(-r, r, 2 * r + 1, device=dev).view(1, 1, 2 * r + 1, 1)
    x0 = dx + coords.reshape(N * H1 * W1, 1, 1, 1)
    grid = torch.cat([2 * x0 / (W2 - 1) - 1, torch.zeros_like(x0)], dim=-1)
    samp = F.grid_sample(vol2.reshape(N * H1 * W1, 1, 1, W2), grid, align_corners=True)
    ref = samp.view(N, H1, W1, -1).permute(0, 3, 1, 2)
    (ref * w).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=5e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), vol2.grad.cpu().numpy(), atol=5e-5)
