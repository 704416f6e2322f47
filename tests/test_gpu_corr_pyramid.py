"""GPU (-m gpu): SURVEY.md section 8(f) row 4 -- correlation volume + pyramid (fp32 MFMA), fused multi-level lookup, convex
upsampling, all through the C-ABI, against the reference's golden vectors and the numpy oracle (oracle/corr_oracle.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _t(a, dtype=None, grad=False):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(grad)


def test_volume_and_pyramid_match_reference_golden_fwd_bwd():
    import torch
    from gps_gaussian_amd import corr as K
    z = np.load(os.path.join(GOLDEN, "corr_pyramid_golden.npz"))
    f1, f2 = _t(z["fmap1"], grad=True), _t(z["fmap2"], grad=True)
    blk = K.CorrBlockFast1D(f1, f2, num_levels=4, radius=4)
    for l in range(4):
        assert tuple(blk.corr_pyramid[l].shape) == z["pyr%d" % l].shape[:3] + (1, z["pyr%d" % l].shape[3])   # the reference's 5-D view
        np.testing.assert_allclose(blk.volumes[l].detach().cpu().numpy(), z["pyr%d" % l], rtol=1e-5, atol=1e-5)
    loss = sum((v * _t(z["grad_pyr%d" % l])).sum() for l, v in enumerate(blk.volumes))
    loss.backward()
    np.testing.assert_allclose(f1.grad.cpu().numpy(), z["grad_fmap1"], rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(f2.grad.cpu().numpy(), z["grad_fmap2"], rtol=1e-5, atol=5e-5)
    # pooling is exact in fp32: level l+1 == 0.5 * (even + odd) of level l, bit for bit
    for l in range(3):
        v, nxt = blk.volumes[l].detach(), blk.volumes[l + 1].detach()
        w = nxt.shape[-1]
        assert torch.equal(nxt, 0.5 * (v[..., 0:2 * w:2] + v[..., 1:2 * w:2]))


def test_fused_lookup_matches_reference_golden_and_the_per_level_sampler_bitwise():
    import torch
    from gps_gaussian_amd import corr as K
    z = np.load(os.path.join(GOLDEN, "corr_sampler_golden.npz"))
    r = int(z["radius"])
    vols = [_t(z["volume%d" % l], grad=True) for l in range(4)]
    coords = _t(z["coords"])
    out = K._LookupPyramid.apply(coords, r, *vols)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=1e-5, atol=1e-5)
    out.backward(_t(z["grad_out"]))
    for l in range(4):
        np.testing.assert_allclose(vols[l].grad.cpu().numpy(), z["grad_volume%d" % l], rtol=1e-5, atol=1e-5)
        # identical arithmetic to four separate corr_sampler.forward / backward calls (the drop-in of row a14)
        ref, = K.forward(vols[l].detach(), coords / 2 ** l, r)
        assert torch.equal(out[:, l * (2 * r + 1):(l + 1) * (2 * r + 1)].detach(), ref)
        gref, = K.backward(vols[l].detach(), coords / 2 ** l, _t(z["grad_out"])[:, l * (2 * r + 1):(l + 1) * (2 * r + 1)].contiguous(), r)
        assert torch.equal(vols[l].grad, gref)


def test_convex_upsample_matches_reference_golden_fwd_bwd():
    from gps_gaussian_amd import corr as K
    z = np.load(os.path.join(GOLDEN, "upsample_golden.npz"))
    flow, mask = _t(z["flow"], grad=True), _t(z["mask"], grad=True)
    up = K.upsample_flow(flow, mask, int(z["factor"]))
    np.testing.assert_allclose(up.detach().cpu().numpy(), z["out"], rtol=1e-5, atol=1e-5)
    up.backward(_t(z["grad_out"]))
    np.testing.assert_allclose(flow.grad.cpu().numpy(), z["grad_flow"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(mask.grad.cpu().numpy(), z["grad_mask"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(4, 192, 128, 128, 128), (1, 37, 2, 70, 33), (2, 8, 1, 1, 9), (1, 64, 3, 130, 257)],
                         ids=["reference-size", "ragged", "tiny", "wide-odd"])
def test_volume_vs_oracle_at_full_and_ragged_sizes(shape):
    import torch
    from gps_gaussian_amd import corr as K
    from oracle import corr_oracle as CO
    N, D, H, W1, W2 = shape
    rng = np.random.default_rng(5)
    f1 = rng.standard_normal((N, D, H, W1)).astype(np.float32)
    f2 = rng.standard_normal((N, D, H, W2)).astype(np.float32)
    levels = 4
    t1, t2 = _t(f1, grad=True), _t(f2, grad=True)
    pyr = K._BuildPyramid.apply(t1, t2, levels)
    want = CO.build_pyramid(f1, f2, levels)
    for l in range(levels):
        assert tuple(pyr[l].shape) == want[l].shape
        if want[l].size:
            np.testing.assert_allclose(pyr[l].detach().cpu().numpy(), want[l], rtol=1e-5, atol=2e-5)
    gs = [rng.standard_normal(w.shape).astype(np.float32) for w in want]
    gs[2] = None if N == 1 else gs[2]   # a level without gradient
    loss = sum((p * _t(g)).sum() for p, g in zip(pyr, gs) if g is not None and g.size)
    loss.backward()
    g1, g2 = CO.build_pyramid_backward(f1, f2, [g if (g is not None and g.size) else None for g in gs])
    scale = max(np.abs(g1).max(), 1.0)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=1e-5, atol=1e-5 * max(np.abs(g2).max(), 1.0))


def test_fp16_path_matches_oracle_within_half_precision():
    import torch
    from gps_gaussian_amd import corr as K
    from oracle import corr_oracle as CO
    rng = np.random.default_rng(9)
    N, D, H, W = 2, 96, 4, 64
    f1 = rng.standard_normal((N, D, H, W)).astype(np.float16)
    f2 = rng.standard_normal((N, D, H, W)).astype(np.float16)
    blk = K.CorrBlockFast1D(_t(f1), _t(f2), num_levels=4, radius=4)
    want = CO.build_pyramid(f1.astype(np.float64), f2.astype(np.float64), 4)
    for l in range(4):
        assert blk.volumes[l].dtype == torch.float16
        np.testing.assert_allclose(blk.volumes[l].float().cpu().numpy(), want[l], rtol=2e-3, atol=2e-3)   # ~2 fp16 ulp at |corr| ~ 1
    coords = _t(rng.uniform(-3, W + 3, (N, 1, H, W)).astype(np.float32))
    out = blk(coords)
    wl = CO.lookup([v.float().cpu().numpy() for v in blk.volumes], coords.cpu().numpy(), 4)
    np.testing.assert_allclose(out.float().cpu().numpy(), wl, rtol=2e-3, atol=2e-3)


def test_mirror_of_the_reference_class_end_to_end_autograd():
    """CorrBlockFast1D(fmap1, fmap2)(coords) -> loss -> gradients of both feature maps, against the oracle chain."""
    from gps_gaussian_amd import corr as K
    from oracle import corr_oracle as CO
    rng = np.random.default_rng(3)
    N, D, H, W = 2, 48, 6, 80
    f1 = rng.standard_normal((N, D, H, W)).astype(np.float32)
    f2 = rng.standard_normal((N, D, H, W)).astype(np.float32)
    coords = np.stack([rng.uniform(-5, W + 5, (N, H, W)), np.zeros((N, H, W))], 1).astype(np.float32)   # the reference passes [N,2,H,W]
    gout = rng.standard_normal((N, 36, H, W)).astype(np.float32)
    t1, t2 = _t(f1, grad=True), _t(f2, grad=True)
    out = K.CorrBlockFast1D(t1, t2, num_levels=4, radius=4)(_t(coords))
    out.backward(_t(gout))
    pyr = CO.build_pyramid(f1, f2, 4)
    np.testing.assert_allclose(out.detach().cpu().numpy(), CO.lookup(pyr, coords[:, :1], 4), rtol=1e-5, atol=2e-5)
    gp = CO.lookup_backward([p.shape[-1] for p in pyr], coords[:, :1], gout, 4)
    g1, g2 = CO.build_pyramid_backward(f1, f2, gp)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=1e-5, atol=2e-5)


def test_upsample_at_reference_size_vs_oracle_and_partition_of_unity():
    import torch
    from gps_gaussian_amd import corr as K
    from oracle import corr_oracle as CO
    rng = np.random.default_rng(1)
    N, H, W, f = 2, 128, 128, 8
    flow = rng.standard_normal((N, 2, H, W)).astype(np.float32)
    mask = (3 * rng.standard_normal((N, 9 * f * f, H, W))).astype(np.float32)
    up = K.upsample_flow(_t(flow), _t(mask), f)
    np.testing.assert_allclose(up.cpu().numpy(), CO.upsample_flow(flow, mask, f), rtol=1e-5, atol=2e-5)
    # a constant flow is reproduced (times the factor) wherever the 3x3 window is inside the image: the weights sum to one
    const = torch.full((N, 2, H, W), 0.75, device="cuda")
    upc = K.upsample_flow(const, _t(mask), f)
    assert torch.allclose(upc[:, :, f:-f, f:-f], torch.full_like(upc[:, :, f:-f, f:-f], 0.75 * f), atol=1e-5)


def test_argument_errors():
    import torch
    from gps_gaussian_amd import corr as K
    with pytest.raises(RuntimeError):
        K.CorrBlockFast1D(torch.zeros(1, 4, 2, 8), torch.zeros(1, 4, 2, 8))            # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        K.CorrBlockFast1D(torch.zeros(1, 4, 2, 8).cuda(), torch.zeros(1, 5, 2, 8).cuda())
    with pytest.raises(RuntimeError):
        K.upsample_flow(torch.zeros(1, 2, 4, 4).cuda(), torch.zeros(1, 9 * 16 + 1, 4, 4).cuda(), 4)
    empty = K.CorrBlockFast1D(torch.zeros(0, 4, 2, 8).cuda(), torch.zeros(0, 4, 2, 8).cuda())
    assert empty.volumes[0].shape == (0, 2, 8, 8)
