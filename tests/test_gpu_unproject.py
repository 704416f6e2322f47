"""GPU (-m gpu): fused flow2depth + depth2pc against fixtures from the reference's lib/utils.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_forward_and_backward_match_reference_fixture():
    import torch
    from gps_gaussian_amd.unproject import unproject
    g = np.load(os.path.join(GOLDEN, "unproject_golden.npz"))
    dev = torch.device("cuda:0")
    flow = torch.from_numpy(g["flow"]).to(dev).requires_grad_(True)
    depth, xyz, valid = unproject(flow, torch.from_numpy(g["mask"]).to(dev), torch.from_numpy(g["ref_intr"]), torch.from_numpy(g["intr"]),
                                  torch.from_numpy(g["extr"]), torch.from_numpy(g["Tf_x"]))
    np.testing.assert_array_equal(depth.detach().cpu().numpy(), g["depth"])           # same operation order: bit-identical
    np.testing.assert_array_equal(valid.cpu().numpy(), g["valid"])
    ok = g["valid"]
    x, xr = xyz.detach().cpu().numpy(), g["xyz"]
    np.testing.assert_allclose(x[ok], xr[ok], rtol=2e-6, atol=2e-6)                   # 3x3 product: summation order only
    np.testing.assert_allclose(x[~ok], xr[~ok], rtol=1e-5)                            # masked pixels sit at ~1e8 m
    gx = torch.from_numpy(g["g_xyz"]).to(dev)
    loss = (depth * torch.from_numpy(g["g_depth"]).to(dev)).sum() + (xyz * gx).sum()
    loss.backward()
    got, want = flow.grad.cpu().numpy(), g["g_flow"]
    v2 = ok.reshape(want.shape)
    np.testing.assert_allclose(got[v2], want[v2], rtol=2e-5, atol=2e-5 * np.abs(want[v2]).max())
    assert (got[~v2] == 0).all() and (want[~v2] == 0).all()                          # masked pixels get no gradient


def test_feeds_the_pack_and_raster_path():
    """unproject -> pack -> render: the xyz/valid it produces are consumed by the fused pack (contiguous layout this time)."""
    import torch
    from gps_gaussian_amd.pack import pack_views
    from gps_gaussian_amd.unproject import unproject
    g = np.load(os.path.join(GOLDEN, "unproject_golden.npz"))
    dev = torch.device("cuda:0")
    B, S = g["flow"].shape[0], g["flow"].shape[-1]
    data = {}
    for v in ("lmain", "rmain"):
        depth, xyz, valid = unproject(torch.from_numpy(g["flow"]).to(dev), torch.from_numpy(g["mask"]).to(dev), torch.from_numpy(g["ref_intr"]),
                                      torch.from_numpy(g["intr"]), torch.from_numpy(g["extr"]), torch.from_numpy(g["Tf_x"]))
        data[v] = dict(xyz=xyz, pts_valid=valid, img=torch.rand(B, 3, S, S, device=dev), rot_maps=torch.rand(B, 4, S, S, device=dev),
                       scale_maps=torch.rand(B, 3, S, S, device=dev), opacity_maps=torch.rand(B, 1, S, S, device=dev))
    out = pack_views(data)
    offs = out[5].cpu().numpy()
    assert offs[-1] == 2 * int(g["valid"].sum())
    n0 = int(g["valid"][0].sum())
    np.testing.assert_array_equal(out[0][:n0].cpu().numpy(), data["lmain"]["xyz"][0][data["lmain"]["pts_valid"][0]].cpu().numpy())


def test_batches_larger_than_one_launch_chunk():
    """The cameras travel by value in the kernel arguments, 16 samples per launch; a batch of 37 (the fixture's samples tiled) must give
    every sample the same bits as a batch of its own -- the reference has no batch limit."""
    import torch
    from gps_gaussian_amd.unproject import unproject
    g = np.load(os.path.join(GOLDEN, "unproject_golden.npz"))
    dev = torch.device("cuda:0")
    B0 = g["flow"].shape[0]
    idx = np.arange(37) % B0
    args = lambda ii: (torch.from_numpy(g["mask"][ii]).to(dev), torch.from_numpy(g["ref_intr"][ii]), torch.from_numpy(g["intr"][ii]),
                       torch.from_numpy(g["extr"][ii]), torch.from_numpy(g["Tf_x"][ii]))
    flow = torch.from_numpy(g["flow"][idx]).to(dev).requires_grad_(True)
    depth, xyz, valid = unproject(flow, *args(idx))
    gx = torch.from_numpy(g["g_xyz"][idx]).to(dev)
    (xyz * gx).sum().backward()
    flow1 = torch.from_numpy(g["flow"]).to(dev).requires_grad_(True)
    d1, x1, v1 = unproject(flow1, *args(np.arange(B0)))
    (x1 * torch.from_numpy(g["g_xyz"]).to(dev)).sum().backward()
    for b, src in enumerate(idx):
        assert torch.equal(depth[b], d1[src]) and torch.equal(xyz[b], x1[src]) and torch.equal(valid[b], v1[src])
        assert torch.equal(flow.grad[b], flow1.grad[src])


def test_host_camera_and_device_camera_entry_points_give_the_same_bits():
    """The C-ABI has two forms (include/gpsgs.h): cameras as HOST arrays (by value into the kernel arguments, 16 samples per launch: what a C host
    with its calibration in memory calls) and as a DEVICE array [B,31] (what the Python wrapper uses: no read-back, no synchronisation in the middle of
    the network forward).  Same arithmetic, same bits, forward and backward, for a batch beyond one host-form chunk; and the device form must not
    synchronise the host."""
    import ctypes as C
    import torch
    from gps_gaussian_amd import _capi
    from gps_gaussian_amd.unproject import unproject
    g = np.load(os.path.join(GOLDEN, "unproject_golden.npz"))
    dev = torch.device("cuda:0")
    lib = _capi.lib()
    B0 = g["flow"].shape[0]
    idx = np.arange(19) % B0
    B, S = len(idx), g["flow"].shape[-1]
    flow = torch.from_numpy(g["flow"][idx]).to(dev).requires_grad_(True)
    mask = torch.from_numpy(g["mask"][idx]).to(dev).float().contiguous()
    cams = [torch.from_numpy(g[k][idx]).to(dev) for k in ("ref_intr", "intr", "extr", "Tf_x")]
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")        # any host synchronisation inside the wrapper raises
    try:
        depth, xyz, valid = unproject(flow, mask, *cams)
        gx = torch.from_numpy(g["g_xyz"][idx]).pin_memory().to(dev, non_blocking=True)
        (xyz * gx).sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    # the host-array form, called directly
    host = [np.ascontiguousarray(a, dtype=np.float32) for a in (g["ref_intr"][idx].reshape(B, 9), g["intr"][idx].reshape(B, 9), g["extr"][idx][:, :3, :4].reshape(B, 12),
                                                               g["Tf_x"][idx].reshape(B))]
    hp = [a.ctypes.data_as(C.c_void_p) for a in host]
    d2 = torch.empty_like(depth); x2 = torch.empty_like(xyz); v2 = torch.empty((B, S * S), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    f = flow.detach().contiguous()
    assert lib.up_unproject_forward(B, S, p(f), p(mask), mask.stride(0), *hp, p(d2), p(x2), p(v2), st) == 0
    gflow2 = torch.empty_like(f)
    assert lib.up_unproject_backward(B, S, p(d2), p(mask), mask.stride(0), *hp, None, p(gx), gx.stride(0), gx.stride(1), gx.stride(2), p(gflow2), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(depth.detach(), d2) and torch.equal(xyz.detach(), x2) and torch.equal(valid, v2.bool()) and torch.equal(flow.grad, gflow2)
