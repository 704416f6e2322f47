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
