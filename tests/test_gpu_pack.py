"""GPU (-m gpu): fused mask-compaction + pack against the fixture produced by the REFERENCE's own
pts2render (tests/golden/pts2render_golden.npz) and against torch's mask-gather path for the gradients."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _data_from_golden(dev, requires_grad=False, permuted_xyz=False):
    import torch
    gold = np.load(os.path.join(GOLDEN, "pts2render_golden.npz"))
    data = {}
    for v in ("lmain", "rmain"):
        d = {k: torch.from_numpy(gold["%s_%s" % (v, k)]).to(dev) for k in ("img", "xyz", "pts_valid", "rot_maps", "scale_maps", "opacity_maps")}
        if permuted_xyz:   # the reference's xyz is depth2pc(...).permute(0,2,1): a [B,S2,3] view of [B,3,S2] memory
            d["xyz"] = d["xyz"].permute(0, 2, 1).contiguous().permute(0, 2, 1)
            assert not d["xyz"].is_contiguous()
        if requires_grad:
            for k in ("xyz", "rot_maps", "scale_maps", "opacity_maps", "img"):
                d[k] = d[k].detach().clone().requires_grad_(True) if not permuted_xyz or k != "xyz" else d[k].requires_grad_(True)
        data[v] = d
    return gold, data


@pytest.mark.parametrize("permuted_xyz", [False, True])
def test_pack_is_bit_exact_against_reference_fixture(permuted_xyz):
    import torch
    from gps_gaussian_amd.pack import pack_views
    dev = torch.device("cuda:0")
    gold, data = _data_from_golden(dev, permuted_xyz=permuted_xyz)
    xyz, rgb, rot, scale, op, offsets = pack_views(data)
    offs = offsets.cpu().numpy()
    B = gold["lmain_img"].shape[0]
    assert offs[0] == 0 and len(offs) == B + 1
    for i in range(B):
        a, b = offs[i], offs[i + 1]
        assert b - a == gold["out%d_xyz" % i].shape[0]
        for t, name in ((xyz, "xyz"), (rgb, "rgb"), (rot, "rot"), (scale, "scale"), (op, "opacity")):
            np.testing.assert_array_equal(t[a:b].cpu().numpy(), gold["out%d_%s" % (i, name)])   # same rows, same order, same bits


def test_pack_backward_matches_torch_mask_gather():
    import torch
    from gps_gaussian_amd.pack import pack_views
    dev = torch.device("cuda:0")
    _, data = _data_from_golden(dev, requires_grad=True)
    out = pack_views(data)
    tot = int(out[5][-1])
    torch.manual_seed(0)
    ws = [torch.randn(tot, t.shape[1], device=dev) for t in out[:5]]
    loss = sum((t[:tot] * w).sum() for t, w in zip(out[:5], ws))
    loss.backward()
    got = {v: {k: data[v][k].grad.clone() for k in ("xyz", "img", "rot_maps", "scale_maps", "opacity_maps")} for v in data}
    # torch reference: the literal per-sample mask gathers of lib/GaussianRender.py:15-34
    for v in data:
        for k in got[v]:
            data[v][k].grad = None
    B = data["lmain"]["img"].shape[0]
    parts = [[], [], [], [], []]
    for i in range(B):
        for v in ("lmain", "rmain"):
            d = data[v]
            valid = d["pts_valid"][i]
            maps = (d["xyz"][i], d["img"][i].permute(1, 2, 0).reshape(-1, 3) * 0.5 + 0.5, d["rot_maps"][i].permute(1, 2, 0).reshape(-1, 4),
                    d["scale_maps"][i].permute(1, 2, 0).reshape(-1, 3), d["opacity_maps"][i].permute(1, 2, 0).reshape(-1, 1))
            for lst, m in zip(parts, maps):
                lst.append(m[valid])
    ref = [torch.cat(p, 0) for p in parts]
    for (a, b) in zip(out[:5], ref):
        assert torch.equal(a[:tot].detach(), b.detach())
    sum((t * w).sum() for t, w in zip(ref, ws)).backward()
    for v in data:
        for k in got[v]:
            assert torch.equal(got[v][k], data[v][k].grad), (v, k)


@pytest.mark.parametrize("mode,streams", [("batch", "0"), ("loop", "0"), ("loop", "1")], ids=["one-node-stream-per-sample", "loop-one-stream", "loop-stream-per-sample"])
def test_fused_pts2render_equals_unfused_and_oracle_path(mode, streams, monkeypatch):
    """pts2render (fused pack + the whole batch as ONE autograd node with the samples on their own streams: the default) and the literal
    per-sample version give the same image bits and the same map gradients -- also in the per-sample loop form (GPSGS_PTS2RENDER=loop),
    with the samples on one stream or on their own streams (GPSGS_PTS2RENDER_STREAMS=1)."""
    import torch
    monkeypatch.setenv("GPSGS_PTS2RENDER", mode)
    monkeypatch.setenv("GPSGS_PTS2RENDER_STREAMS", streams)
    from conftest import simple_scene
    from gps_gaussian_amd import render_api
    dev = torch.device("cuda:0")
    B, side = 2, 64
    cam = simple_scene(side, side, 48.0)

    def run(fn):
        _, data = _data_from_golden(dev, requires_grad=True)
        for v in ("lmain", "rmain"):
            data[v]["xyz"] = (data[v]["xyz"] * 0.1 + torch.tensor([0.0, 0.0, 2.0], device=dev))
            data[v]["scale_maps_in"] = data[v]["scale_maps"]
            data[v]["scale_maps"] = data[v]["scale_maps"] * 5
        data["novel_view"] = dict(
            FovX=torch.tensor([2 * np.arctan(cam["tanfovx"])] * B), FovY=torch.tensor([2 * np.arctan(cam["tanfovy"])] * B),
            width=torch.tensor([side] * B), height=torch.tensor([side] * B),
            world_view_transform=torch.from_numpy(cam["view"])[None].repeat(B, 1, 1),
            full_proj_transform=torch.from_numpy(cam["proj"])[None].repeat(B, 1, 1), camera_center=torch.zeros(B, 3))
        img = fn(data, [0.2, 0.3, 0.4])["novel_view"]["img_pred"]
        torch.manual_seed(1)
        (img * torch.randn_like(img)).sum().backward()
        g = {v: {k: data[v][k].grad.clone() for k in ("rot_maps", "scale_maps_in", "opacity_maps")} for v in ("lmain", "rmain")}
        return img.detach(), g

    img_f, g_f = run(render_api.pts2render)
    img_u, g_u = run(render_api.pts2render_unfused)
    assert torch.equal(img_f, img_u)
    for v in g_f:
        for k in g_f[v]:
            assert torch.equal(g_f[v][k], g_u[v][k]), (v, k)


def test_batch_with_a_sample_without_valid_points(monkeypatch):
    """A sample whose validity masks are all false has P = 0: upstream returns its zero-initialised image (NOT the background) and
    no gradient reaches its maps; the other sample of the batch is unaffected.  Both forms of pts2render agree with the literal mirror."""
    import torch
    from conftest import simple_scene
    from gps_gaussian_amd import render_api
    dev = torch.device("cuda:0")
    B, side = 2, 64
    cam = simple_scene(side, side, 48.0)
    res = {}
    for mode, fn in (("batch", render_api.pts2render), ("loop", render_api.pts2render), ("literal", render_api.pts2render_unfused)):
        monkeypatch.setenv("GPSGS_PTS2RENDER", "loop" if mode == "loop" else "batch")
        _, data = _data_from_golden(dev, requires_grad=True)
        for v in ("lmain", "rmain"):
            data[v]["xyz"] = (data[v]["xyz"] * 0.1 + torch.tensor([0.0, 0.0, 2.0], device=dev))
            data[v]["scale_maps_in"] = data[v]["scale_maps"]
            data[v]["scale_maps"] = data[v]["scale_maps"] * 5
            data[v]["pts_valid"] = data[v]["pts_valid"].clone()
            data[v]["pts_valid"][0] = False      # sample 0: nothing to draw
        data["novel_view"] = dict(
            FovX=torch.tensor([2 * np.arctan(cam["tanfovx"])] * B), FovY=torch.tensor([2 * np.arctan(cam["tanfovy"])] * B),
            width=torch.tensor([side] * B), height=torch.tensor([side] * B),
            world_view_transform=torch.from_numpy(cam["view"])[None].repeat(B, 1, 1),
            full_proj_transform=torch.from_numpy(cam["proj"])[None].repeat(B, 1, 1), camera_center=torch.zeros(B, 3))
        img = fn(data, [0.2, 0.3, 0.4])["novel_view"]["img_pred"]
        torch.manual_seed(1)
        (img * torch.randn_like(img)).sum().backward()
        res[mode] = (img.detach(), {v: data[v]["opacity_maps"].grad.clone() for v in ("lmain", "rmain")})
    img_b, g_b = res["batch"]
    assert float(img_b[0].abs().max()) == 0.0          # zero image, not the background colour
    assert float(img_b[1].abs().max()) > 0.0
    for v in g_b:
        assert float(g_b[v][0].abs().max()) == 0.0     # no gradient into the maps of the empty sample
        assert float(g_b[v][1].abs().max()) > 0.0
    for mode in ("loop", "literal"):
        assert torch.equal(res[mode][0], img_b), mode
        for v in g_b:
            assert torch.equal(res[mode][1][v], g_b[v]), (mode, v)


def _stage2_batch(dev, B=2, res=128, seed=5):
    import torch
    from gps_gaussian_amd import synthetic as S
    samples = [S.make_stereo_sample(res, 9000 + 2000 * i, seed=seed + i, render_res=2 * res) for i in range(B)]
    keys = ("xyz", "img", "rot_maps", "scale_maps", "opacity_maps")
    data = {}
    for v in ("lmain", "rmain"):
        data[v] = {k: torch.from_numpy(np.stack([s[v][k] for s in samples])).to(dev).requires_grad_(True) for k in keys}
        data[v]["pts_valid"] = torch.from_numpy(np.stack([s[v]["pts_valid"] for s in samples])).to(dev)
    nv = [s["novel_view"] for s in samples]
    data["novel_view"] = dict(FovX=torch.tensor([float(c["FovX"]) for c in nv]), FovY=torch.tensor([float(c["FovY"]) for c in nv]),
                              width=torch.tensor([c["width"] for c in nv]), height=torch.tensor([c["height"] for c in nv]),
                              world_view_transform=torch.from_numpy(np.stack([c["world_view_transform"] for c in nv])).to(dev),
                              full_proj_transform=torch.from_numpy(np.stack([c["full_proj_transform"] for c in nv])).to(dev),
                              camera_center=torch.from_numpy(np.stack([c["camera_center"] for c in nv])))
    return data, keys


def test_pts2render_never_reads_the_row_offsets_back(monkeypatch):
    """SURVEY section 8(f) row 1, "no host syncs": the batch path hands every view a DEVICE pointer to its row range; no .tolist() /
    .item() / .cpu() of a GPU tensor happens between the pack and the images (the capacity check reads pinned host memory the scan
    kernel wrote: not a stream synchronisation)."""
    import torch
    from gps_gaussian_amd import render_api
    dev = torch.device("cuda:0")
    data, _ = _stage2_batch(dev)
    render_api.pts2render(data, [0, 0, 0])            # warm-up: capacities learnt
    torch.cuda.synchronize()
    data, _ = _stage2_batch(dev)
    with torch.cuda.stream(torch.cuda.Stream()):      # set_sync_debug_mode does not police the legacy default stream
        torch.cuda.set_sync_debug_mode("error")
        try:
            img = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"]
        finally:
            torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert float(img.abs().max()) > 0


def test_pack_render_loss_backward_chain_replays_from_one_hip_graph(monkeypatch):
    """GPSGS_CHECK=none: pack -> B renders -> L1 + SSIM -> backward down to the per-pixel maps is launches + memsets only; captured once,
    replayed, it reproduces the eager step bit for bit -- also after the maps were changed in place (the row offsets are recomputed
    on the device inside the graph)."""
    import torch
    from gps_gaussian_amd import loss as L
    from gps_gaussian_amd import render_api
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        data, keys = _stage2_batch(dev)
        gt = torch.rand(2, 3, 256, 256, device=dev)
        leaves = [data[v][k] for v in ("lmain", "rmain") for k in keys]

        def step():
            img = render_api.pts2render(data, [0, 0, 0])["novel_view"]["img_pred"]
            return img, torch.autograd.grad(L.stage2_photometric_loss(img, gt), leaves)

        img_e, g_e = step()                           # eager, exact capacity check: learns the capacities
        monkeypatch.setenv("GPSGS_CHECK", "none")
        step()
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            img_g, g_g = step()
        graph.replay()
        st.synchronize()
        assert torch.equal(img_g, img_e)
        for a, b in zip(g_g, g_e):
            assert torch.equal(a, b)
        with torch.no_grad():                         # fewer valid pixels in sample 0, other opacities: same graph
            data["lmain"]["pts_valid"][0, ::3] = False
            data["rmain"]["opacity_maps"].mul_(0.7)
        graph.replay()
        st.synchronize()
        monkeypatch.setenv("GPSGS_CHECK", "sync")
        img_e2, g_e2 = step()
        st.synchronize()
        assert torch.equal(img_g, img_e2) and not torch.equal(img_e2, img_e)
        for a, b in zip(g_g, g_e2):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
