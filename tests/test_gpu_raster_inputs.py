"""GPU (-m gpu): inputs the reference's real callers produce that the other parity files leave cold ("Weak" 1-3):

  * the GradScaler's loss scale (train_stage2.py:83 `scaler.scale(loss).backward()`, initial scale 65536): dL/dpix x 65536 must give
    65536 x the gradients, bit for bit (a power of two; the backward is linear in dL/dpix and atomic-free);
  * a NON-black background at BASELINE config 2's size (the `walk(std::true_type{})` instantiation of the tile backward, whose
    background term of dL/dalpha the black-background stage-2 path leaves out), lists spanning many 64-splat rounds;
  * splat centres EXACTLY on pixel centres (power == 0 exactly at the centre pixel: upstream's `power > 0` skip must not fire, and the
    tile family, which evaluates the exponent exactly, must agree with the fp32-order oracle there);
  * a pts2render batch whose samples differ in size 1 : 10 : 0 through the one-node batch path.
"""
import numpy as np
import pytest

from conftest import assert_grad_parity, fragile_bounds, hip_render, oracle_render, parity_report, simple_scene, touched_by_fragile

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


@pytest.fixture(params=["valu", "tiles"])
def family(request, monkeypatch):
    monkeypatch.setenv("GPSGS_COMPOSITE", request.param)
    return request.param


@pytest.fixture(params=["compiled", "python"])
def host(request, monkeypatch):
    """Both host paths of the drop-in module: the compiled one (csrc/host_shim.cpp, a C++ autograd Function) and the Python one (rasterizer.py)."""
    monkeypatch.setenv("GPSGS_FAST", "1" if request.param == "compiled" else "0")
    return request.param


@pytest.mark.parametrize("bg", [(0.0, 0.0, 0.0), (0.3, 0.1, 0.2)], ids=["black", "colour"])
def test_gradscaler_loss_scale_is_exact(family, bg):
    from gps_gaussian_amd import synthetic as S
    g = S.make_scene(256, 30000, render_res=512)
    g["bg"] = np.asarray(bg, np.float32)
    dpix = np.random.default_rng(3).standard_normal((3, 512, 512)).astype(np.float32)
    _, _, g1, _ = hip_render(g, dpix)
    _, _, g2, _ = hip_render(g, 65536.0 * dpix)
    for k in g1:
        assert np.abs(g1[k]).max() > 0 or k == "means2D"
        np.testing.assert_array_equal(g2[k], 65536.0 * g1[k], err_msg=k)
    # and the scaled gradients still equal the oracle's on the scaled seed (nothing saturates at the GradScaler's magnitudes)
    o, _, oradii = oracle_render(g, "f32")
    og = o.backward(65536.0 * dpix)
    _, touched, bounds = fragile_bounds(o, 65536.0 * dpix)
    assert_grad_parity(g2, og, touched, oradii > 0, bounds=bounds)


def test_nonblack_background_at_config2_size(family):
    from test_gpu_raster import _assert_full_size_grads
    from gps_gaussian_amd import synthetic as S
    g = S.make_scene(1024, 600000)
    g["bg"] = np.array([0.3, 0.1, 0.2], np.float32)
    dpix = np.random.default_rng(4).standard_normal((3, 1024, 1024)).astype(np.float32)
    img, radii, grads, _ = hip_render(g, dpix)
    o, oimg, oradii = oracle_render(g, "f32")
    np.testing.assert_array_equal(radii, oradii)
    solid, touched, bounds = fragile_bounds(o, dpix)
    err = np.abs(img - oimg).max(0)
    og = o.backward(dpix)
    parity_report("config2_colour_background[%s]" % family, img, oimg, grads, og, solid, touched, visible=oradii > 0, bounds=bounds)
    assert solid.mean() > 0.998 and err[solid].max() <= RGB_TOL and (err > RGB_TOL).sum() <= 100
    assert float(np.abs(img[:, 0, 0] - g["bg"]).max()) == 0.0           # an empty corner pixel shows the background exactly
    _assert_full_size_grads(grads, og, touched, bounds)
    assert ((oradii > 0) & ~touched).sum() >= 0.5 * (oradii > 0).sum()
    # the background term is live: the same scene on black gives different opacity gradients
    g0 = dict(g); g0["bg"] = np.zeros(3, np.float32)
    _, _, grads0, _ = hip_render(g0, dpix)
    assert np.abs(grads0["opacities"] - grads["opacities"]).max() > 1e-3 * np.abs(grads["opacities"]).max()


def test_splat_centres_exactly_on_pixel_centres(family):
    """fx = W / 2 = 128, cx = cy = 128, depths that are powers of two >= 2: every product on the way to the pixel coordinate is exact in
    fp32 (1 / (z + 1e-7) rounds to 1 / z exactly), so the projected centres are integers + exactly 0 and power == 0 at the centre pixel."""
    import torch
    from gps_gaussian_amd import rasterizer as RZ
    W = H = 256
    cam = simple_scene(W, H, 128.0, bg=(0.05, 0.1, 0.15))
    rng = np.random.default_rng(8)
    P = 20000
    u, v = rng.integers(0, W, P), rng.integers(0, H, P)
    z = np.array([2.0, 4.0, 8.0])[rng.integers(0, 3, P)]
    xyz = np.stack([(u + 0.5 - 128.0) * z / 128.0, (v + 0.5 - 128.0) * z / 128.0, z], 1).astype(np.float32)
    q = rng.standard_normal((P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g = dict(cam, means3D=xyz, colors=rng.uniform(0, 1, (P, 3)).astype(np.float32), rotations=q,
             scales=(np.exp(rng.normal(np.log(0.02), 0.5, (P, 3))) * z[:, None] / 2).astype(np.float32),
             opacities=(1 / (1 + np.exp(-rng.normal(0.0, 2.0, (P, 1))))).astype(np.float32))
    dpix = rng.standard_normal((3, H, W)).astype(np.float32)
    img, radii, grads, info = hip_render(g, dpix)
    o, oimg, oradii = oracle_render(g, "f32")
    np.testing.assert_array_equal(radii, oradii)
    st = RZ.export_state(info["ws"], P, W, H, info["cap"], info.get("bin_cap", 0))
    xy = st["xy"].cpu().numpy()
    vis = oradii > 0
    on_centre = (xy[vis] == np.round(xy[vis])).all(1)
    assert on_centre.mean() > 0.99, on_centre.mean()                      # the construction really snaps the centres
    # compare EVERYTHING the alpha / T thresholds leave solid -- including the centre pixels, where power == 0 exactly
    solid, touched, bounds = fragile_bounds(o, dpix, power_band=0.0)
    solid_default, _ = touched_by_fragile(o)                              # the default band also flags |power| < 1e-6: reported
    err = np.abs(img - oimg).max(0)
    og = o.backward(dpix)
    parity_report("pixel_centre_snapped[%s]" % family, img, oimg, grads, og, solid, touched, visible=vis,
                  extra=dict(pixels_flagged_by_power_band=int((solid & ~solid_default).sum()), centres_on_pixel_centres=float(on_centre.mean())), bounds=bounds)
    assert (solid & ~solid_default).sum() > 1000                           # thousands of pixels carry a pair with power == 0 ...
    assert solid.mean() > 0.98 and err[solid].max() <= RGB_TOL, err[solid].max()   # ... and every one of them agrees with the oracle
    assert_grad_parity(grads, og, touched, vis, bounds=bounds)


def test_render_batch_with_samples_of_very_different_size(monkeypatch):
    """_RenderBatch (ONE autograd node for the batch) with P ratios 1 : 10 : 0 -- a small sample, a full one, an empty one -- against
    the literal per-sample mirror: same image bits, same map gradient bits."""
    import torch
    from gps_gaussian_amd import render_api, synthetic as S
    dev = torch.device("cuda:0")
    res, B = 256, 3
    samples = [S.make_stereo_sample(res, 40000, seed=21 + i, render_res=2 * res) for i in range(B)]
    rng = np.random.default_rng(2)
    for v in ("lmain", "rmain"):
        keep = rng.uniform(size=samples[0][v]["pts_valid"].shape) < 0.1
        samples[0][v]["pts_valid"] = samples[0][v]["pts_valid"] & keep      # 1 : 10
        samples[2][v]["pts_valid"] = np.zeros_like(samples[2][v]["pts_valid"])   # : 0
    keys = ("xyz", "img", "rot_maps", "scale_maps", "opacity_maps")

    def make():
        data = {}
        for v in ("lmain", "rmain"):
            data[v] = {k: torch.from_numpy(np.stack([s[v][k] for s in samples])).to(dev).requires_grad_(True) for k in keys}
            data[v]["pts_valid"] = torch.from_numpy(np.stack([s[v]["pts_valid"] for s in samples])).to(dev)
        nv = [s["novel_view"] for s in samples]
        data["novel_view"] = dict(FovX=torch.tensor([float(c["FovX"]) for c in nv]), FovY=torch.tensor([float(c["FovY"]) for c in nv]),
                                  width=torch.tensor([c["width"] for c in nv]), height=torch.tensor([c["height"] for c in nv]),
                                  world_view_transform=torch.from_numpy(np.stack([c["world_view_transform"] for c in nv])).pin_memory(),
                                  full_proj_transform=torch.from_numpy(np.stack([c["full_proj_transform"] for c in nv])).pin_memory(),
                                  camera_center=torch.from_numpy(np.stack([c["camera_center"] for c in nv])))
        return data

    gout = torch.randn(B, 3, 2 * res, 2 * res, generator=torch.Generator().manual_seed(6)).to(dev)
    out = {}
    for name, fn in (("batch", render_api.pts2render), ("literal", render_api.pts2render_unfused)):
        data = make()
        img = fn(data, [0.0, 0.0, 0.0])["novel_view"]["img_pred"]
        img.backward(gout)
        torch.cuda.synchronize()
        out[name] = (img.detach(), {v: {k: data[v][k].grad.clone() for k in keys} for v in ("lmain", "rmain")})
    n_valid = [int(sum(s[v]["pts_valid"].sum() for v in ("lmain", "rmain"))) for s in samples]
    assert n_valid[2] == 0 and 5 * n_valid[0] < n_valid[1]
    assert torch.equal(out["batch"][0], out["literal"][0])
    assert float(out["batch"][0][2].abs().max()) == 0.0 and float(out["batch"][0][0].abs().max()) > 0
    for v in ("lmain", "rmain"):
        for k in keys:
            assert torch.equal(out["batch"][1][v][k], out["literal"][1][v][k]), (v, k)


@pytest.mark.parametrize("bg", [(0.0, 0.0, 0.0), (0.3, 0.1, 0.2)], ids=["black", "colour"])
def test_backward_without_the_colour_gradient(family, bg, host):
    """Stage 2 never differentiates the colours (input pixels: colors_precomp.requires_grad is False in the reference's training loop).
    The autograd module then asks the backward to leave the colour sums out (GSR_FLAG_NO_COLOR_GRAD): every other gradient must be
    bit-identical to the full backward, the colours get no gradient."""
    import torch
    from gps_gaussian_amd import rasterizer as RZ
    from gps_gaussian_amd import synthetic as S
    g = S.make_scene(256, 30000, render_res=512)
    g["bg"] = np.asarray(bg, np.float32)
    dpix = np.random.default_rng(9).standard_normal((3, 512, 512)).astype(np.float32)
    _, _, full, _ = hip_render(g, dpix)
    dev = torch.device("cuda:0")
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    t = {k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32)).to(dev).requires_grad_(k != "colors") for k in names}
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                          torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                          torch.from_numpy(g["campos"]).to(dev), False, False)
    img, _ = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"],
                                       scales=t["scales"], rotations=t["rotations"])
    assert getattr(img.grad_fn, "color_grad", False) is False  # (the Python host path's node says so; the compiled one decides from needs_input_grad)
    img.backward(torch.from_numpy(dpix).to(dev))
    assert t["colors"].grad is None
    for k in ("means3D", "opacities", "scales", "rotations"):
        np.testing.assert_array_equal(t[k].grad.cpu().numpy(), full[k], err_msg=k)
    np.testing.assert_array_equal(m2.grad.cpu().numpy(), full["means2D"])
    assert np.abs(full["colors"]).max() > 0


@pytest.mark.gpu
def test_mark_visible_is_the_near_plane_test_of_the_forward():
    """GaussianRasterizer.markVisible(positions) (upstream K10, the one member of the imported class round 5 lacked): bool [P], true iff view-space
    z > 0.2.  Checked against the oracle's restatement, against the forward itself (radii > 0 implies visible; a visible point with radius 0 was
    dropped later: zero determinant / empty tile rect), on points straddling the near plane bit by bit, and for the error behaviour of the module."""
    import torch
    from gps_gaussian_amd import rasterizer as RZ, synthetic as S
    from oracle.gsr_oracle import mark_visible
    g = S.make_uniform_cloud(50000, 320, 200, seed=9, scale_med=0.02, z_range=(0.05, 3.0), behind_frac=0.2)
    # a run of points whose view-space depth steps through the fp32 neighbours of 0.2 (identity rotation: z_view = z + t_z)
    zs = np.float32(0.2) + np.arange(-8, 9, dtype=np.float32) * np.float32(np.spacing(np.float32(0.2)))
    w2c = np.asarray(g["view"], np.float32).reshape(4, 4).T
    extra = np.linalg.solve(w2c[:3, :3].astype(np.float64), (np.stack([np.zeros(17), np.zeros(17), zs.astype(np.float64)], 1) - w2c[:3, 3]).T).T
    pos = np.concatenate([g["means3D"], extra.astype(np.float32)])
    dev = torch.device("cuda:0")
    rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                          torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                          torch.from_numpy(g["campos"]).to(dev), False, False)
    rz = RZ.GaussianRasterizer(rs)
    vis = rz.markVisible(torch.from_numpy(pos).to(dev))
    assert vis.dtype is torch.bool and tuple(vis.shape) == (pos.shape[0],) and not vis.requires_grad
    want = mark_visible(pos, g["view"])
    np.testing.assert_array_equal(vis.cpu().numpy(), want)
    assert 0 < want[-17:].sum() < 17 and 0.05 < want.mean() < 0.99        # both sides of the plane are present, also among the neighbours of 0.2
    _, radii, _, _ = hip_render(g)
    v0 = vis.cpu().numpy()[:g["means3D"].shape[0]]
    assert (v0[radii > 0]).all() and (radii[~v0] == 0).all()
    with pytest.raises(RuntimeError):
        rz.markVisible(torch.from_numpy(pos))                              # CPU tensor: no fallback
    assert rz.markVisible(torch.empty((0, 3), device=dev)).numel() == 0


def test_compiled_and_python_host_paths_give_identical_bits_and_the_compiled_one_is_taken(monkeypatch):
    """The compiled host path (lib/_gpsgs_fast.so) calls the same C-ABI with the same arguments as rasterizer.py's Python path: image, radii and all six
    gradients must agree bit for bit -- and on a machine with a GPU the compiled path must really be the one that ran (no silent fallback): the image's
    grad_fn is the C++ node.  Non-contiguous / half-precision dL/dpix (hazard H3), inference under no_grad and an overflowing first call included."""
    import torch
    from gps_gaussian_amd import rasterizer as RZ, synthetic as S
    dev = torch.device("cuda:0")
    g = S.make_scene(256, 30000, render_res=384)
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                          torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                          torch.from_numpy(g["campos"]).to(dev), False, False)
    gout = torch.randn(g["H"], g["W"], 3, device=dev).permute(2, 0, 1)  # not contiguous

    def run(fast, grad=True):
        monkeypatch.setenv("GPSGS_FAST", "1" if fast else "0")
        t = {k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32)).to(dev).requires_grad_(grad) for k in names}
        m2 = torch.zeros_like(t["means3D"], requires_grad=grad)
        img, radii = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=None, colors_precomp=t["colors"],
                                               scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        node = img.grad_fn.name() if img.grad_fn is not None else None
        if grad:
            img.backward(gout)
        return img.detach(), radii, [t[k].grad for k in names] + [m2.grad], node

    assert RZ._fast_module() is not False and RZ._fast_module().abi_version() == 4
    a, b = run(True), run(False)
    assert "CppNode" in a[3] and "Rasterize" in a[3] and "CppNode" not in b[3], (a[3], b[3])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert x is not None and torch.equal(x, y)
    with torch.no_grad():
        c = run(True, grad=False)
    assert c[3] is None and torch.equal(c[0], a[0]) and not c[0].requires_grad
    # an overflowing first attempt: the compiled path reports it, the Python path repairs it -- same bits, and the capacity is learnt
    RZ._state.clear()
    real, calls = RZ._capacity_for, []
    monkeypatch.setattr(RZ, "_capacity_for", lambda st, P: (calls.append(1), 1024 if len(calls) == 1 else real(st, P))[1])
    d = run(True)
    assert len(calls) >= 2 and torch.equal(d[0], a[0])
    for x, y in zip(d[2], a[2]):
        assert torch.equal(x, y)
    monkeypatch.undo()
    assert RZ._fast_module().slots_in_flight(0) == 0
    # what the compiled path does not take falls through to the Python path: fp16 colours are converted there
    monkeypatch.setenv("GPSGS_FAST", "1")
    t = {k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32)).to(dev) for k in names}
    img16, _ = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], colors_precomp=t["colors"].half(),
                                         scales=t["scales"], rotations=t["rotations"])
    assert (img16 - a[0]).abs().max() < 2e-3



def test_two_host_threads_through_the_compiled_host_path(monkeypatch):
    """The compiled host path releases the GIL while it enqueues and spins on its notification word; two host threads rendering different views through
    GaussianRasterizer + backward at the same time (each on its own HIP stream) must get exactly what each gets alone: per-thread header hand-back,
    notification slots handed out under a mutex, capacity policy updated under the module's lock."""
    import threading
    import torch
    from gps_gaussian_amd import rasterizer as RZ, synthetic as S
    monkeypatch.setenv("GPSGS_FAST", "1")
    dev = torch.device("cuda:0")
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    scenes = [S.make_scene(256, 30000, render_res=320), S.make_uniform_cloud(9000, 280, 200, seed=31, scale_med=0.03)]

    def render(g, n, out, stream=None):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            rs = RZ.GaussianRasterizationSettings(g["H"], g["W"], g["tanfovx"], g["tanfovy"], torch.from_numpy(g["bg"]).to(dev), 1.0,
                                                  torch.from_numpy(g["view"]).to(dev), torch.from_numpy(g["proj"]).to(dev), 3,
                                                  torch.from_numpy(g["campos"]).to(dev), False, False)
            gout = torch.from_numpy(np.random.default_rng(3).standard_normal((3, g["H"], g["W"])).astype(np.float32)).to(dev)
            res, fast = None, 0
            for it in range(n):
                t = {k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32)).to(dev).requires_grad_(True) for k in names}
                m2 = torch.zeros_like(t["means3D"], requires_grad=True)
                img, radii = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors"],
                                                       scales=t["scales"], rotations=t["rotations"])
                # (a view that outgrows the learnt capacity is handed to the Python path, which repairs it: the first of a scene may)
                fast += "CppNode" in img.grad_fn.name()
                img.backward(gout)
                cur = [img.detach().clone(), radii.clone()] + [t[k].grad.clone() for k in names] + [m2.grad.clone()]
                if res is not None:
                    assert all(torch.equal(a, b) for a, b in zip(res, cur))
                res = cur
            torch.cuda.current_stream().synchronize()
            assert fast >= n - 2, (fast, n)
            out.append(res)

    alone = []
    for g in scenes:
        render(g, 2, alone)
    both, errs = [[], []], []

    def worker(i):
        try:
            render(scenes[i], 25, both[i], torch.cuda.Stream(device=dev))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for i in range(2):
        assert all(torch.equal(a, b) for a, b in zip(alone[i], both[i][0])), i
    assert RZ._fast_module().slots_in_flight(0) == 0
