"""CPU: pins oracle/gsr_oracle.c (fp32 and fp64 builds) against analytic known answers and an independent autograd restatement."""
import numpy as np
import pytest

import kat_cases
from conftest import oracle_render

import gps_gaussian_amd  # noqa: F401
from gps_gaussian_amd import synthetic as S
from oracle.gsr_oracle import OracleRasterizer
from oracle.gsr_torch_ref import grads_ref


def _render(kind):
    def f(scene):
        o, img, radii = oracle_render(scene, kind)
        extra = o.binning() if scene["means3D"].shape[0] else None
        return img, radii, extra
    f.fp32_decisions = kind == "f32"
    return f


@pytest.mark.parametrize("case", kat_cases.ALL, ids=lambda c: c.__name__)
@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_known_answers(case, kind):
    # inputs are fp32-rounded (0.05f, 0.7f ...), so even the fp64 build is compared at ~1e-6
    case(_render(kind), 2e-6 if kind == "f32" else 5e-7)


@pytest.mark.parametrize("seed,W,H,n,scale", [(0, 40, 36, 120, 0.05), (1, 48, 32, 200, 0.15), (2, 33, 47, 150, 0.02)])
def test_backward_matches_independent_autograd(seed, W, H, n, scale):
    """The hand-derived backward of the C oracle (fp64) equals torch.autograd through an independent forward restatement."""
    g = S.make_uniform_cloud(n, W, H, seed=seed, scale_med=scale, z_range=(0.3, 4.0))
    dpix = np.random.default_rng(seed).standard_normal((3, H, W))
    o, img, radii = oracle_render(g, "f64")
    gr = o.backward(dpix)
    img2, radii2, gr2 = grads_ref(g, W, H, g["tanfovx"], g["tanfovy"], dpix)
    assert (radii == radii2).all() and (radii > 0).sum() > n // 3
    np.testing.assert_allclose(img, img2, atol=1e-12)
    for k in gr:
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


@pytest.mark.parametrize("seed,W,H,n,scale,zr", [(11, 64, 64, 400, 0.3, (1.0, 3.0)), (12, 57, 23, 300, 0.004, (0.25, 0.8)),
                                                   (13, 24, 72, 350, 0.05, (0.21, 9.0)), (14, 80, 48, 500, 0.6, (2.0, 2.2)),
                                                   (15, 16, 16, 250, 0.1, (0.5, 1.5)), (16, 96, 96, 800, 0.02, (0.4, 5.0))])
def test_backward_matches_independent_autograd_sweep(seed, W, H, n, scale, zr):
    """The same pin over screen-filling and sub-pixel splats, a depth range that starts at the near plane, nearly coplanar layers
    (many pixels saturate: the T < 1e-4 stop and the 0.99 clamp are active) and non-square images, with exactly-zero and
    exactly-one opacities mixed in."""
    g = S.make_uniform_cloud(n, W, H, seed=seed, scale_med=scale, z_range=zr, behind_frac=0.1)
    g["opacities"][:5] = 0.0
    g["opacities"][5:25] = 1.0
    dpix = np.random.default_rng(seed).standard_normal((3, H, W))
    o, img, radii = oracle_render(g, "f64")
    gr = o.backward(dpix)
    img2, radii2, gr2 = grads_ref(g, W, H, g["tanfovx"], g["tanfovy"], dpix)
    assert (radii == radii2).all() and (radii > 0).sum() > n // 4
    np.testing.assert_allclose(img, img2, atol=1e-12)
    for k in gr:
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


@pytest.mark.parametrize("seed,W,H,n,mod", [(3, 128, 128, 2000, 1.0), (4, 96, 80, 700, 0.5)])
def test_backward_matches_independent_autograd_with_the_fov_clamp_active(seed, W, H, n, mod):
    """Same pin on the branches the uniform cloud leaves cold: general (rotated + translated) view matrix, centres far outside the
    1.3 * tanfov guard band (clamped view-space x/y with zeroed gradient terms), non-unit quaternions, scale_modifier != 1."""
    from conftest import clamp_active
    g = S.make_clamp_cloud(n, W, H, seed=seed, scale_med=0.08 / mod)
    g["scale_modifier"] = mod
    dpix = np.random.default_rng(seed).standard_normal((3, H, W))
    o, img, radii = oracle_render(g, "f64")
    gr = o.backward(dpix)
    img2, radii2, gr2 = grads_ref(g, W, H, g["tanfovx"], g["tanfovy"], dpix, scale_modifier=mod)
    assert (radii == radii2).all()
    contributing = np.abs(gr2["means3D"]).max(1) > 0
    frac = (clamp_active(g) & contributing).sum() / max(1, contributing.sum())
    qn = np.linalg.norm(g["rotations"], axis=1)
    print("contributing %d of %d, clamp active on %.0f %% of them, |q| in [%.2f, %.2f]" % (contributing.sum(), n, 100 * frac, qn.min(), qn.max()))
    assert contributing.sum() >= n // 10 and frac >= 0.05 and qn.min() < 0.6 and qn.max() > 1.8
    np.testing.assert_allclose(img, img2, atol=1e-12)
    for k in gr:
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


def test_fp32_and_fp64_builds_agree_on_config1():
    """BASELINE config 1 (256^2, ~30k Gaussians, CPU only): the two builds agree away from branch thresholds."""
    g = S.make_scene(256, 30000)
    o32, i32, r32 = oracle_render(g, "f32")
    o64, i64, r64 = oracle_render(g, "f64", decisions=o32.geom())
    assert (r32 == r64).all() and o32.num_rendered == o64.num_rendered
    solid = o32.fragility() > 1e-4
    assert solid.mean() > 0.999
    assert np.abs(i32 - i64).max(0)[solid].max() < 1e-4  # the north-star RGB tolerance
    assert (i32.mean() > 0.05) and (r32 > 0).all()


def test_decisions_override_is_idempotent():
    g = S.make_uniform_cloud(3000, 96, 80, seed=5, scale_med=0.03)
    o, img, radii = oracle_render(g, "f32")
    o2, img2, radii2 = oracle_render(g, "f32", decisions=o.geom())
    assert (radii == radii2).all()
    np.testing.assert_array_equal(img, img2)
    np.testing.assert_array_equal(o.binning()["point_list"], o2.binning()["point_list"])


# ---- the other half of the upstream interface: SH colours and precomputed 3D covariances -------------------------------
def _oracle_ex(g, kind, shs=None, degree=0, cov=None, decisions=None):
    o = OracleRasterizer(kind)
    img, radii = o.forward(g["means3D"], None if shs is not None else g["colors"], g["opacities"], None if cov is not None else g["scales"],
                           None if cov is not None else g["rotations"], g["view"], g["proj"], g["W"], g["H"], g["tanfovx"], g["tanfovy"], g["bg"],
                           scale_modifier=float(g.get("scale_modifier", 1.0)), decisions=decisions, shs=shs, sh_degree=degree, campos=g["campos"],
                           cov3D_precomp=cov)
    return o, img, radii


@pytest.mark.parametrize("degree,coeffs", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16), (0, 16)])
def test_sh_colours_backward_matches_independent_autograd(degree, coeffs):
    """dL/dsh and the view-direction term of dL/dmeans3D (hand-derived in the C oracle) equal torch.autograd through the published SH basis;
    coefficients beyond the active degree get exact zeros; a share of the colour channels sits on the max(., 0) clamp."""
    W, H, n = 48, 40, 260
    g = S.make_clamp_cloud(n, W, H, seed=20 + degree, scale_med=0.06, spread=0.1, opacity_logit=(0.0, 1.5))  # rotated + translated camera: campos != 0
    shs = S.random_shs(n, coeffs, seed=degree)
    dpix = np.random.default_rng(degree).standard_normal((3, H, W))
    o, img, radii = _oracle_ex(g, "f64", shs=shs, degree=degree)
    gr = o.backward(dpix)
    img2, radii2, gr2 = grads_ref(dict(g, shs=shs), W, H, g["tanfovx"], g["tanfovy"], dpix, sh_degree=degree)
    vis = radii > 0
    rgb = o.rgb()
    clamped = (rgb[vis] == 0.0).mean()
    print("visible %d of %d, colour channels on the clamp: %.1f %%" % (vis.sum(), n, 100 * clamped))
    assert (radii == radii2).all() and vis.sum() > n // 4 and 0.03 < clamped < 0.6
    np.testing.assert_allclose(img, img2, atol=1e-12)
    nb = (degree + 1) ** 2
    assert np.abs(gr["shs"][:, nb:]).max(initial=0.0) == 0.0 and np.abs(gr["shs"][~vis]).max(initial=0.0) == 0.0
    assert np.abs(gr["shs"][:, :nb]).max() > 0
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


def test_sh_degree_0_equals_precomputed_colours():
    """Degree 0 is direction-free: the image and every shared gradient equal those of colors_precomp = max(C0 sh0 + 0.5, 0), bit for bit."""
    W, H, n = 64, 48, 500
    g = S.make_uniform_cloud(n, W, H, seed=31, scale_med=0.05)
    g["campos"] = np.array([0.3, -0.2, -0.5], np.float32)
    shs = S.random_shs(n, 16, seed=3)
    o, img, radii = _oracle_ex(g, "f32", shs=shs, degree=0)
    col = o.rgb().copy()
    # invisible Gaussians have no evaluated colour; give them anything
    o2 = OracleRasterizer("f32")
    img2, radii2 = o2.forward(g["means3D"], col, g["opacities"], g["scales"], g["rotations"], g["view"], g["proj"], W, H, g["tanfovx"], g["tanfovy"], g["bg"])
    np.testing.assert_array_equal(img, img2)
    dpix = np.random.default_rng(5).standard_normal((3, H, W)).astype(np.float32)
    a, b = o.backward(dpix), o2.backward(dpix)
    for k in ("means3D", "means2D", "opacities", "scales", "rotations", "colors"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    want = np.float32(0.28209479177387814) * (b["colors"] * (col > 0))
    np.testing.assert_allclose(a["shs"][:, 0], want, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("seed,mod", [(41, 1.0), (42, 0.7)])
def test_precomputed_covariance_backward_matches_independent_autograd(seed, mod):
    """cov3D_precomp replaces scale + rotation: same image as the scale/rotation form of the same covariances (fp64: to rounding), and
    dL/dcov3D (upstream's 6-vector: off-diagonals carry both symmetric positions) equals autograd through a symmetric matrix built from it."""
    W, H, n = 56, 44, 300
    g = S.make_clamp_cloud(n, W, H, seed=seed, scale_med=0.07, spread=0.3)
    g["scale_modifier"] = mod
    cov = S.covariances_from(g["scales"], g["rotations"], mod).astype(np.float64)
    dpix = np.random.default_rng(seed).standard_normal((3, H, W))
    o, img, radii = _oracle_ex(g, "f64", cov=cov)
    gr = o.backward(dpix)
    o0, img0, radii0 = oracle_render(g, "f64")
    assert (radii == radii0).all() and (radii > 0).sum() > n // 5
    np.testing.assert_allclose(img, img0, atol=1e-6)  # cov was rounded to fp32 by the helper
    img2, radii2, gr2 = grads_ref(dict(g, cov3D_precomp=cov), W, H, g["tanfovx"], g["tanfovy"], dpix, scale_modifier=mod)
    assert (radii == radii2).all()
    np.testing.assert_allclose(img, img2, atol=1e-12)
    assert np.abs(gr["scales"]).max() == 0.0 and np.abs(gr["rotations"]).max() == 0.0
    for k in ("means3D", "means2D", "colors", "opacities", "cov3D_precomp"):
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


def test_sh_and_precomputed_covariance_together():
    W, H, n = 40, 40, 200
    g = S.make_clamp_cloud(n, W, H, seed=51, scale_med=0.08, spread=0.2)
    cov = S.covariances_from(g["scales"], g["rotations"]).astype(np.float64)
    shs = S.random_shs(n, 16, seed=9)
    dpix = np.random.default_rng(9).standard_normal((3, H, W))
    o, img, radii = _oracle_ex(g, "f64", shs=shs, degree=3, cov=cov)
    gr = o.backward(dpix)
    img2, radii2, gr2 = grads_ref(dict(g, shs=shs, cov3D_precomp=cov), W, H, g["tanfovx"], g["tanfovy"], dpix, sh_degree=3)
    assert (radii == radii2).all()
    np.testing.assert_allclose(img, img2, atol=1e-12)
    for k in ("means3D", "means2D", "shs", "opacities", "cov3D_precomp"):
        np.testing.assert_allclose(gr[k], gr2[k], rtol=1e-9, atol=1e-10 * max(1.0, np.abs(gr2[k]).max()), err_msg=k)


def test_mark_visible_restatement_agrees_with_the_near_cull_of_the_oracle():
    """oracle.gsr_oracle.mark_visible (upstream K10) against the C oracle's own near cull: every rendered Gaussian is visible, every Gaussian behind
    the 0.2 plane has radius 0, and the fp32 neighbours of 0.2 fall on the documented side (z <= 0.2 is culled)."""
    from gps_gaussian_amd import synthetic as S
    from oracle.gsr_oracle import OracleRasterizer, mark_visible
    g = S.make_uniform_cloud(4000, 160, 120, seed=2, scale_med=0.02, z_range=(0.05, 3.0), behind_frac=0.3)
    o = OracleRasterizer("f32")
    _, radii = o.forward(g["means3D"], g["colors"], g["opacities"], g["scales"], g["rotations"], g["view"], g["proj"], g["W"], g["H"], g["tanfovx"],
                         g["tanfovy"], g["bg"])
    vis = mark_visible(g["means3D"], g["view"])
    assert vis[radii > 0].all() and (radii[~vis] == 0).all() and 0.1 < vis.mean() < 0.95
    z = np.float32(0.2) + np.arange(-2, 3, dtype=np.float32) * np.float32(np.spacing(np.float32(0.2)))
    pts = np.stack([np.zeros(5, np.float32), np.zeros(5, np.float32), z], 1)
    np.testing.assert_array_equal(mark_visible(pts, np.eye(4, dtype=np.float32).reshape(-1)), [False, False, False, True, True])
