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
