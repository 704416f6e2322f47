"""CPU: the gradient parity criterion itself (tests/conftest.py::fragile_bounds / assert_grad_parity, oracle/gsr_oracle.c flip_bound).
A criterion is only worth something if (a) it passes what it must forgive -- a correct evaluation that takes the other branch at a fragile
pixel -- and (b) it FAILS on a localized bug, also on a Gaussian that sits under a fragile pixel (until round 5 such
Gaussians were excused without any limit)."""
import numpy as np
import pytest

from conftest import assert_grad_parity, fragile_bounds, oracle_render

import gps_gaussian_amd  # noqa: F401
from gps_gaussian_amd import synthetic as S


def _scene(which):
    if which == "cloud":
        return S.make_uniform_cloud(20000, 333, 277, seed=3, scale_med=0.02), 1e-4, 1e-6
    if which == "big_splats":
        return S.make_uniform_cloud(3000, 96, 80, seed=5, scale_med=0.08), 1e-4, 1e-6
    return S.make_scene(1024, 600000), 1e-4, 2e-6   # BASELINE config 2


@pytest.mark.parametrize("which", ["cloud", "big_splats", "config2"])
def test_real_branch_flips_stay_inside_the_budget_and_a_localized_bug_does_not(which):
    g, thresh, eps = _scene(which)
    H, W = g["H"], g["W"]
    dpix = np.random.default_rng(11).standard_normal((3, H, W)).astype(np.float32)
    o, img, radii = oracle_render(g, "f32")
    og = o.backward(dpix)
    solid, touched, bounds = fragile_bounds(o, dpix, thresh=thresh)
    vis = radii > 0
    # (a) a second correct evaluation: opacities moved by eps << thresh -- the smooth change of every gradient is ~eps, but the pairs whose
    #     alpha sits within eps of 1/255 (and the stops within the amplified eps of 1e-4) take the other branch
    g2 = dict(g, opacities=(g["opacities"].astype(np.float64) * (1 + eps)).astype(np.float32))
    o2, img2, _ = oracle_render(g2, "f32")
    og2 = o2.backward(dpix)
    flipped = int((o.binning()["n_contrib"] != o2.binning()["n_contrib"]).sum()) + int((np.abs(img - img2).max(0) > 1e-4).sum())
    frac = assert_grad_parity(og2, og, touched, vis, bounds=bounds)
    print("%s: %d fragile pixels, %d pixels really flipped, %d Gaussians touched, strict set %.3f" % (which, int((~solid).sum()), flipped, int(touched.sum()), frac))
    assert flipped > 0 and touched.sum() > 0 and frac > 0.5
    moved = max(float((np.abs(og2[k] - og[k]) / (1e-3 * np.abs(og[k]) + 1e-6 * np.abs(og[k]).max())).max()) for k in og)
    assert moved > 1.0, "the perturbation flipped nothing that shows in a gradient: the test would prove nothing"
    # (b) the same evaluation with ONE touched Gaussian corrupted by a few times its own budget: must fail, whichever tensor
    cand = np.nonzero(touched & vis)[0]
    for k in ("means3D", "opacities", "scales"):
        sc = np.abs(og[k]).max()
        budget = 1e-3 * np.abs(og[k][cand]) + 1e-6 * sc + 2 * bounds[k][cand]
        i = cand[np.argmax(budget.max(-1))]                       # the most forgiving case: the Gaussian with the LARGEST budget
        bad = {kk: v.copy() for kk, v in og2.items()}
        j = int(np.argmax(budget[np.nonzero(cand == i)[0][0]]))
        bad[k][i, j] = og[k][i, j] + 4.0 * float(budget.max())
        with pytest.raises(AssertionError):
            assert_grad_parity(bad, og, touched, vis, bounds=bounds)
    # (c) VERDICT r05 weak 2a: with strict_max_over = 16 (what the large-splat test passes) FIVE touched Gaussians off by 10 x their budget used to be
    #     excused without any limit; an excused Gaussian is now capped at excused_cap (3) x its budget
    k = "means3D"
    sc = np.abs(og[k]).max()
    bad = {kk: v.copy() for kk, v in og2.items()}
    for i in cand[:5]:
        budget_i = 1e-3 * np.abs(og[k][i]) + 1e-6 * sc + 2 * bounds[k][i]
        j = int(np.argmax(budget_i))
        bad[k][i, j] = og[k][i, j] + 10.0 * float(budget_i[j])
    with pytest.raises(AssertionError, match="excused Gaussian"):
        assert_grad_parity(bad, og, touched, vis, bounds=bounds, strict_max_over=16, global_frac=1.0)
    # ... the same five off by 2 x their budget ARE what strict_max_over = 16 is there to excuse
    ok2 = {kk: v.copy() for kk, v in og2.items()}
    for i in cand[:5]:
        budget_i = 1e-3 * np.abs(og[k][i]) + 1e-6 * sc + 2 * bounds[k][i]
        j = int(np.argmax(budget_i))
        ok2[k][i, j] = og[k][i, j] + 2.0 * float(budget_i[j])
    assert_grad_parity(ok2, og, touched, vis, bounds=bounds, strict_max_over=16, global_frac=1.0)
    with pytest.raises(AssertionError):
        assert_grad_parity(ok2, og, touched, vis, bounds=bounds, strict_max_over=4, global_frac=1.0)
    # (d) VERDICT r05 weak 2b: a strict set below strict_min fails instead of printing a remark
    with pytest.raises(AssertionError, match="strict set"):
        assert_grad_parity(og2, og, touched, vis, bounds=bounds, strict_min=1.01)
    # ... and an untouched Gaussian off by 2e-3 of its value fails the strict part
    un = vis & ~touched
    i = np.nonzero(un)[0][np.argmax(np.abs(og["means3D"]).max(-1)[un])]
    bad = {kk: v.copy() for kk, v in og2.items()}
    bad["means3D"][i] = og["means3D"][i] * 1.004 + 4e-6 * np.abs(og["means3D"]).max()
    with pytest.raises(AssertionError):
        assert_grad_parity(bad, og, touched, vis, bounds=bounds)


def test_touched_means_contribution_not_bounding_square():
    """Screen-filling splats: the old rule (radius + 1 bounding square around a fragile pixel) excused most of the cloud; contribution does not."""
    g = S.make_uniform_cloud(3000, 96, 80, seed=5, scale_med=0.12)
    dpix = np.random.default_rng(1).standard_normal((3, 80, 96)).astype(np.float32)
    o, _, radii = oracle_render(g, "f32")
    solid, touched, bounds = fragile_bounds(o, dpix, thresh=1e-4)
    geom = o.geom()
    fy, fx = np.nonzero(~solid)
    old = np.zeros(radii.shape[0], bool)
    for y, x in zip(fy, fx):
        old |= (np.abs(geom["xy"][:, 0] - x) <= geom["radii"] + 1) & (np.abs(geom["xy"][:, 1] - y) <= geom["radii"] + 1) & (geom["radii"] > 0)
    vis = radii > 0
    print("fragile pixels %d: old rule excuses %.3f of the visible cloud, contribution rule %.3f" % (len(fy), old[vis].mean(), touched[vis].mean()))
    assert len(fy) > 0 and (touched & ~old).sum() == 0 and touched[vis].mean() < 0.5 * old[vis].mean()
    for k, b in bounds.items():
        assert (b[~touched] == 0).all() and np.isfinite(b).all(), k
