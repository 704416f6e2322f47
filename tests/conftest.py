import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_list_policy(request):
    """GPU tests: which list form a view gets (direct bin lists / scanned lists) follows what the device has rendered before (rasterizer._note_longest).
    Every test starts from the same state -- "no long list seen" -- so that its path does not depend on the tests that ran before it."""
    if request.node.get_closest_marker("gpu") is not None:
        from gps_gaussian_amd import rasterizer as RZ
        for st in list(RZ._state.values()):
            for k in ("big_bins", "short_streak", "longest"):
                st.pop(k, None)
    yield


def simple_scene(W, H, fx, cx=None, cy=None, bg=(0.0, 0.0, 0.0)):
    """Identity-pose pinhole camera in the layout render() consumes + empty Gaussian lists to fill in."""
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import synthetic as S

    cx = W / 2 if cx is None else cx
    cy = H / 2 if cy is None else cy
    K = np.array([[fx, 0, cx], [0, fx, cy], [0, 0, 1]], np.float32)
    view = np.eye(4, dtype=np.float32)
    proj = (view @ S.projection_matrix(0.01, 100.0, K, H, W).T).astype(np.float32)
    return dict(view=view, proj=proj, campos=np.zeros(3, np.float32), W=W, H=H, tanfovx=W / (2 * fx), tanfovy=H / (2 * fx),
                bg=np.asarray(bg, np.float32), K=K)


def gaussians(xyz, rgb, opacity, scale, quat=None):
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    rgb = np.broadcast_to(np.asarray(rgb, np.float32), (n, 3)).copy()
    opacity = np.broadcast_to(np.asarray(opacity, np.float32).reshape(-1, 1), (n, 1)).copy()
    scale = np.broadcast_to(np.asarray(scale, np.float32), (n, 3)).copy()
    quat = np.broadcast_to(np.asarray([1, 0, 0, 0] if quat is None else quat, np.float32), (n, 4)).copy()
    return dict(means3D=xyz, colors=rgb, opacities=opacity, scales=scale, rotations=quat)


def oracle_render(scene, kind="f32", decisions=None):
    from oracle.gsr_oracle import OracleRasterizer

    o = OracleRasterizer(kind)
    img, radii = o.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                           scene["view"], scene["proj"], scene["W"], scene["H"], scene["tanfovx"], scene["tanfovy"], scene["bg"],
                           scale_modifier=float(scene.get("scale_modifier", 1.0)), decisions=decisions)
    return o, img, radii


def hip_render(scene, dpix=None, debug=False, shs=None, sh_degree=3, cov3D_precomp=None):
    """Runs the HIP rasteriser through the drop-in module (C-ABI underneath). Returns (img, radii, grads|None, info|None) with info = dict(ws=workspace tensor, cap=instance capacity).
    shs [P,M,3] (+ sh_degree): SH colours INSTEAD of scene["colors"]; cov3D_precomp [P,6]: INSTEAD of scales + rotations (the gradient dict then
    carries "shs" / "cov3D_precomp" in place of "colors" / "scales" + "rotations")."""
    import torch
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import rasterizer as RZ

    dev = torch.device("cuda:0")
    src = dict(scene)
    names = ["means3D", "opacities"]
    if shs is None:
        names.append("colors")
    else:
        names.append("shs"); src["shs"] = shs
    if cov3D_precomp is None:
        names += ["scales", "rotations"]
    else:
        names.append("cov3D_precomp"); src["cov3D_precomp"] = cov3D_precomp
    t = {k: torch.from_numpy(np.ascontiguousarray(src[k], dtype=np.float32)).to(dev).requires_grad_(dpix is not None) for k in names}
    m2 = torch.zeros_like(t["means3D"], requires_grad=dpix is not None)
    rs = RZ.GaussianRasterizationSettings(
        image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        bg=torch.from_numpy(scene["bg"]).to(dev), scale_modifier=float(scene.get("scale_modifier", 1.0)), viewmatrix=torch.from_numpy(scene["view"]).to(dev),
        projmatrix=torch.from_numpy(scene["proj"]).to(dev), sh_degree=sh_degree, campos=torch.from_numpy(scene["campos"]).to(dev),
        prefiltered=False, debug=debug)
    RZ._debug_keep_ws = True  # both host paths (compiled / Python) then leave the forward's workspace for export_state
    try:
        img, radii = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, shs=t.get("shs"), colors_precomp=t.get("colors"),
                                               opacities=t["opacities"], scales=t.get("scales"), rotations=t.get("rotations"),
                                               cov3D_precomp=t.get("cov3D_precomp"))
    finally:
        RZ._debug_keep_ws = False
    grads = None
    info = None
    last = RZ._tls.__dict__.pop("last_ws", None)
    if dpix is not None:
        info = last
        img.backward(torch.from_numpy(np.ascontiguousarray(dpix, dtype=np.float32)).to(dev))
        grads = {k: t[k].grad.cpu().numpy() for k in names}
        grads["means2D"] = m2.grad.cpu().numpy()
    return img.detach().cpu().numpy(), radii.cpu().numpy(), grads, info


def clamp_active(scene):
    """Mask of Gaussians whose view-space x/z or y/z lies outside the 1.3 * tanfov guard band (the EWA clamp and its zeroed gradient
    terms are active for them)."""
    w2c = np.asarray(scene["view"], np.float64).T
    pc = np.asarray(scene["means3D"], np.float64) @ w2c[:3, :3].T + w2c[:3, 3]
    z = np.where(np.abs(pc[:, 2]) < 1e-12, 1e-12, pc[:, 2])
    return (np.abs(pc[:, 0] / z) > 1.3 * scene["tanfovx"]) | (np.abs(pc[:, 1] / z) > 1.3 * scene["tanfovy"])


def fragile_bounds(oracle, dpix=None, thresh=1e-5, power_band=1e-6):
    """-> (solid [H,W] bool, touched [P] bool, bounds dict | None).
    solid: pixels farther than `thresh` (relative) from every branch threshold (alpha = 1/255, T = 1e-4 and -- unless power_band is 0 -- an exponent
    within power_band of upstream's `power > 0` skip).  touched: Gaussians that TAKE PART in a fragile pixel -- alpha >= 1/255 there, inside their tile
    rect, in front of a clear stop -- i.e. the only Gaussians a flipped branch decision can move (round 5: until then every Gaussian whose
    radius + 1 bounding SQUARE held a fragile pixel was excused, which under large splats was most of the cloud).  bounds[k][g]: how far a flip at
    those pixels can move gradient k of Gaussian g (oracle/gsr_oracle.c flip_bound: the pair's own contribution if it is the one that flips,
    ~alpha_f of it for a bystander), zero for untouched Gaussians.  Without dpix: touched only (bounds None)."""
    frag = oracle.fragility(power_band)
    H, W = frag.shape
    touched, bounds = oracle.flip_bound(np.ones((3, H, W), np.float32) if dpix is None else dpix, thresh=thresh, power_band=power_band)
    return frag > thresh, touched, (bounds if dpix is not None else None)


def touched_by_fragile(oracle, thresh=1e-5, power_band=1e-6):
    """(solid, touched) of fragile_bounds() for callers that only look at the image."""
    solid, touched, _ = fragile_bounds(oracle, None, thresh, power_band)
    return solid, touched


def _allowed(og_k, bound_k, grad_tol):
    """Per-element error budget: the ordinary tolerance (relative to the element + grad_tol of the tensor's largest element) plus twice what flipped
    branch decisions at the Gaussian's fragile pixels can explain (zero for an untouched Gaussian)."""
    s_ = np.abs(og_k).max() + 1e-30
    a = grad_tol * np.abs(og_k) + grad_tol * grad_tol * s_
    # (the normalised error e = |d| / (|ref| + tol * s) <= tol  <=>  |d| <= tol |ref| + tol^2 s)
    if bound_k is not None:
        a = a + 2.0 * bound_k.reshape(a.shape)
    return a


def parity_report(name, img, oimg, grads, og, solid, touched, extra=None, rgb_tol=1e-4, grad_tol=1e-3, visible=None, bounds=None):
    """What SURVEY.md section 7 ("Hard parts") asks every parity test to REPORT, not just assert: max abs RGB error and the number of
    pixels over tolerance (all pixels / pixels away from a branch threshold), and per gradient array the max normalised error
    |a - ref| / (|ref| + tol * max|ref|) over Gaussians that take part in no fragile pixel AND over those that do, the largest error of a touched
    Gaussian relative to its budget (tolerance + 2 x flip bound; must stay <= 1), the 99.9 % quantile over all elements, and the
    number of Gaussians with any element over tolerance.  Printed (pytest -s) and appended to gpurun_out/parity_report.jsonl."""
    import json
    err = np.abs(img - oimg).max(0)
    rep = dict(test=name, pixels=int(err.size), fragile_pixels=int((~solid).sum()), rgb_max_err=float(err.max()),
               rgb_max_err_solid=float(err[solid].max()) if solid.any() else 0.0, pixels_over_tol=int((err > rgb_tol).sum()),
               gaussians=int(touched.size), gaussians_touching_fragile=int(touched.sum()))
    if visible is not None:
        # the strict gradient check covers the visible Gaussians that take part in no fragile pixel: how much of the cloud is that?
        rep["visible_gaussians"] = int(visible.sum())
        rep["strict_set_fraction"] = float((visible & ~touched).sum() / max(1, int(visible.sum())))
    if grads is not None:
        rep["grads"] = {}
        for k in grads:
            s_ = np.abs(og[k]).max() + 1e-30
            d = np.abs(grads[k] - og[k])
            e = d / (np.abs(og[k]) + grad_tol * s_)
            over = (e > grad_tol).reshape(e.shape[0], -1).any(axis=-1)
            row = dict(max_err_untouched=float(e[~touched].max()) if (~touched).any() else 0.0,
                       max_err_touched=float(e[touched].max()) if touched.any() else 0.0, q999=float(np.quantile(e, 0.999)),
                       fraction_over_tol=float(over.mean()),
                       gaussians_over_tol=int(over.sum()), gaussians_over_tol_untouched=int((over & ~touched).sum()))
            if bounds is not None and touched.any():
                row["max_err_touched_over_budget"] = float((d / _allowed(og[k], bounds.get(k), grad_tol))[touched].max())
            rep["grads"][k] = row
    if extra:
        rep.update(extra)
    line = json.dumps(rep)
    print(line)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(line + "\n")
    return rep


def assert_grad_parity(grads, og, touched, visible, grad_tol=1e-3, global_frac=2e-3, strict_min=0.5, strict_max_over=0, bounds=None, strict_cap=None,
                       excused_cap=3.0):
    """The gradient criterion every parity test applies (so that every one of them can FAIL):
      * STRICT: the visible Gaussians that take part in no fragile pixel (fragile_bounds) are within grad_tol (normalised error
        |a - ref| / (|ref| + grad_tol max|ref|)), at most strict_max_over of them excepted -- and an excepted one by at most excused_cap x grad_tol;
      * CAPPED (round 5; needs `bounds`): a Gaussian that does take part in a fragile pixel may exceed grad_tol only by what flipped branch
        decisions at those pixels can explain: |a - ref| <= tolerance + 2 x its flip bound, element by element -- never "anything finite".  At most
        strict_max_over of them may exceed that budget, and (round 6) NONE by more than excused_cap x the budget: until then an excused Gaussian
        was excused without any limit;
      * GLOBAL: over ALL Gaussians, the fraction with any element off by more than grad_tol stays below global_frac (>= 2 allowed: tiny clouds);
      * invisible Gaussians receive exactly zero;
      * the STRICT set must be at least strict_min of the visible cloud (round 6: an assertion, it used to be a printed remark) -- a test whose scene
        cannot reach 0.5 passes its own, justified, strict_min.
    Returns the strict-set fraction."""
    nvis = int(visible.sum())
    strict = visible & ~touched
    frac = float(strict.sum() / max(1, nvis))
    for k in grads:
        assert np.isfinite(grads[k]).all(), k
        s_ = np.abs(og[k]).max() + 1e-30
        d = np.abs(grads[k] - og[k])
        e = (d / (np.abs(og[k]) + grad_tol * s_)).reshape(d.shape[0], -1)
        over = (e > grad_tol).any(axis=-1)
        assert np.abs(grads[k][~visible]).max(initial=0.0) == 0.0, k
        assert int(over.sum()) <= max(2, int(global_frac * over.size)), "%s: %d of %d Gaussians over %g (strict set %.3f of the visible cloud)" % (
            k, int(over.sum()), over.size, grad_tol, frac)
        assert int((over & strict).sum()) <= strict_max_over, "%s: %d strict-set Gaussians over %g (max err %.3e)" % (
            k, int((over & strict).sum()), grad_tol, e[strict].max())
        if strict.any():
            cap_ = excused_cap if strict_cap is None else min(strict_cap, excused_cap)
            assert e[strict].max() <= cap_ * grad_tol, "%s: a strict-set Gaussian is off by %.3e (cap %g x %g)" % (k, e[strict].max(), cap_, grad_tol)
        if bounds is not None:
            ratio = (d / _allowed(og[k], bounds.get(k), grad_tol)).reshape(d.shape[0], -1).max(axis=-1)
            bad = (ratio > 1.0) & touched
            assert int(bad.sum()) <= strict_max_over, "%s: %d Gaussians that take part in a fragile pixel are off by more than a flipped decision there can explain " \
                "(worst: %.2f x its budget of tolerance + 2 x flip bound)" % (k, int(bad.sum()), float(ratio[touched].max()))
            if touched.any():
                assert float(ratio[touched].max()) <= excused_cap, "%s: an excused Gaussian is off by %.2f x its budget (cap %g): no flipped decision explains that" % (
                    k, float(ratio[touched].max()), excused_cap)
    assert frac >= strict_min, "the strict set is %.3f of the visible cloud (< %.2f): the test leans on the flip budget alone" % (frac, strict_min)
    return frac
