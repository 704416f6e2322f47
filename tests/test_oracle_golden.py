"""CPU: the oracle / host-side restatements against fixtures produced by the REFERENCE's own code (tests/golden/make_golden.py)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

import gps_gaussian_amd  # noqa: F401
from gps_gaussian_amd import synthetic as S
from oracle import gsr_oracle


def _aux():
    gsr_oracle.build()
    return C.CDLL(os.path.join(os.path.dirname(gsr_oracle.__file__), "_build", "libaux_oracle.so"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_novel_camera_matches_reference_get_novel_calib():
    g = np.load(os.path.join(GOLDEN, "camera_golden.npz"))
    for hr in (0, 1):
        for ratio in (0.1, 0.5, 0.9):
            side = 2048 if hr else 1024
            cam = S.novel_camera(g["intr0"], g["extr0"], g["intr1"], g["extr1"], ratio, side, side, hr=bool(hr))
            tag = "hr%d_r%02d_" % (hr, int(ratio * 10))
            for k in ("FovX", "FovY", "world_view_transform", "full_proj_transform", "camera_center"):
                np.testing.assert_allclose(np.asarray(cam[k]), g[tag + k], rtol=2e-6, atol=2e-6, err_msg=tag + k)


def test_depth2pc_matches_reference():
    g = np.load(os.path.join(GOLDEN, "depth2pc_golden.npz"))
    xyz = S.depth2pc_np(g["inv_depth"], g["extr"], g["intr"])
    # invalid pixels (inverse depth 0) unproject to ~1e8 m: compare relatively
    np.testing.assert_allclose(xyz, g["xyz"], rtol=2e-5, atol=2e-5 * np.abs(g["xyz"]).max())


def test_corr_sampler_oracle_matches_reference_corrblock1d():
    g = np.load(os.path.join(GOLDEN, "corr_sampler_golden.npz"))
    aux = _aux()
    r = int(g["radius"])
    rd = 2 * r + 1
    coords = g["coords"][:, 0]
    for lvl in range(4):
        vol = np.ascontiguousarray(g["volume%d" % lvl])
        N, H1, W1, W2 = vol.shape
        c = np.ascontiguousarray(coords / 2 ** lvl, np.float32)
        out = np.zeros((N, rd, H1, W1), np.float32)
        aux.cs_oracle_forward(_p(vol), _p(c), _p(out), N, H1, W1, W2, r)
        ref = g["out"][:, lvl * rd:(lvl + 1) * rd]
        # grid_sample normalises/un-normalises the coordinate (rounding ~1e-6 px) -> 1e-5 * |volume| tolerance
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5 * np.abs(vol).max(), err_msg="level %d fwd" % lvl)
        go = np.ascontiguousarray(g["grad_out"][:, lvl * rd:(lvl + 1) * rd])
        gv = np.zeros_like(vol)
        aux.cs_oracle_backward(_p(c), _p(go), _p(gv), N, H1, W1, W2, r)
        np.testing.assert_allclose(gv, g["grad_volume%d" % lvl], rtol=0, atol=2e-5 * np.abs(go).max(), err_msg="level %d bwd" % lvl)


def test_compaction_matches_reference_pts2render():
    g = np.load(os.path.join(GOLDEN, "pts2render_golden.npz"))
    B = g["lmain_img"].shape[0]
    for i in range(B):
        sample = {v: dict(img=g[v + "_img"][i], xyz=g[v + "_xyz"][i], pts_valid=g[v + "_pts_valid"][i], rot_maps=g[v + "_rot_maps"][i],
                          scale_maps=g[v + "_scale_maps"][i], opacity_maps=g[v + "_opacity_maps"][i]) for v in ("lmain", "rmain")}
        c = S.compact_sample(sample)
        for mine, theirs in (("means3D", "xyz"), ("colors", "rgb"), ("rotations", "rot"), ("scales", "scale"), ("opacities", "opacity")):
            np.testing.assert_array_equal(c[mine], g["out%d_%s" % (i, theirs)])


def test_zsplat_oracle_sequential_semantics():
    aux = _aux()
    res = 8
    pts = np.array([[[2.2, 3.9, 0.5, 1, 0, 0], [2.7, 3.1, 0.8, 0, 1, 0], [2.0, 3.0, 0.8, 0, 0, 1], [-5, 100, 0.1, 1, 1, 1],
                     [4, 4, 0.9, 1, 1, 0]]], np.float32)
    mask = np.array([[1, 1, 1, 1, 0]], np.float32)
    depth = np.zeros((1, res, res), np.float32)
    color = -np.ones((1, 3, res, res), np.float32)
    aux.zsplat_oracle(_p(pts), _p(mask), _p(depth), _p(color), 1, pts.shape[1], res)
    assert depth[0, 3, 2] == np.float32(0.8) and tuple(color[0, :, 3, 2]) == (0, 0, 1)   # nearest wins; later tie wins
    assert depth[0, 7, 0] == np.float32(0.1) and tuple(color[0, :, 7, 0]) == (1, 1, 1)   # clamped to the border
    assert depth[0, 4, 4] == 0 and tuple(color[0, :, 4, 4]) == (-1, -1, -1)              # masked point ignored


def test_loss_oracle_matches_reference_l1_and_ssim_values_and_gradients():
    """oracle/loss_oracle.py (fp64 numpy) vs values/gradients produced by the reference's own lib/loss.py."""
    from oracle import loss_oracle as L
    g = np.load(os.path.join(GOLDEN, "loss_golden.npz"))
    s, gr = L.ssim(g["pred"], g["gt"], with_grad=True)
    assert abs(L.l1(g["pred"], g["gt"]) - float(g["l1"])) < 1e-7
    assert abs(s - float(g["ssim"])) < 1e-6
    np.testing.assert_allclose(gr, g["grad_ssim"], rtol=0, atol=1e-5 * np.abs(g["grad_ssim"]).max())
    np.testing.assert_allclose(np.sign(g["pred"].astype(np.float64) - g["gt"]) / g["pred"].size, g["grad_l1"], atol=1e-9)


# ---- row f4: correlation volume + pyramid, multi-level lookup, convex upsampling -------------------------------------------------
def test_corr_pyramid_oracle_matches_reference_golden():
    from oracle import corr_oracle as CO
    z = np.load(os.path.join(GOLDEN, "corr_pyramid_golden.npz"))
    pyr = CO.build_pyramid(z["fmap1"], z["fmap2"], 4)
    for l in range(4):
        assert pyr[l].shape == z["pyr%d" % l].shape
        np.testing.assert_allclose(pyr[l], z["pyr%d" % l], rtol=2e-5, atol=2e-5)   # the golden is fp32 torch
    g1, g2 = CO.build_pyramid_backward(z["fmap1"], z["fmap2"], [z["grad_pyr%d" % l] for l in range(4)])
    np.testing.assert_allclose(g1, z["grad_fmap1"], rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(g2, z["grad_fmap2"], rtol=2e-5, atol=5e-5)


def test_multilevel_lookup_oracle_matches_reference_golden():
    from oracle import corr_oracle as CO
    z = np.load(os.path.join(GOLDEN, "corr_sampler_golden.npz"))
    vols = [z["volume%d" % l] for l in range(4)]
    out = CO.lookup(vols, z["coords"], int(z["radius"]))
    np.testing.assert_allclose(out, z["out"], rtol=1e-5, atol=1e-5)
    gv = CO.lookup_backward([v.shape[-1] for v in vols], z["coords"], z["grad_out"], int(z["radius"]))
    for l in range(4):
        np.testing.assert_allclose(gv[l], z["grad_volume%d" % l], rtol=1e-5, atol=1e-5)


def test_upsample_oracle_matches_reference_golden():
    from oracle import corr_oracle as CO
    z = np.load(os.path.join(GOLDEN, "upsample_golden.npz"))
    f = int(z["factor"])
    np.testing.assert_allclose(CO.upsample_flow(z["flow"], z["mask"], f), z["out"], rtol=1e-5, atol=1e-5)
    gf, gm = CO.upsample_flow_backward(z["flow"], z["mask"], z["grad_out"], f)
    np.testing.assert_allclose(gf, z["grad_flow"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gm, z["grad_mask"], rtol=1e-5, atol=2e-5)
