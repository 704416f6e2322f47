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
                           decisions=decisions)
    return o, img, radii


def hip_render(scene, dpix=None, debug=False):
    """Runs the HIP rasteriser through the drop-in module (C-ABI underneath). Returns (img, radii, grads|None, info|None) with info = dict(ws=workspace tensor, cap=instance capacity)."""
    import torch
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import rasterizer as RZ

    dev = torch.device("cuda:0")
    names = ("means3D", "colors", "opacities", "scales", "rotations")
    t = {k: torch.from_numpy(np.ascontiguousarray(scene[k], dtype=np.float32)).to(dev).requires_grad_(dpix is not None) for k in names}
    m2 = torch.zeros_like(t["means3D"], requires_grad=dpix is not None)
    rs = RZ.GaussianRasterizationSettings(
        image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        bg=torch.from_numpy(scene["bg"]).to(dev), scale_modifier=1.0, viewmatrix=torch.from_numpy(scene["view"]).to(dev),
        projmatrix=torch.from_numpy(scene["proj"]).to(dev), sh_degree=3, campos=torch.from_numpy(scene["campos"]).to(dev),
        prefiltered=False, debug=debug)
    img, radii = RZ.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, shs=None, colors_precomp=t["colors"],
                                           opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    grads = None
    info = None
    if dpix is not None:
        fn = img.grad_fn  # grab the forward's workspace before backward frees the saved tensors
        ws = [x for x in fn.saved_tensors if x.dtype == torch.uint8][0]
        info = dict(ws=ws, cap=fn.cap)
        img.backward(torch.from_numpy(np.ascontiguousarray(dpix, dtype=np.float32)).to(dev))
        grads = {k: t[k].grad.cpu().numpy() for k in names}
        grads["means2D"] = m2.grad.cpu().numpy()
    return img.detach().cpu().numpy(), radii.cpu().numpy(), grads, info


def touched_by_fragile(oracle, thresh=1e-4):
    """Mask of Gaussians whose footprint covers a pixel that sits on a branch threshold (alpha=1/255, T=1e-4)."""
    frag = oracle.fragility()
    geom = oracle.geom()
    fy, fx = np.nonzero(frag <= thresh)
    touched = np.zeros(geom["radii"].shape[0], bool)
    for y, x in zip(fy, fx):
        touched |= (np.abs(geom["xy"][:, 0] - x) <= geom["radii"] + 1) & (np.abs(geom["xy"][:, 1] - y) <= geom["radii"] + 1) & (geom["radii"] > 0)
    return frag > thresh, touched
