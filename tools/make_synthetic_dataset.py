#!/usr/bin/env python
"""A synthetic stand-in for the rendered THuman2.0 set, in the ON-DISK layout the reference's loader reads (harness, not product).

    python tools/make_synthetic_dataset.py OUT [--res 1024] [--train 4] [--val 2] [--seed 1314]

Layout (lib/human_loader.py:109-115, written by prepare_data/render_data.py:60-85 in the reference's pipeline), per phase directory
OUT/{train,val}/ and per sample name:
    img/<name>/<cam>.jpg          RGB, res x res                    (cameras 0, 1 = the stereo pair; 2, 3, 4 = novel views between them)
    img/<name>/<cam>_hr.jpg       RGB, 2 res x 2 res, novel cameras (config/stage2.yaml:15 `use_hr_img`)
    mask/<name>/<cam>.png         3-channel 0 / 255
    depth/<name>/<cam>.png        16-bit, inverse depth x 2^15      (lib/human_loader.py:92-93)
    parm/<name>/<cam>_intrinsic.npy [3,3], <cam>_extrinsic.npy [3,4] float64
The "human" is the analytic ellipsoid of gps_gaussian_amd.synthetic with a smooth procedural texture painted on its surface in world
space (so the two source views are photo-consistent and the novel views are a valid target); the rig is the reference generator's:
radius 2.0 m about (0, 0.85, 0), pitch -8 degrees, 22.5 degrees between the sources, fx = fy = 0.8 res, cy offset 25 px at 1024
(prepare_data/render_data.py:113-160,205-216).  OUT/val doubles as a `--test_data_root` for test_view_interp.py (it needs img/, mask/, parm/).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _texture(xyz, phase):
    """Smooth colour field over world space, [N,3] in [0,1]."""
    f = np.array([[9.0, 5.0, 7.0], [4.0, 11.0, 6.0], [8.0, 3.0, 10.0]])
    s = np.sin(xyz.astype(np.float64) @ f.T + phase)
    return 0.5 + 0.45 * s


def _render(S, res, intr, extr, semi, centre, phase, hr=False):
    if hr:
        intr = intr.copy()
        intr[:2] *= 2
        res = res * 2
    inv = S._ellipsoid_inv_depth(res, intr, extr, semi, centre)
    hit = inv > 0
    xyz = S.depth2pc_np(inv, extr, intr)
    rgb = (_texture(xyz, phase) * 255.0).reshape(res, res, 3)
    img = np.where(hit[..., None], rgb, 0.0).astype(np.uint8)
    return img, hit, inv


def make_dataset(out, res=1024, n_train=4, n_val=2, seed=1314, fill=0.28, quiet=False):
    """fill: fraction of a source image the human covers (0.28 -> ~290k valid pixels per 1024^2 view, P ~ 580k: BASELINE config 2's size)."""
    from PIL import Image

    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import synthetic as S

    rng = np.random.default_rng(seed)
    for phase, n in (("train", n_train), ("val", n_val)):
        for k in range(n):
            name = "%04d_%03d" % (k, int(rng.integers(0, 16)))
            base = float(rng.uniform(0.0, 360.0))
            u1, u3 = float(rng.uniform()), float(rng.uniform())
            angles = [base, base + 22.5, base + u1 * 11.25, base + 11.25, base + 22.5 - u3 * 11.25]
            b_px = 0.9 * 0.8 * res / 2.0                      # projected half height of the 1.8 m ellipsoid at the rig radius
            a = fill * res * res / (np.pi * b_px) * 2.0 / (0.8 * res) * float(rng.uniform(0.95, 1.05))
            semi = (a, 0.9, 0.6 * a)
            centre = (float(rng.uniform(-0.05, 0.05)), 0.9, float(rng.uniform(-0.05, 0.05)))
            tex_phase = rng.uniform(0, 6.28, 3)
            d = os.path.join(out, phase)
            for sub in ("img", "mask", "depth", "parm"):
                os.makedirs(os.path.join(d, sub, name), exist_ok=True)
            for cam, ang in enumerate(angles):
                intr, extr = S.source_camera(res, ang)
                img, hit, inv = _render(S, res, intr, extr, semi, centre, tex_phase)
                Image.fromarray(img).save(os.path.join(d, "img", name, "%d.jpg" % cam), quality=95)
                Image.fromarray(np.repeat((hit * 255).astype(np.uint8)[..., None], 3, -1)).save(os.path.join(d, "mask", name, "%d.png" % cam))
                Image.fromarray(np.clip(inv * 2.0 ** 15, 0, 65535).astype(np.uint16)).save(os.path.join(d, "depth", name, "%d.png" % cam))
                np.save(os.path.join(d, "parm", name, "%d_intrinsic.npy" % cam), intr.astype(np.float64))  # float64 like the taichi exporter: lib/graphics_utils.py:41 assigns K-derived numpy scalars into a torch tensor
                np.save(os.path.join(d, "parm", name, "%d_extrinsic.npy" % cam), extr.astype(np.float64))
                if cam >= 2:
                    img_hr, _, _ = _render(S, res, intr, extr, semi, centre, tex_phase, hr=True)
                    Image.fromarray(img_hr).save(os.path.join(d, "img", name, "%d_hr.jpg" % cam), quality=95)
            if not quiet:
                print("make_synthetic_dataset: %s/%s" % (phase, name), flush=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--train", type=int, default=4)
    ap.add_argument("--val", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1314)
    ap.add_argument("--fill", type=float, default=0.28)
    a = ap.parse_args()
    make_dataset(a.out, a.res, a.train, a.val, a.seed, a.fill)
