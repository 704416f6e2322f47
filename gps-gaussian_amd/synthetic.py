"""Deterministic synthetic stereo-human scenes for parity tests and bench.py (no dataset, no checkpoint offline).

Geometry follows the reference's own data generator and model ranges:
  * camera rig: circle of radius 2.0 m about look-at (0, 0.85, 0), pitch -8 deg, 22.5 deg between the two source
    cameras, fx = fy = 0.8*res, cx = res/2, cy = res/2 + 25*res/1024
    (/root/reference/prepare_data/render_data.py:113-114,130,153-160,214-216);
  * novel camera: SLERP/lerp between the sources exactly as lib/utils.py:9-55 (get_novel_calib), projection as
    lib/graphics_utils.py:31-48, znear/zfar 0.01/100 (config/stereo_human_config.py:25-26);
  * points: per source pixel inverse depth -> xyz with the depth2pc formula (lib/utils.py:88-110);
  * per-Gaussian attributes inside the ranges the regression heads can emit
    (lib/gs_parm_network.py:39-50,73,76): unit quaternion, scale in (0, 0.01], opacity in (0,1), rgb in [0,1].
Nothing here imports the reference; tests/golden/ pins the camera maths against it.
"""
import math

import numpy as np
from scipy.spatial.transform import Rotation as Rot
from scipy.spatial.transform import Slerp

SEED = 1314  # train_stage2.py:206-207


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def source_camera(res, angle_deg, radius=2.0, pitch_deg=-8.0, look_at=(0.0, 0.85, 0.0)):
    """World->camera [3,4] extrinsic (x right, y down, z forward) and [3,3] intrinsic of one rig camera."""
    look_at = np.asarray(look_at, np.float64)
    pos = look_at + _rot_y(math.radians(angle_deg)) @ _rot_x(math.radians(pitch_deg)) @ np.array([0, 0, radius])
    f = look_at - pos
    f /= np.linalg.norm(f)
    up = np.array([0.0, 1.0, 0.0])
    d = -(up - up.dot(f) * f)
    d /= np.linalg.norm(d)
    r = np.cross(d, f)
    R = np.stack([r, d, f])
    extr = np.concatenate([R, (-R @ pos)[:, None]], 1).astype(np.float32)
    s = res / 1024.0
    intr = np.array([[0.8 * res, 0, 0.5 * res], [0, 0.8 * res, 0.5 * res + 25.0 * s], [0, 0, 1]], np.float32)
    return intr, extr


def projection_matrix(znear, zfar, K, h, w):
    """Off-centre pinhole projection with z in [0, 1] and w_clip = z_view, straight from the intrinsics:
    x_ndc = (2 fx x / z + (2 cx - w)) / w,  y_ndc = (2 fy y / z + (2 cy - h)) / h  (what lib/graphics_utils.py:31-48 evaluates through
    the near-plane frustum bounds; tests/golden/camera_golden.npz pins the agreement)."""
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    P = np.zeros((4, 4), np.float64)
    P[0] = (2.0 * fx / w, 0.0, (2.0 * cx - w) / w, 0.0)
    P[1] = (0.0, 2.0 * fy / h, (2.0 * cy - h) / h, 0.0)
    P[2] = (0.0, 0.0, zfar / (zfar - znear), -zfar * znear / (zfar - znear))
    P[3] = (0.0, 0.0, 1.0, 0.0)
    return P.astype(np.float32)


def novel_camera(intr0, extr0, intr1, extr1, ratio, width, height, znear=0.01, zfar=100.0, hr=False):
    """Novel-view camera tensors in the layout render() consumes (restates lib/utils.py:9-55).

    Returns dict(FovX, FovY, world_view_transform[4,4], full_proj_transform[4,4], camera_center[3], width, height);
    both matrices are TRANSPOSED (row-vector convention), i.e. flat memory is column-major.
    """
    rots = Rot.from_matrix(np.stack([extr0[:3, :3], extr1[:3, :3]]).astype(np.float64))
    rot = Slerp([0, 1], rots)(ratio)
    npose = np.eye(4, dtype=np.float32)
    npose[:3, :3] = rot.as_matrix()
    npose[:3, 3] = ((1.0 - ratio) * extr0 + ratio * extr1)[:3, 3]
    intr_new = ((1.0 - ratio) * intr0 + ratio * intr1).astype(np.float32)
    if hr:
        intr_new[:2] *= 2
    Rt = np.zeros((4, 4), np.float64)  # getWorld2View2 with R = extr[:3,:3]^T handed in, trans 0, scale 1
    Rt[:3, :3] = npose[:3, :3]
    Rt[:3, 3] = npose[:3, 3]
    Rt[3, 3] = 1.0
    w2c = np.linalg.inv(np.linalg.inv(Rt)).astype(np.float32)
    wvt = w2c.T.copy()
    proj = projection_matrix(znear, zfar, intr_new, height, width).T
    full = (wvt @ proj).astype(np.float32)
    center = np.linalg.inv(wvt)[3, :3].astype(np.float32)
    return dict(
        FovX=np.float32(2 * math.atan(width / (2 * intr_new[0, 0]))),
        FovY=np.float32(2 * math.atan(height / (2 * intr_new[1, 1]))),
        world_view_transform=wvt.astype(np.float32), full_proj_transform=full, camera_center=center,
        width=int(width), height=int(height), intr=intr_new, extr=npose[:3, :],
    )


def _ellipsoid_inv_depth(res, intr, extr, semi, centre):
    """Ray-cast an axis-aligned ellipsoid from a pinhole camera; returns inverse depth [res,res] (0 = miss)."""
    R, t = extr[:3, :3].astype(np.float64), extr[:3, 3].astype(np.float64)
    o = -R.T @ t
    v, u = np.meshgrid(np.arange(res) + 0.5, np.arange(res) + 0.5, indexing="ij")
    dirs_c = np.stack([(u - intr[0, 2]) / intr[0, 0], (v - intr[1, 2]) / intr[1, 1], np.ones_like(u)], -1)
    dirs_w = dirs_c @ R  # (R^T d) for every pixel; the ray parameter is camera-space z
    s = np.asarray(semi, np.float64)
    oc = (o - np.asarray(centre, np.float64)) / s
    dw = dirs_w / s
    a = (dw * dw).sum(-1)
    b = 2 * (dw * oc).sum(-1)
    c = (oc * oc).sum() - 1.0
    disc = b * b - 4 * a * c
    hit = disc > 0
    z = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0))) / (2 * a), 0.0)
    hit &= z > 0.2
    return np.where(hit, 1.0 / np.where(hit, z, 1.0), 0.0).astype(np.float32)


def depth2pc_np(inv_depth, extr, intr):
    """xyz [res*res,3] of every pixel centre from inverse depth (restates lib/utils.py:88-110, fp32)."""
    S = inv_depth.shape[0]
    y, x = np.meshgrid(np.linspace(0.5, S - 0.5, S, dtype=np.float32), np.linspace(0.5, S - 0.5, S, dtype=np.float32), indexing="ij")
    z = (1.0 / (inv_depth + np.float32(1e-8))).astype(np.float32)
    px = (x - intr[0, 2]) * z / intr[0, 0]
    py = (y - intr[1, 2]) * z / intr[1, 1]
    pts = np.stack([px, py, z], 0).reshape(3, -1).astype(np.float32)
    rot_t = extr[:3, :3].T.astype(np.float32)
    out = rot_t @ pts - rot_t @ extr[:3, 3:4].astype(np.float32)
    return out.T.astype(np.float32)


def make_stereo_sample(res=1024, n_gauss=600_000, seed=SEED, render_res=None, ratio=0.5, angle0=0.0, attributes="trained"):
    """One synthetic stereo pair in the reference's per-view map layout plus the novel camera.

    attributes = "trained": the SURVEY section 8(d) distribution (scales ~ exp(N(ln 0.002, 0.35^2)) capped at 0.01 m, opacity sigmoid(N(2, 1.5^2))).
    attributes = "untrained": what the regression heads emit with freshly initialised weights, i.e. what BASELINE configs 3 / 4 really render
    offline and what stage-2 training STARTS from: every scale at (or just under) the 0.01 m clamp of lib/gs_parm_network.py:76, opacities
    around sigmoid(0) = 0.5 -- splats of ~25 px radius at 2048^2, ~55 bins per Gaussian, per-bin lists of 1,000-3,000 entries
    (profiles/r03_full_pipeline.md measured ~30 M (Gaussian, bin) instances per view with the real networks).

    Returns dict with 'lmain'/'rmain' -> {img[3,res,res] in [-1,1], xyz[res*res,3], pts_valid[res*res] bool,
    rot_maps[4,res,res], scale_maps[3,res,res], opacity_maps[1,res,res], intr, extr} and 'novel_view' -> novel_camera().
    The ellipsoid's horizontal semi-axes are tuned so that each view has n_gauss/2 valid pixels (excess hit pixels,
    <1%, are dropped at random), so the compacted cloud has exactly n_gauss Gaussians.
    """
    rng = np.random.default_rng(seed)
    render_res = render_res or res
    cams = [source_camera(res, angle0), source_camera(res, angle0 + 22.5)]
    target = n_gauss // 2
    centre = (0.0, 0.9, 0.0)
    lo, hi = 0.02, 1.2
    for _ in range(24):  # bisection on the horizontal semi-axis
        mid = 0.5 * (lo + hi)
        n = min(int((_ellipsoid_inv_depth(res, c[0], c[1], (mid, 0.9, 0.6 * mid), centre) > 0).sum()) for c in cams)
        lo, hi = (mid, hi) if n < target else (lo, mid)
    semi = (hi, 0.9, 0.6 * hi)
    out = {}
    for name, (intr, extr) in zip(("lmain", "rmain"), cams):
        inv = _ellipsoid_inv_depth(res, intr, extr, semi, centre)
        valid = (inv > 0).reshape(-1)
        idx = np.flatnonzero(valid)
        if idx.size > target:
            drop = rng.choice(idx, idx.size - target, replace=False)
            valid[drop] = False
            inv = inv * valid.reshape(res, res)
        n_px = res * res
        q = rng.standard_normal((4, n_px)).astype(np.float32)
        q /= np.linalg.norm(q, axis=0, keepdims=True)
        if attributes == "untrained":
            scale = np.minimum(0.01, rng.uniform(0.0085, 0.0115, (3, n_px))).astype(np.float32)
            opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 0.25, (1, n_px))))).astype(np.float32)
        elif attributes == "trained":
            scale = np.minimum(0.01, np.exp(rng.normal(math.log(0.002), 0.35, (3, n_px)))).astype(np.float32)
            opac = (1.0 / (1.0 + np.exp(-rng.normal(2.0, 1.5, (1, n_px))))).astype(np.float32)
        else:
            raise ValueError("attributes must be 'trained' or 'untrained'")
        img = rng.uniform(-1.0, 1.0, (3, n_px)).astype(np.float32)
        out[name] = dict(
            img=img.reshape(3, res, res), xyz=depth2pc_np(inv, extr, intr), pts_valid=valid, depth=inv,
            rot_maps=q.reshape(4, res, res), scale_maps=scale.reshape(3, res, res),
            opacity_maps=opac.reshape(1, res, res), intr=intr, extr=extr,
        )
    (i0, e0), (i1, e1) = cams
    scale_hr = render_res // res if render_res % res == 0 and render_res >= res else None
    if scale_hr is None:
        k = render_res / res  # arbitrary render size: scale the intrinsics
        i0 = i0.copy(); i1 = i1.copy(); i0[:2] *= k; i1[:2] *= k
        out["novel_view"] = novel_camera(i0, e0, i1, e1, ratio, render_res, render_res)
    elif scale_hr == 2:
        out["novel_view"] = novel_camera(i0, e0, i1, e1, ratio, render_res, render_res, hr=True)
    else:
        i0 = i0.copy(); i1 = i1.copy(); i0[:2] *= scale_hr; i1[:2] *= scale_hr
        out["novel_view"] = novel_camera(i0, e0, i1, e1, ratio, render_res, render_res)
    return out


def compact_sample(sample):
    """Flatten + mask-compact + concat both views (numpy restatement of lib/GaussianRender.py:15-34)."""
    parts = {k: [] for k in ("means3D", "colors", "rotations", "scales", "opacities")}
    for view in ("lmain", "rmain"):
        d = sample[view]
        v = d["pts_valid"]
        parts["means3D"].append(d["xyz"][v])
        parts["colors"].append(d["img"].reshape(3, -1).T[v] * 0.5 + 0.5)
        parts["rotations"].append(d["rot_maps"].reshape(4, -1).T[v])
        parts["scales"].append(d["scale_maps"].reshape(3, -1).T[v])
        parts["opacities"].append(d["opacity_maps"].reshape(1, -1).T[v])
    return {k: np.ascontiguousarray(np.concatenate(v, 0), dtype=np.float32) for k, v in parts.items()}


def make_scene(res=1024, n_gauss=600_000, seed=SEED, render_res=None, ratio=0.5, attributes="trained"):
    """Compacted Gaussian cloud + camera for one novel view: the rasteriser's direct inputs."""
    s = make_stereo_sample(res, n_gauss, seed, render_res, ratio, attributes=attributes)
    g = compact_sample(s)
    cam = s["novel_view"]
    g.update(
        view=cam["world_view_transform"], proj=cam["full_proj_transform"], campos=cam["camera_center"],
        W=cam["width"], H=cam["height"], tanfovx=math.tan(float(cam["FovX"]) * 0.5),
        tanfovy=math.tan(float(cam["FovY"]) * 0.5), bg=np.zeros(3, np.float32),
    )
    return g


def make_uniform_cloud(n_gauss, W, H, seed=SEED, z_range=(0.5, 6.0), scale_med=0.01, fov_deg=60.0, behind_frac=0.05):
    """Stress variant: Gaussians uniform in (and slightly outside) the frustum of an identity-pose camera,
    larger scales, a fraction behind the near plane.  Exercises culling, big radii, image-border rects."""
    rng = np.random.default_rng(seed)
    fx = 0.5 * W / math.tan(math.radians(fov_deg) / 2)
    K = np.array([[fx, 0, W / 2 + 3.3], [0, fx, H / 2 - 2.1], [0, 0, 1]], np.float32)
    z = rng.uniform(z_range[0], z_range[1], n_gauss)
    nb = int(behind_frac * n_gauss)
    z[:nb] = rng.uniform(-1.0, 0.25, nb)
    u = rng.uniform(-0.15 * W, 1.15 * W, n_gauss)
    v = rng.uniform(-0.15 * H, 1.15 * H, n_gauss)
    za = np.where(np.abs(z) < 1e-3, 1e-3, z)
    xyz = np.stack([(u - K[0, 2]) * za / fx, (v - K[1, 2]) * za / fx, z], 1).astype(np.float32)
    perm = rng.permutation(n_gauss)
    xyz = xyz[perm]
    q = rng.standard_normal((n_gauss, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    wvt = np.eye(4, dtype=np.float32)
    proj = projection_matrix(0.01, 100.0, K, H, W).T
    return dict(
        means3D=xyz, colors=rng.uniform(0, 1, (n_gauss, 3)).astype(np.float32), rotations=q,
        scales=np.exp(rng.normal(math.log(scale_med), 0.6, (n_gauss, 3))).astype(np.float32),
        opacities=(1 / (1 + np.exp(-rng.normal(0.0, 2.0, (n_gauss, 1))))).astype(np.float32),
        view=wvt, proj=(wvt @ proj).astype(np.float32), campos=np.zeros(3, np.float32), W=W, H=H,
        tanfovx=W / (2 * fx), tanfovy=H / (2 * fx), bg=np.array([0.1, 0.2, 0.3], np.float32),
    )


def make_clamp_cloud(n_gauss, W, H, seed=SEED, z_range=(0.5, 4.0), scale_med=0.08, fov_deg=60.0, quat_norm=(0.5, 2.0), spread=0.6,
                     opacity_logit=(-2.0, 1.0)):
    """Stress variant for the branches the other generators leave cold: a ROTATED and TRANSLATED camera (general view matrix),
    centres spread `spread` image widths beyond every image border -- far outside the 1.3 * tanfov guard band, so the clamp of the
    view-space x/y and its zeroed gradient terms fire -- large splats (so that clamped ones still reach the image), and
    quaternions that are NOT unit length (the rasteriser must not re-normalise them)."""
    rng = np.random.default_rng(seed)
    fx = 0.5 * W / math.tan(math.radians(fov_deg) / 2)
    K = np.array([[fx, 0, W / 2 - 2.7], [0, fx, H / 2 + 1.9], [0, 0, 1]], np.float32)
    z = rng.uniform(z_range[0], z_range[1], n_gauss)
    u = rng.uniform(-spread * W, (1 + spread) * W, n_gauss)
    v = rng.uniform(-spread * H, (1 + spread) * H, n_gauss)
    xc = np.stack([(u - K[0, 2]) * z / fx, (v - K[1, 2]) * z / fx, z], 1)
    # camera pose: rotation by ~25 degrees about a tilted axis, translated off the origin
    ax = np.array([0.3, 1.0, 0.2]); ax /= np.linalg.norm(ax)
    Rcw = Rot.from_rotvec(ax * math.radians(25.0)).as_matrix()          # world -> camera rotation
    tcw = np.array([0.4, -0.25, 0.6])
    xw = (xc - tcw) @ Rcw                                              # x_c = Rcw x_w + tcw  =>  x_w = Rcw^T (x_c - tcw)
    w2c = np.eye(4); w2c[:3, :3] = Rcw; w2c[:3, 3] = tcw
    wvt = w2c.T.astype(np.float32)
    q = rng.standard_normal((n_gauss, 4))
    q *= (rng.uniform(quat_norm[0], quat_norm[1], (n_gauss, 1)) / np.linalg.norm(q, axis=1, keepdims=True))
    proj = projection_matrix(0.01, 100.0, K, H, W).T
    campos = (-Rcw.T @ tcw).astype(np.float32)
    return dict(
        means3D=xw.astype(np.float32), colors=rng.uniform(0, 1, (n_gauss, 3)).astype(np.float32), rotations=q.astype(np.float32),
        scales=np.exp(rng.normal(math.log(scale_med), 0.5, (n_gauss, 3))).astype(np.float32),
        opacities=(1 / (1 + np.exp(-rng.normal(opacity_logit[0], opacity_logit[1], (n_gauss, 1))))).astype(np.float32),
        view=wvt, proj=(wvt @ proj).astype(np.float32), campos=campos, W=W, H=H,
        tanfovx=W / (2 * fx), tanfovy=H / (2 * fx), bg=np.array([0.3, 0.1, 0.2], np.float32),
    )


def random_shs(n_gauss, coeffs=16, seed=SEED, dc_spread=1.2, band_sigma=0.35):
    """Random spherical-harmonics coefficients [n, coeffs, 3]: a DC term that puts most colours into (0, 1) and a minority below zero (so that
    the `max(colour, 0)` clamp of the SH path and its cut gradient are exercised), higher bands at `band_sigma`."""
    rng = np.random.default_rng(seed + 7919)
    sh = rng.normal(0.0, band_sigma, (n_gauss, coeffs, 3))
    sh[:, 0, :] = rng.uniform(-0.5 - dc_spread, 0.5 + dc_spread, (n_gauss, 3)) / 0.28209479177387814 * 0.5
    return sh.astype(np.float32)


def covariances_from(scales, rotations, scale_modifier=1.0):
    """[n, 6] upper triangles (xx, xy, xz, yy, yz, zz) of R diag(s)^2 R^T, the layout of `cov3D_precomp` (quaternions (w, x, y, z), used
    un-normalised like the rasteriser does)."""
    q = np.asarray(rotations, np.float64)
    s = np.asarray(scales, np.float64) * scale_modifier
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                  np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                  np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    S = np.einsum("nij,nj,nkj->nik", R, s * s, R)
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
