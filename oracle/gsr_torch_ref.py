"""Independent fp64 torch-autograd restatement of the rasteriser, used ONLY to pin oracle/gsr_oracle.c.

TEST INFRASTRUCTURE ONLY.  Dense (every pixel x every Gaussian), so small cases only.

The C oracle carries a hand-derived backward; this file re-states just the FORWARD
(SURVEY.md sections 9.1-9.2) with torch ops and lets autograd differentiate it, with the three places where the
upstream gradient is deliberately not the autograd one encoded explicitly:
  * min(0.99, .) is straight-through (gradient passes the clamp unchanged);
  * the conic's gradient uses 1/(det^2 + 1e-7) instead of 1/det^2;
  * the clamped view-space x/y (1.3*tanfov) is a constant for the gradient;
radius / tile membership / branch decisions / depth order are constants (computed under no_grad).
"""
import math

import torch


class _Conic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gx, gy_true, gz):
        # gy_true = dL/dB for the conic's off-diagonal B (power has the term -B dx dy once).  Upstream accumulates
        # HALF of it in dL_dconic.y and compensates with the factor 2 in dL_db; the net result is below.
        a, b, c = ctx.saved_tensors
        denom = a * c - b * b
        k = 1.0 / (denom * denom + 1e-7)
        gy = 0.5 * gy_true
        da = k * (-c * c * gx + 2 * b * c * gy + (denom - a * c) * gz)
        dc = k * (-a * a * gz + 2 * a * b * gy + (denom - a * c) * gx)
        db = k * 2 * (b * c * gx - (denom + 2 * b * b) * gy + a * b * gz)
        return da, db, dc


def _xf(m, p, rows):
    # m flat[16] column-major; p [P,3]
    return torch.stack([m[r] * p[:, 0] + m[4 + r] * p[:, 1] + m[8 + r] * p[:, 2] + m[12 + r] for r in range(rows)], 1)


_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
          -0.5900435899266435)


def sh_to_rgb(shs, degree, means3D, campos):
    """Colour of every Gaussian from its real spherical-harmonics coefficients shs [P,M,3] evaluated in the direction camera -> Gaussian
    (the published basis of the 3D-Gaussian-splatting rasteriser: constants above), + 0.5, clamped at 0.  Plain torch ops: autograd
    differentiates the clamp (zero where the sum is negative), the basis and the normalisation of the direction."""
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = _SH_C0 * shs[:, 0]
    if degree > 0:
        c = c - _SH_C1 * y * shs[:, 1] + _SH_C1 * z * shs[:, 2] - _SH_C1 * x * shs[:, 3]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c = (c + _SH_C2[0] * xy * shs[:, 4] + _SH_C2[1] * yz * shs[:, 5] + _SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + _SH_C2[3] * xz * shs[:, 7] + _SH_C2[4] * (xx - yy) * shs[:, 8])
    if degree > 2:
        c = (c + _SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + _SH_C3[1] * xy * z * shs[:, 10] + _SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
             + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + _SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
             + _SH_C3[5] * z * (xx - yy) * shs[:, 14] + _SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    c = c + 0.5
    return torch.where(c < 0, torch.zeros_like(c), c)


def render_ref(means3D, colors, opacities, scales, rotations, view, proj, W, H, tanfovx, tanfovy, bg,
               scale_modifier=1.0, ndc_offset=None, shs=None, sh_degree=0, campos=None, cov3D_precomp=None):
    """All tensor inputs fp64.  Returns (image[3,H,W], radii[P]).  ndc_offset: optional zeros[P,2] leaf whose
    gradient is the upstream `dL_dmeans2D[:, :2]`.  shs [P,M,3] + sh_degree + campos replace `colors` (None), cov3D_precomp [P,6]
    (xx, xy, xz, yy, yz, zz) replaces scales + rotations (None)."""
    dt = torch.float64
    P = means3D.shape[0]
    view = view.reshape(16).to(dt)
    proj = proj.reshape(16).to(dt)
    bg = bg.to(dt)
    img = torch.zeros(3, H, W, dtype=dt)
    if P == 0:
        return img, torch.zeros(0, dtype=torch.int32)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    gx, gy = (W + 15) // 16, (H + 15) // 16

    p_view = _xf(view, means3D, 3)
    p_hom = _xf(proj, means3D, 4)
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    if ndc_offset is not None:
        ndc = ndc + ndc_offset

    if shs is not None:
        colors = sh_to_rgb(shs, sh_degree, means3D, campos.to(dt))
    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sigma = torch.stack([torch.stack([c6[:, 0], c6[:, 1], c6[:, 2]], 1), torch.stack([c6[:, 1], c6[:, 3], c6[:, 4]], 1),
                             torch.stack([c6[:, 2], c6[:, 4], c6[:, 5]], 1)], 1)
    else:
        r, x, y, z = rotations.unbind(1)
        Rm = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)  # [P,3,3]
        S2 = torch.diag_embed((scale_modifier * scales) ** 2)
        Sigma = Rm @ S2 @ Rm.transpose(1, 2)

    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], 1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], 1)], 1)  # [P,2,3]
    Rw = torch.stack([torch.stack([view[0], view[4], view[8]]), torch.stack([view[1], view[5], view[9]]),
                      torch.stack([view[2], view[6], view[10]])])
    T = J @ Rw
    cov = T @ Sigma @ T.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    A, B, C = _Conic.apply(a, b, c)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    with torch.no_grad():
        det = a * c - b * b
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
        r0x = torch.clamp(((px - radius) / 16.0).to(torch.int64), 0, gx)   # .to(int64) truncates toward zero
        r0y = torch.clamp(((py - radius) / 16.0).to(torch.int64), 0, gy)
        r1x = torch.clamp(((px + radius + 15.0) / 16.0).to(torch.int64), 0, gx)
        r1y = torch.clamp(((py + radius + 15.0) / 16.0).to(torch.int64), 0, gy)
        vis = (tz > 0.2) & (det != 0) & ((r1x - r0x) * (r1y - r0y) > 0)
        radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)
        depth32 = tz.to(torch.float32)  # order by fp32-rounded depth, ties by index (stable)
        order = torch.argsort(depth32, stable=True)
        order = order[vis[order]]
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        tix, tiy = (xs // 16).reshape(-1), (ys // 16).reshape(-1)

    pixx = xs.reshape(-1).to(dt)
    pixy = ys.reshape(-1).to(dt)
    Tt = torch.ones(H * W, dtype=dt)
    Cc = torch.zeros(H * W, 3, dtype=dt)
    done = torch.zeros(H * W, dtype=torch.bool)
    for g in order.tolist():
        with torch.no_grad():
            member = (tix >= r0x[g]) & (tix < r1x[g]) & (tiy >= r0y[g]) & (tiy < r1y[g]) & ~done
        dx = px[g] - pixx
        dy = py[g] - pixy
        power = -0.5 * (A[g] * dx * dx + C[g] * dy * dy) - B[g] * dx * dy
        raw = opacities[g] * torch.exp(power)
        alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # straight-through clamp
        with torch.no_grad():
            ev = member & (power <= 0) & (alpha >= 1.0 / 255.0)
            test_T = Tt * (1 - alpha)
            stop = ev & (test_T < 1e-4)
            done = done | stop
            use = ev & ~stop
        a_eff = torch.where(use, alpha, torch.zeros_like(alpha))
        Cc = Cc + colors[g][None, :] * (a_eff * Tt)[:, None]
        Tt = Tt * (1 - a_eff)
    out = Cc + Tt[:, None] * bg[None, :]
    return out.t().reshape(3, H, W), radii


def grads_ref(inputs, W, H, tanfovx, tanfovy, dL_dpix, scale_modifier=1.0, sh_degree=None):
    """inputs: dict of numpy/tensors (means3D, colors, opacities[P,1], scales, rotations, view, proj, bg; optionally shs [P,M,3] +
    campos -- used instead of colors when sh_degree is given -- and cov3D_precomp [P,6], used instead of scales + rotations).
    Returns (image, radii, dict of gradients named like the oracle's)."""
    dt = torch.float64
    use_sh = sh_degree is not None
    use_cov = inputs.get("cov3D_precomp") is not None
    t = {k: torch.as_tensor(v).to(dt).clone() for k, v in inputs.items() if k in
         ("means3D", "colors", "opacities", "scales", "rotations", "view", "proj", "bg", "shs", "campos", "cov3D_precomp") and v is not None}
    t["opacities"] = t["opacities"].reshape(-1)
    leaves = ["means3D", "shs" if use_sh else "colors", "opacities"] + (["cov3D_precomp"] if use_cov else ["scales", "rotations"])
    for k in leaves:
        t[k].requires_grad_(True)
    off = torch.zeros(t["means3D"].shape[0], 2, dtype=dt, requires_grad=True)
    img, radii = render_ref(t["means3D"], None if use_sh else t["colors"], t["opacities"], None if use_cov else t["scales"],
                            None if use_cov else t["rotations"], t["view"], t["proj"], W, H, tanfovx, tanfovy, t["bg"], scale_modifier, off,
                            shs=t["shs"] if use_sh else None, sh_degree=sh_degree or 0, campos=t.get("campos"),
                            cov3D_precomp=t["cov3D_precomp"] if use_cov else None)
    loss = (img * torch.as_tensor(dL_dpix).to(dt)).sum()
    if loss.requires_grad:
        gs = torch.autograd.grad(loss, [t[k] for k in leaves] + [off], allow_unused=True)
    else:
        gs = [None] * (len(leaves) + 1)
    out = {}
    for k, g in zip(leaves, gs[:-1]):
        out[k] = (torch.zeros_like(t[k]) if g is None else g).detach().numpy()
    out["opacities"] = out["opacities"].reshape(-1, 1)
    g2 = torch.zeros_like(off) if gs[-1] is None else gs[-1]
    out["means2D"] = torch.cat([g2, torch.zeros(g2.shape[0], 1, dtype=dt)], 1).detach().numpy()
    return img.detach().numpy(), radii.numpy(), out
