"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the reference's correlation volume + pyramid, multi-level
lookup and convex upsampling.  Nothing in the product path imports this file; only tests/, __graft_entry__.smoke() and bench
baselines may.

Follows (paths relative to /root/reference):
  build_pyramid / build_pyramid_backward   core/corr.py:31-42 (CorrBlockFast1D.__init__) and :53-61 (.corr)
  lookup / lookup_backward                 core/corr.py:44-51 (__call__) with the tap semantics of CorrBlock1D, core/corr.py:127-146
  upsample_flow / upsample_flow_backward   core/raft_stereo_human.py:69-81

Pinned by tests/test_oracle_golden.py against tests/golden/{corr_pyramid,corr_sampler,upsample}_golden.npz, which were produced
by running the reference's own PyTorch code (tests/golden/make_golden.py)."""
import numpy as np


def build_pyramid(f1, f2, levels=4):
    """f1[N,D,H,W1], f2[N,D,H,W2] -> list of [N,H,W1,W2>>l]."""
    f1 = np.asarray(f1, np.float64)
    f2 = np.asarray(f2, np.float64)
    D = f1.shape[1]
    corr = np.einsum("aijk,aijh->ajkh", f1, f2) / np.sqrt(np.float64(D))
    out = [corr]
    for _ in range(1, levels):
        w = out[-1].shape[-1] // 2
        out.append(0.5 * (out[-1][..., 0:2 * w:2] + out[-1][..., 1:2 * w:2]))  # avg_pool2d([1,2], stride [1,2]) floors
    return out


def _unpool_exact(g, widths, l):
    """Un-pool level-l gradient to level 0 respecting the floor at every level."""
    g = np.asarray(g, np.float64)
    for lv in range(l, 0, -1):
        up = np.zeros(g.shape[:-1] + (widths[lv - 1],), np.float64)
        w = widths[lv]
        up[..., 0:2 * w:2] = 0.5 * g
        up[..., 1:2 * w:2] = 0.5 * g
        g = up
    return g


def build_pyramid_backward(f1, f2, grads):
    f1 = np.asarray(f1, np.float64)
    f2 = np.asarray(f2, np.float64)
    D, W2 = f1.shape[1], f2.shape[3]
    widths = [W2 >> l for l in range(len(grads))]
    g0 = np.zeros((f1.shape[0], f1.shape[2], f1.shape[3], W2), np.float64)
    for l, g in enumerate(grads):
        if g is not None:
            g0 += _unpool_exact(g, widths, l)
    s = 1.0 / np.sqrt(np.float64(D))
    gf1 = np.einsum("ajkh,aijh->aijk", g0, f2) * s
    gf2 = np.einsum("ajkh,aijk->aijh", g0, f1) * s
    return gf1, gf2


def lookup(pyr, coords, radius):
    """pyr[l][N,H,W1,W2>>l], coords[N,1,H,W1] -> [N, L*(2r+1), H, W1]: linear interpolation, zero outside the row."""
    c = np.asarray(coords, np.float64)[:, 0]
    N, H, W1 = c.shape
    rd = 2 * radius + 1
    out = np.zeros((N, len(pyr) * rd, H, W1), np.float64)
    for l, v in enumerate(pyr):
        v = np.asarray(v, np.float64)
        wl = v.shape[-1]
        x0 = c / (2.0 ** l)
        fl = np.floor(x0)
        dx = x0 - fl
        for k in range(rd):
            xa = (fl - radius + k).astype(np.int64)
            xb = xa + 1
            va = np.where((xa >= 0) & (xa < wl), np.take_along_axis(v, np.clip(xa, 0, max(wl - 1, 0))[..., None], -1)[..., 0], 0.0) if wl else 0.0
            vb = np.where((xb >= 0) & (xb < wl), np.take_along_axis(v, np.clip(xb, 0, max(wl - 1, 0))[..., None], -1)[..., 0], 0.0) if wl else 0.0
            out[:, l * rd + k] = va * (1.0 - dx) + vb * dx
    return out


def lookup_backward(widths, coords, grad_out, radius):
    """-> list of grad volumes [N,H,W1,w] for w in widths."""
    c = np.asarray(coords, np.float64)[:, 0]
    g = np.asarray(grad_out, np.float64)
    N, H, W1 = c.shape
    rd = 2 * radius + 1
    res = []
    ii = np.indices((N, H, W1))
    for l, wl in enumerate(widths):
        gv = np.zeros((N, H, W1, wl), np.float64)
        x0 = c / (2.0 ** l)
        fl = np.floor(x0)
        dx = x0 - fl
        for k in range(rd):
            xa = (fl - radius + k).astype(np.int64)
            for xs, wgt in ((xa, 1.0 - dx), (xa + 1, dx)):
                ok = (xs >= 0) & (xs < wl)
                np.add.at(gv, (ii[0][ok], ii[1][ok], ii[2][ok], xs[ok]), (g[:, l * rd + k] * wgt)[ok])
        res.append(gv)
    return res


def _softmax9(mask, f):
    N, _, H, W = mask.shape
    m = np.asarray(mask, np.float64).reshape(N, 9, f, f, H, W)
    m = m - m.max(axis=1, keepdims=True)
    e = np.exp(m)
    return e / e.sum(axis=1, keepdims=True)


def _taps(flow, f):
    N, C, H, W = flow.shape
    pad = np.zeros((N, C, H + 2, W + 2), np.float64)
    pad[:, :, 1:-1, 1:-1] = f * np.asarray(flow, np.float64)
    return np.stack([pad[:, :, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], axis=2)  # [N,C,9,H,W], unfold order


def upsample_flow(flow, mask, f):
    N, C, H, W = flow.shape
    p = _softmax9(mask, f)                      # [N,9,f,f,H,W]
    t = _taps(flow, f)                          # [N,C,9,H,W]
    up = np.einsum("nkijhw,nckhw->ncijhw", p, t)
    return up.transpose(0, 1, 4, 2, 5, 3).reshape(N, C, f * H, f * W)


def upsample_flow_backward(flow, mask, grad_out, f):
    N, C, H, W = flow.shape
    p = _softmax9(mask, f)
    t = _taps(flow, f)
    g = np.asarray(grad_out, np.float64).reshape(N, C, H, f, W, f).transpose(0, 1, 3, 5, 2, 4)   # [N,C,i,j,H,W]
    u = np.einsum("ncijhw,nckhw->nkijhw", g, t)                 # d out / d p
    dot = (p * u).sum(axis=1, keepdims=True)
    gmask = (p * (u - dot)).reshape(N, 9 * f * f, H, W)
    gt = np.einsum("ncijhw,nkijhw->nckhw", g, p)                # gradient of every (shifted, scaled) tap plane
    gpad = np.zeros((N, C, H + 2, W + 2), np.float64)
    k = 0
    for ky in range(3):
        for kx in range(3):
            gpad[:, :, ky:ky + H, kx:kx + W] += gt[:, :, k]
            k += 1
    return f * gpad[:, :, 1:-1, 1:-1], gmask
