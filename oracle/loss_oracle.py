"""fp64 numpy restatement of the reference's L1 + SSIM losses.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/lib/loss.py:36-37 (l1_loss) and :40-83 (gaussian window sigma 1.5, 11x11, zero padding, C1 = 0.01^2,
C2 = 0.03^2, mean over everything).  PINNED: tests/golden/loss_golden.npz holds values and autograd gradients produced by
the reference's own code (tests/golden/make_golden.py)."""
import numpy as np
from scipy.signal import correlate2d


def window(size=11, sigma=1.5):
    g = np.array([np.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], np.float32)
    g = (g / g.sum()).astype(np.float32)          # the reference normalises in fp32 ...
    return np.outer(g, g).astype(np.float32).astype(np.float64)  # ... and forms the 2-D window as an fp32 outer product


def _filt(x, w):
    return np.stack([np.stack([correlate2d(x[b, c], w, mode="same", boundary="fill") for c in range(x.shape[1])]) for b in range(x.shape[0])])


def l1(img1, img2):
    return np.abs(img1.astype(np.float64) - img2.astype(np.float64)).mean()


def ssim(img1, img2, with_grad=False):
    x1, x2 = img1.astype(np.float64), img2.astype(np.float64)
    w = window()
    mu1, mu2 = _filt(x1, w), _filt(x2, w)
    e11, e22, e12 = _filt(x1 * x1, w), _filt(x2 * x2, w), _filt(x1 * x2, w)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    A1, A2, B1, B2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2, mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    smap = A1 * A2 / (B1 * B2)
    if not with_grad:
        return smap.mean()
    # d ssim_map / d(mu1 | sigma fixed), d/d sigma1^2, d/d sigma12, then total derivative w.r.t. x1 through the three filters
    ds_dmu1 = 2 * mu2 * A2 / (B1 * B2) - 2 * mu1 * A1 * A2 / (B1 * B1 * B2)
    ds_ds1 = -A1 * A2 / (B1 * B2 * B2)
    ds_ds12 = 2 * A1 / (B1 * B2)
    M1 = ds_dmu1 - 2 * mu1 * ds_ds1 - mu2 * ds_ds12
    g = _filt(M1, w) + 2 * x1 * _filt(ds_ds1, w) + x2 * _filt(ds_ds12, w)   # window is symmetric: correlation = its adjoint
    return smap.mean(), g / smap.size
