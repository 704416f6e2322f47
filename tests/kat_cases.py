"""Known-answer cases derived from the algorithm's definition (SURVEY.md section 8c list, section 9).  Each case takes a
`render(scene) -> (img[3,H,W], radii[P], extra)` callable so the same cases pin the CPU oracle (CPU suite) and the
HIP path (GPU suite)."""
import math

import numpy as np

from conftest import gaussians, simple_scene

FX = 64.0


def _merge(cam, g):
    s = dict(cam)
    s.update(g)
    return s


def kat_single_isotropic(render, tol):
    """(1) one isotropic Gaussian on the optical axis: alpha = min(.99, o*exp(-r^2/(2(sigma^2+0.3)))) inside its rect."""
    W, H, z, s, o = 48, 32, 2.0, 0.05, 0.7
    cam = simple_scene(W, H, FX, cx=20.5, cy=12.5)  # mean lands on pixel (20, 12) exactly
    rgb = np.array([0.2, 0.5, 0.9], np.float32)
    img, radii, _ = render(_merge(cam, gaussians([[0, 0, z]], rgb, o, s)))
    sigma2 = (FX * s / z) ** 2 + 0.3
    radius = math.ceil(3 * math.sqrt(sigma2))
    assert radii[0] == radius
    ys, xs = np.mgrid[0:H, 0:W]
    r2 = (xs - 20.0) ** 2 + (ys - 12.0) ** 2
    alpha = np.minimum(0.99, o * np.exp(-0.5 * r2 / sigma2))
    alpha[alpha < 1 / 255] = 0
    r0x, r1x = int((20.0 - radius) / 16), int((20.0 + radius + 15) / 16)
    r0y, r1y = int((12.0 - radius) / 16), int((12.0 + radius + 15) / 16)
    in_rect = (xs // 16 >= r0x) & (xs // 16 < r1x) & (ys // 16 >= r0y) & (ys // 16 < r1y)
    alpha = alpha * in_rect
    np.testing.assert_allclose(img, rgb[:, None, None] * alpha[None], atol=tol)


def kat_two_layers_and_swap(render, tol):
    """(2) two coincident footprints: C = a1 c1 + (1-a1) a2 c2; swapping depths swaps the order."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5)
    c1, c2 = np.array([1, 0, 0], np.float32), np.array([0, 0, 1], np.float32)
    for z1, z2 in ((2.0, 3.0), (3.0, 2.0)):
        # same on-screen sigma for both: scale proportional to depth
        g = gaussians([[0, 0, z1], [0, 0, z2]], [c1, c2], [[0.6], [0.8]], [[0.05 * z1 / 2] * 3, [0.05 * z2 / 2] * 3])
        img, _, _ = render(_merge(cam, g))
        sigma2 = (FX * 0.05 / 2) ** 2 + 0.3
        a1, a2 = 0.6, 0.8
        front, back = ((a1, c1), (a2, c2)) if z1 < z2 else ((a2, c2), (a1, c1))
        expect = front[0] * front[1] + (1 - front[0]) * back[0] * back[1]
        np.testing.assert_allclose(img[:, 16, 16], expect, atol=tol)
        # one pixel off-centre
        gch = math.exp(-0.5 * 1.0 / sigma2)
        expect1 = front[0] * gch * front[1] + (1 - front[0] * gch) * back[0] * gch * back[1]
        np.testing.assert_allclose(img[:, 16, 17], expect1, atol=tol)


def kat_alpha_clamp(render, tol):
    """(3) opacity 1 -> alpha clamps at 0.99 at the centre."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5, bg=(0.0, 1.0, 0.0))
    img, _, _ = render(_merge(cam, gaussians([[0, 0, 2.0]], [1.0, 0.0, 0.0], 1.0, 0.05)))
    np.testing.assert_allclose(img[:, 16, 16], [0.99, 0.01, 0.0], atol=tol)


def kat_alpha_threshold(render, tol):
    """(4) alpha just below / above 1/255 at the centre pixel."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5)
    for o, visible in ((1 / 255 * (1 - 1e-3), False), (1 / 255 * (1 + 1e-3), True)):
        img, _, _ = render(_merge(cam, gaussians([[0, 0, 2.0]], [1.0, 1.0, 1.0], o, 0.05)))
        if visible:
            np.testing.assert_allclose(img[:, 16, 16], o, atol=tol)
        else:
            assert img[:, 16, 16].max() == 0.0
        assert img[:, 16, 17].max() == 0.0  # off-centre is always below the threshold


def kat_transmittance_stop(render, tol):
    """(5) front-to-back stop: with alpha = 0.98 the third layer would give T = 8e-6 < 1e-4 and is dropped."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5, bg=(1.0, 1.0, 1.0))
    cols = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    g = gaussians([[0, 0, 2.0], [0, 0, 2.5], [0, 0, 3.0]], cols, 0.98, [[0.05] * 3, [0.0625] * 3, [0.075] * 3])
    img, _, extra = render(_merge(cam, g))
    T2 = 0.02 * 0.02
    np.testing.assert_allclose(img[:, 16, 16], [0.98 + T2, 0.02 * 0.98 + T2, T2], atol=tol)
    if extra is not None:
        assert extra["n_contrib"][16, 16] == 2
        np.testing.assert_allclose(extra["final_T"][16, 16], T2, rtol=1e-4)


def kat_near_plane(render, tol):
    """(6) view.z <= 0.2 is culled, just beyond is kept."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5)
    g = gaussians([[0, 0, 0.19999], [0, 0, 0.2001], [0, 0, -1.0], [0, 0, 0.2]], [1, 1, 1], 0.5, 0.001)
    _, radii, _ = render(_merge(cam, g))
    assert radii[0] == 0 and radii[1] > 0 and radii[2] == 0
    if getattr(render, "fp32_decisions", True):  # exactly 0.2f is culled ('<='); in fp64, double(0.2f) > 0.2
        assert radii[3] == 0


def kat_offscreen_centre(render, tol):
    """(7) centre outside the image but footprint overlapping it."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5)
    x = (-3.0 - 16.0) * 2.0 / FX  # lands on pixel x = -3
    img, radii, _ = render(_merge(cam, gaussians([[x, 0, 2.0]], [1, 1, 1], 0.9, 0.1)))
    # off-axis: J = [[fx/z, 0, -fx x/z^2], [0, fy/z, 0]]  ->  a = s^2 (J00^2 + J02^2) + 0.3, c = s^2 J11^2 + 0.3, b = 0
    a = 0.01 * ((FX / 2.0) ** 2 + (FX * x / 4.0) ** 2) + 0.3
    c = 0.01 * (FX / 2.0) ** 2 + 0.3
    assert radii[0] == math.ceil(3 * math.sqrt(max(a, c)))
    np.testing.assert_allclose(img[0, 16, 0], 0.9 * math.exp(-0.5 * 9.0 / a), atol=tol)
    np.testing.assert_allclose(img[0, 19, 0], 0.9 * math.exp(-0.5 * (9.0 / a + 9.0 / c)), atol=tol)
    far = (-200.0 - 16.0) * 2.0 / FX  # rect clamps to an empty range -> not rendered, radius reported 0
    _, radii, _ = render(_merge(cam, gaussians([[far, 0, 2.0]], [1, 1, 1], 0.9, 0.1)))
    assert radii[0] == 0


def kat_equal_depth_index_order(render, tol):
    """(8) equal depth: lower Gaussian index composites first."""
    cam = simple_scene(32, 32, FX, cx=16.5, cy=16.5)
    g = gaussians([[0, 0, 2.0], [0, 0, 2.0]], [[1, 0, 0], [0, 0, 1]], 0.5, 0.05)
    img, _, _ = render(_merge(cam, g))
    np.testing.assert_allclose(img[:, 16, 16], [0.5, 0.0, 0.25], atol=tol)


def kat_empty(render, tol):
    """(9) P = 0 -> all-zero image, NOT the background."""
    cam = simple_scene(40, 24, FX, bg=(0.3, 0.6, 0.9))
    g = gaussians(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 1)), np.zeros((0, 3)), np.zeros((0, 4)))
    img, radii, _ = render(_merge(cam, g))
    assert img.shape == (3, 24, 40) and float(np.abs(img).max()) == 0.0 and radii.shape[0] == 0
    # but: Gaussians present and all culled -> background everywhere
    img, radii, _ = render(_merge(cam, gaussians([[0, 0, -1.0]], [1, 1, 1], 0.5, 0.01)))
    np.testing.assert_allclose(img, np.broadcast_to(np.array([0.3, 0.6, 0.9], np.float32)[:, None, None], img.shape), atol=tol)


def kat_ragged_image(render, tol):
    """(10) non-square image whose sides are not multiples of 16."""
    W, H = 37, 21
    cam = simple_scene(W, H, FX, cx=30.5, cy=18.5, bg=(0.1, 0.1, 0.1))
    img, _, _ = render(_merge(cam, gaussians([[0, 0, 2.0]], [1.0, 0.5, 0.0], 0.8, 0.08)))
    sigma2 = (FX * 0.08 / 2) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    a = np.minimum(0.99, 0.8 * np.exp(-0.5 * ((xs - 30.0) ** 2 + (ys - 18.0) ** 2) / sigma2))
    a[a < 1 / 255] = 0
    rgb = np.array([1.0, 0.5, 0.0])[:, None, None]
    np.testing.assert_allclose(img, rgb * a[None] + (1 - a[None]) * 0.1, atol=tol)


ALL = [kat_single_isotropic, kat_two_layers_and_swap, kat_alpha_clamp, kat_alpha_threshold, kat_transmittance_stop, kat_near_plane,
       kat_offscreen_centre, kat_equal_depth_index_order, kat_empty, kat_ragged_image]
