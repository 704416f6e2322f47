"""Stand-in for OpenCV (absent from the MI355X image), HARNESS ONLY: exactly the calls the reference's stage-2 scripts make --
cv2.imwrite (train_stage2.py:123, test_view_interp.py:47), and what lib/human_loader.py needs to read a rendered data set and rectify a
stereo pair on the fly (imread :93, stereoRectify :273, initUndistortRectifyMap / remap :292-298, :68-69, erode :319) -- written against
OpenCV's documented behaviour for the zero-distortion pinhole case the loader uses (dist0 = dist1 = zeros, :271), with numpy / scipy /
PIL.  Put on sys.path by tools/launch_stage2.py and tools/run_reference.py ONLY when the real package is missing; not part of the
product (the data-set loader is outside the hot path)."""
import numpy as np

IMREAD_UNCHANGED, IMREAD_COLOR, IMREAD_GRAYSCALE = -1, 1, 0
INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3
CV_32FC1, CV_16SC2 = 5, 11
CALIB_ZERO_DISPARITY = 1024


def imwrite(path, img):
    from PIL import Image
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]  # BGR -> RGB
    Image.fromarray(np.ascontiguousarray(a.astype(np.uint8))).save(path)
    return True


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    try:
        im = Image.open(path)
    except (FileNotFoundError, OSError):
        return None  # OpenCV's behaviour
    if flags == IMREAD_UNCHANGED:
        a = np.array(im)  # 16-bit PNG -> uint16 (mode "I;16"), 8-bit stays uint8
        if a.dtype == np.int32:
            a = a.astype(np.uint16)
        return a[:, :, ::-1].copy() if a.ndim == 3 and a.shape[2] >= 3 else a
    if flags == IMREAD_GRAYSCALE:
        return np.array(im.convert("L"))
    return np.array(im.convert("RGB"))[:, :, ::-1].copy()


def _rodrigues(v):
    from scipy.spatial.transform import Rotation
    v = np.asarray(v, np.float64)
    if v.shape == (3, 3):
        return Rotation.from_matrix(v).as_rotvec()
    return Rotation.from_rotvec(v.reshape(3)).as_matrix()


def stereoRectify(cameraMatrix1, distCoeffs1, cameraMatrix2, distCoeffs2, imageSize, R, T, R1=None, R2=None, P1=None, P2=None, Q=None,
                  flags=CALIB_ZERO_DISPARITY, alpha=-1, newImageSize=(0, 0)):
    """Bouguet's rectification as cv::stereoRectify documents it, for undistorted pinhole cameras and alpha < 0 (no rescaling):
    both cameras are rotated by half of R, then about the axis that brings the baseline onto the x (or y) axis; the common focal
    length is the smaller of the two, the principal points keep the image corners centred (averaged along the epipolar direction's
    normal, or in both directions with CALIB_ZERO_DISPARITY).  -> (R1, R2, P1, P2, Q, roi1, roi2)"""
    K1, K2 = np.asarray(cameraMatrix1, np.float64), np.asarray(cameraMatrix2, np.float64)
    for d in (distCoeffs1, distCoeffs2):
        if d is not None and np.any(np.asarray(d) != 0):
            raise NotImplementedError("cv2 stand-in: stereoRectify with lens distortion")
    if alpha is not None and alpha >= 0:
        raise NotImplementedError("cv2 stand-in: stereoRectify with alpha >= 0")
    R = np.asarray(R, np.float64).reshape(3, 3)
    T = np.asarray(T, np.float64).reshape(3)
    om = _rodrigues(R)
    r_r = _rodrigues(-0.5 * om)
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3)
    uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0:
        ww *= np.arccos(min(1.0, abs(c) / nt)) / nw
    wR = _rodrigues(ww)
    Ra, Rb = wR @ r_r.T, wR @ r_r
    t = Rb @ T
    nx, ny = int(imageSize[0]), int(imageSize[1])
    fc = min(K1[idx ^ 1, idx ^ 1], K2[idx ^ 1, idx ^ 1])
    cc = []
    for K, Rk in ((K1, Ra), (K2, Rb)):
        corners = np.array([[0, 0], [nx - 1, 0], [0, ny - 1], [nx - 1, ny - 1]], np.float64)
        n = np.stack([(corners[:, 0] - K[0, 2]) / K[0, 0], (corners[:, 1] - K[1, 2]) / K[1, 1], np.ones(4)], 1) @ Rk.T
        avg = (fc * n[:, :2] / n[:, 2:]).mean(0)
        cc.append(np.array([(nx - 1) / 2 - avg[0], (ny - 1) / 2 - avg[1]]))
    if flags & CALIB_ZERO_DISPARITY:
        cc[0] = cc[1] = 0.5 * (cc[0] + cc[1])
    else:
        m = 0.5 * (cc[0][idx ^ 1] + cc[1][idx ^ 1])
        cc[0][idx ^ 1] = cc[1][idx ^ 1] = m
    Pa, Pb = np.zeros((3, 4)), np.zeros((3, 4))
    for P_, c_ in ((Pa, cc[0]), (Pb, cc[1])):
        P_[0, 0] = P_[1, 1] = fc
        P_[0, 2], P_[1, 2], P_[2, 2] = c_[0], c_[1], 1.0
    Pb[idx, 3] = t[idx] * fc
    Qm = np.array([[1, 0, 0, -cc[0][0]], [0, 1, 0, -cc[0][1]], [0, 0, 0, fc],
                   [0, 0, -1.0 / t[idx], (cc[0][idx] - cc[1][idx]) / t[idx]]], np.float64)
    roi = (0, 0, nx, ny)
    return Ra, Rb, Pa, Pb, Qm, roi, roi


def initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type=CV_32FC1):
    """For every pixel (u, v) of the rectified image: the source pixel it samples, (map_x, map_y) float32 [h, w] (no distortion)."""
    if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
        raise NotImplementedError("cv2 stand-in: initUndistortRectifyMap with lens distortion")
    K = np.asarray(cameraMatrix, np.float64)
    P = np.asarray(newCameraMatrix, np.float64)
    w, h = int(size[0]), int(size[1])
    iR = np.linalg.inv(P[:3, :3] @ np.asarray(R, np.float64).reshape(3, 3))
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[0, 0] * u + iR[0, 1] * v + iR[0, 2]
    Y = iR[1, 0] * u + iR[1, 1] * v + iR[1, 2]
    Wc = iR[2, 0] * u + iR[2, 1] * v + iR[2, 2]
    return (K[0, 0] * X / Wc + K[0, 2]).astype(np.float32), (K[1, 1] * Y / Wc + K[1, 2]).astype(np.float32)


def remap(src, map1, map2, interpolation=INTER_LINEAR, borderMode=0, borderValue=0):
    """dst(v, u) = src(map2(v, u), map1(v, u)), bilinear (or nearest), constant-zero border."""
    from scipy import ndimage
    a = np.asarray(src)
    order = 0 if interpolation == INTER_NEAREST else 1
    coords = np.stack([np.asarray(map2, np.float64), np.asarray(map1, np.float64)])

    def one(ch):
        return ndimage.map_coordinates(ch.astype(np.float32), coords, order=order, mode="constant", cval=float(borderValue), prefilter=False)

    out = one(a) if a.ndim == 2 else np.stack([one(a[:, :, i]) for i in range(a.shape[2])], -1)
    if np.issubdtype(a.dtype, np.integer):
        info = np.iinfo(a.dtype)
        out = np.clip(np.rint(out), info.min, info.max)
    return out.astype(a.dtype)


def erode(src, kernel, dst=None, anchor=None, iterations=1, **_):
    from scipy import ndimage
    if not isinstance(iterations, int):  # the reference passes `1` positionally into the dst slot (lib/human_loader.py:319)
        iterations = 1
    a = np.asarray(src)
    fp = np.asarray(kernel) != 0
    for _i in range(max(1, iterations)):
        a = ndimage.minimum_filter(a, footprint=fp, mode="nearest")
    return a


def __getattr__(name):
    def missing(*a, **k):
        raise ImportError("cv2.%s: OpenCV is not installed in this image (tools/shims/cv2 provides only what the reference's stage-2 "
                          "scripts call)" % name)
    return missing
