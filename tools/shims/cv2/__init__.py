"""Minimal stand-in for OpenCV (absent from the MI355X image): the reference's stage-2 TRAINING loop only calls cv2.imwrite
(train_stage2.py:123, one validation preview per evaluation); the dataset loader's cv2 calls (lib/human_loader.py) read the rendered
THuman2.0 set, which needs the real package.  Put on sys.path by tools/launch_stage2.py ONLY when the real package is missing."""
import numpy as np

IMREAD_UNCHANGED, IMREAD_COLOR, INTER_AREA, INTER_LINEAR, INTER_NEAREST = -1, 1, 3, 1, 0


def imwrite(path, img):
    from PIL import Image
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]  # BGR -> RGB
    Image.fromarray(np.ascontiguousarray(a.astype(np.uint8))).save(path)
    return True


def __getattr__(name):
    def missing(*a, **k):
        raise ImportError("cv2.%s: OpenCV is not installed in this image (tools/shims/cv2 only provides imwrite); the reference's dataset "
                          "loader needs the real opencv-python" % name)
    return missing
