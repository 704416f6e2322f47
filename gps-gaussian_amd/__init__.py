"""gps_gaussian_amd -- MI355X-native render hot path of GPS-Gaussian.

Layout:
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/gpsgs.h) -> lib/libgpsgs_hip.so
  _capi.py     ctypes binding of that C-ABI (the only place the .so is touched)
  rasterizer.py  GaussianRasterizationSettings / GaussianRasterizer (drop-in for `diff_gaussian_rasterization`)
  corr.py      forward / backward (drop-in for `corr_sampler`)
  render_api.py  render(data, idx, ...) / pts2render(data, bg_color) with the reference's signatures
  dropin/      import-name shims: put this directory on sys.path and the reference's scripts run unmodified
  synthetic.py deterministic synthetic stereo-human scenes (tests / bench)

There is no CPU fallback: every op raises if the HIP library is missing or the tensors are not on a GPU.
"""
import os

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
DROPIN_DIR = os.path.join(PACKAGE_DIR, "dropin")
__version__ = "0.1.0"


def build(verbose=False):
    """Compile csrc/ into lib/libgpsgs_hip.so with hipcc for gfx950 (works without a GPU)."""
    import subprocess

    cmd = ["make", "-C", os.path.join(PACKAGE_DIR, "csrc"), "-j8", "all"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libgpsgs_hip.so failed")
    return os.path.join(PACKAGE_DIR, "lib", "libgpsgs_hip.so")
