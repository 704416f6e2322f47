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
    build_host_shim(verbose)
    return os.path.join(PACKAGE_DIR, "lib", "libgpsgs_hip.so")


def build_host_shim(verbose=False):
    """Compile csrc/host_shim.cpp into lib/_gpsgs_fast.so: the compiled per-call host path of the drop-in rasteriser module (a torch C++ autograd
    Function over the same C-ABI; no device code).  g++ against this interpreter's torch headers; rebuilt when the source, the header or the
    library is newer."""
    import subprocess
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(PACKAGE_DIR, "csrc", "host_shim.cpp")
    out = os.path.join(PACKAGE_DIR, "lib", "_gpsgs_fast.so")
    deps = [src, os.path.join(os.path.dirname(PACKAGE_DIR), "include", "gpsgs.h"), os.path.join(PACKAGE_DIR, "lib", "libgpsgs_hip.so")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out, "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-DTORCH_EXTENSION_NAME=_gpsgs_fast", "-DTORCH_API_INCLUDE_EXTENSION_H"]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", "-L" + os.path.join(PACKAGE_DIR, "lib"), "-lgpsgs_hip",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building _gpsgs_fast.so failed")
    return out
