"""Locates the gps_gaussian_amd package from inside the drop-in import shims (this directory is what users put on
sys.path / PYTHONPATH so that the reference's `import diff_gaussian_rasterization` / `import corr_sampler` resolve)."""
import importlib
import importlib.util
import os
import sys


def package():
    if "gps_gaussian_amd" in sys.modules:
        return sys.modules["gps_gaussian_amd"]
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gps_gaussian_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gps_gaussian_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def submodule(name):
    package()
    mod = importlib.import_module("gps_gaussian_amd." + name)
    if os.environ.get("GPSGS_ACCELERATE"):
        # opt-in (default off): rebind the reference's pts2render / l1_loss / ssim / CorrBlockFast1D / upsample_flow / flow2depth / depth2pc to
        # the fused kernels as the reference's modules load -- accelerate.py; no reference file is touched
        importlib.import_module("gps_gaussian_amd.accelerate").install()
    return mod
