"""Import-name shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(/root/reference/gaussian_renderer/__init__.py:14) resolves to the MI355X-native implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import submodule  # noqa: E402

sys.path.pop(0)
_r = submodule("rasterizer")
GaussianRasterizationSettings = _r.GaussianRasterizationSettings
GaussianRasterizer = _r.GaussianRasterizer
rasterize_gaussians = _r.rasterize_gaussians
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
