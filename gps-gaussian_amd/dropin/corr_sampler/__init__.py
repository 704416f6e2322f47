"""Import-name shim: `import corr_sampler` (/root/reference/core/corr.py:6) resolves to the MI355X-native kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import submodule  # noqa: E402

sys.path.pop(0)
_c = submodule("corr")
forward = _c.forward
backward = _c.backward
__all__ = ["forward", "backward"]
