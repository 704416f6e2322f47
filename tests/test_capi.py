"""CPU: the C-ABI library builds/loads, exports every symbol include/gpsgs.h declares, and the host shims refuse to run
without a GPU (there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import gps_gaussian_amd
from gps_gaussian_amd import _capi


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gpsgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t|const char \*)\s*\*?\s*((?:gsr|cs|gpsgs|fl|up|cv|cu)_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    gps_gaussian_amd.build()
    declared = _declared_symbols()
    assert len(declared) >= 9
    assert sorted(_capi.SYMBOLS) == declared
    raw = C.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    lib = _capi.lib()
    assert lib.gpsgs_abi_version() == 4
    assert b"gfx950" in lib.gpsgs_build_info()


def test_python_constants_mirror_the_header():
    """Every GSR_FLAG_* / GPSGS_E_* the ctypes shim uses has the value include/gpsgs.h gives it (a flag added on one side only would
    silently select a different kernel path)."""
    src = open(os.path.join(ROOT, "include", "gpsgs.h")).read()
    flags = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(GSR_FLAG_[A-Z_]+)\s+(\d+)u\b", src)}
    assert {"GSR_FLAG_DEBUG", "GSR_FLAG_TIMING", "GSR_FLAG_NO_LARGE_SORT", "GSR_FLAG_COMPOSITE_TILES", "GSR_FLAG_NO_COLOR_GRAD", "GSR_FLAG_WAVE_PRIORITY"} <= set(flags)
    for name, value in flags.items():
        assert getattr(_capi, name) == value, name
    vals = sorted(flags.values())
    assert all(v & (v - 1) == 0 for v in vals) and len(set(vals)) == len(vals)   # distinct single bits
    assert not any(v & 0xf0 for v in vals)                                       # bits 4..7 carry the stage of GSR_FLAG_TIMING_STAGE
    errs = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(GPSGS_E_[A-Z_]+)\s*=\s*(-?\d+)", src)}
    assert errs and all(getattr(_capi, k) == v for k, v in errs.items() if hasattr(_capi, k))
    assert _capi.GPSGS_E_INTERNAL == errs["GPSGS_E_INTERNAL"]


def test_code_object_targets_gfx950():
    out = subprocess.run(["strings", "-a", _capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "gfx950" in out


def test_workspace_bytes_monotone_and_aligned():
    lib = _capi.lib()
    a = lib.gsr_workspace_bytes(1000, 256, 256, 10000)
    b = lib.gsr_workspace_bytes(1000, 256, 256, 20000)
    c = lib.gsr_workspace_bytes(2000, 256, 256, 20000)
    assert 0 < a < b < c and a % 256 == 0
    # 600k Gaussians, 1024^2, 4M instances: a few hundred MB at most
    assert lib.gsr_workspace_bytes(600000, 1024, 1024, 4_000_000) < 512 << 20
    assert lib.gsr_workspace_bytes(-1, 16, 16, 1) == 0


def test_workspace_bytes_ex_and_the_limits_of_direct_lists():
    lib = _capi.lib()
    for args in ((1000, 256, 256, 10000), (600000, 1024, 1024, 4_000_000), (0, 16, 16, 0)):
        assert lib.gsr_workspace_bytes_ex(*args, 0, 0) == lib.gsr_workspace_bytes(*args)
        assert lib.gsr_workspace_bytes_ex(*args, 0, 1) == lib.gsr_workspace_bytes_forward_only(*args)
    # direct lists: bins x capacity list entries (12 bytes each) instead of instance_capacity of them
    d = lib.gsr_workspace_bytes_ex(600000, 1024, 1024, 4_000_000, 1024, 0)
    s_ = lib.gsr_workspace_bytes(600000, 1024, 1024, 4_000_000)
    assert d % 256 == 0 and 100 << 20 < d - s_ < 200 << 20
    assert lib.gsr_direct_lists_ok(1024, 1024, 1024) == 1 and lib.gsr_direct_lists_ok(2048, 2048, 512) == 1
    assert lib.gsr_direct_lists_ok(4096, 2104, 1024) == 0      # more than 65,536 bins
    assert lib.gsr_direct_lists_ok(1024, 1024, 2048) == 0 and lib.gsr_direct_lists_ok(1024, 1024, 100) == 0 and lib.gsr_direct_lists_ok(1024, 1024, 0) == 0
    assert lib.gsr_workspace_bytes_ex(1000, 4096, 2104, 10000, 1024, 0) == 0


def test_invalid_arguments_are_rejected_without_touching_the_gpu():
    lib = _capi.lib()
    assert lib.gsr_forward(10, 0, 16, *([None] * 5), 1.0, 1.0, 1.0, *([None] * 6), 0, 0, 0, None) == _capi.GPSGS_E_INVALID
    assert lib.gsr_forward(10, 16, 16, *([None] * 5), 1.0, 1.0, 1.0, *([None] * 6), 0, 0, 0, None) == _capi.GPSGS_E_INVALID
    assert lib.cs_forward(None, None, None, 1, 2, 3, 4, 4, 7, None) == _capi.GPSGS_E_INVALID
    assert lib.cs_forward(None, None, None, 0, 2, 3, 4, 4, 0, None) == _capi.GPSGS_OK  # empty batch: nothing to do


def test_no_cpu_fallback():
    import torch
    from gps_gaussian_amd import corr, rasterizer as RZ

    rs = RZ.GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3, torch.zeros(3), False, False)
    r = RZ.GaussianRasterizer(rs)
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x)
    with pytest.raises(RuntimeError, match="GPU"):
        corr.forward(torch.zeros(1, 2, 3, 4), torch.zeros(1, 1, 2, 3), 4)


def test_dropin_import_names_resolve():
    code = ("import sys; sys.path.insert(0, %r); import diff_gaussian_rasterization as d, corr_sampler as c; "
            "assert d.GaussianRasterizationSettings._fields[0] == 'image_height' and len(d.GaussianRasterizationSettings._fields) == 12; "
            "assert callable(c.forward) and callable(c.backward); print('ok')") % gps_gaussian_amd.DROPIN_DIR
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gps-gaussian_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gsr_oracle" not in txt, os.path.join(dp, f)
                # ... nor the reference build under oracle/_ref (test infrastructure: the CALLER of the product in tests/ and tools/ only)
                assert "_ref" not in txt and "stage_ref" not in txt and "refenv" not in txt, os.path.join(dp, f)


def test_nothing_of_the_reference_travels_to_the_gpu_box():
    """The reference is Python: it may be imported in the build container (golden vectors, the CPU tests that drive its Trainer) but does not travel in any
    form.  Through round 5 a sourceless-bytecode build of it was staged under oracle/_ref for the -m gpu reference tests; the recipe is gone, the directory
    is ignored by git AND by the GPU snapshot, and no file of the repository carries bytecode."""
    assert not os.path.exists(os.path.join(ROOT, "oracle", "stage_ref.py"))
    with open(os.path.join(ROOT, ".gitignore")) as f:
        assert "oracle/_ref/" in f.read().split()
    with open(os.path.join(ROOT, ".gpurunignore")) as f:
        assert "oracle/_ref/" in f.read().split()
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files"], stdout=subprocess.PIPE, text=True).stdout.split()
    assert not [n for n in tracked if n.endswith((".pyc", ".pyo"))]


def test_list_form_policy_hysteresis_and_the_compiled_host_path_declines_what_it_does_not_take(monkeypatch):
    """CPU: (a) rasterizer._note_longest -- one bin list beyond 768 entries puts a device on scanned lists (and the large-list sorts), eight views in a
    row below 512 bring direct lists back, anything in between resets the streak; GPSGS_LISTS overrides.  (b) lib/_gpsgs_fast.so loads without a GPU,
    reports the library's ABI version, and rasterize() returns None -- "not mine, take the Python path" -- for CPU tensors instead of touching them."""
    import torch
    from gps_gaussian_amd import rasterizer as RZ
    st = dict(ratio=4.0, floor=1 << 16, pending=[])
    monkeypatch.delenv("GPSGS_LISTS", raising=False)
    assert RZ._bin_capacity(st, 1024, 1024) == RZ._DIRECT_CAP and RZ._bin_capacity(st, 4096, 2104) == 0   # (more than 65,536 bins: scanned)
    RZ._note_longest(st, 700)
    assert not st.get("big_bins", False)
    RZ._note_longest(st, 769)
    assert st["big_bins"] is True and RZ._bin_capacity(st, 1024, 1024) == 0
    for k in range(7):
        RZ._note_longest(st, 300)
    assert st["big_bins"] is True
    RZ._note_longest(st, 600)          # neither long nor short: the streak starts over
    for k in range(7):
        RZ._note_longest(st, 300)
    assert st["big_bins"] is True
    RZ._note_longest(st, 512)
    assert st["big_bins"] is False and RZ._bin_capacity(st, 1024, 1024) == RZ._DIRECT_CAP
    monkeypatch.setenv("GPSGS_LISTS", "scanned")
    assert RZ._bin_capacity(st, 1024, 1024) == 0
    monkeypatch.setenv("GPSGS_LISTS", "direct")
    RZ._note_longest(st, 5000)
    assert st["big_bins"] is True and RZ._bin_capacity(st, 1024, 1024) == RZ._DIRECT_CAP   # forced: the repair run of a too-long list still takes scanned lists
    monkeypatch.setenv("GPSGS_LISTS", "nonsense")
    with pytest.raises(ValueError):
        RZ._bin_capacity(st, 1024, 1024)
    monkeypatch.delenv("GPSGS_LISTS")
    monkeypatch.setenv("GPSGS_FAST", "0")
    assert RZ._fast_module() is False
    monkeypatch.setenv("GPSGS_FAST", "1")
    F = RZ._fast_module()
    assert F and F.abi_version() == _capi.lib().gpsgs_abi_version() and F.slots_in_flight(0) == 0
    z = torch.zeros
    assert F.rasterize(z(4, 3), z(4, 3), z(4, 3), z(4, 1), z(4, 3), z(4, 4), z(3), z(16), z(16), 8, 8, 1.0, 1.0, 1.0, 100, 0, 0, 0) is None
    assert len(F.last_header()) == 7
