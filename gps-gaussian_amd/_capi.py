"""ctypes binding of include/gpsgs.h.  No fallback: a missing library is an ImportError with build instructions."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPSGS_LIB") or os.path.join(_HERE, "lib", "libgpsgs_hip.so")  # GPSGS_LIB: development (another build of the same library)

# every symbol include/gpsgs.h declares (tests/test_capi_symbols.py cross-checks this list against the header)
SYMBOLS = (
    "gpsgs_abi_version", "gpsgs_build_info", "gpsgs_measure_sclk", "gsr_debug_set_wg_trace", "gsr_workspace_bytes", "gsr_workspace_bytes_forward_only", "gsr_workspace_bytes_ex", "gsr_direct_lists_ok", "gsr_forward", "gsr_forward_notify", "gsr_forward_ex", "gsr_backward", "gsr_backward_ex", "gsr_copy_header_async", "gsr_read_header",
    "gsr_export_state", "gsr_mark_visible", "gsr_selftest", "gsr_timing_read", "gsr_debug_count_records", "gsr_pack_scratch_bytes", "gsr_pack_views", "gsr_pack_views_backward", "fl_scratch_bytes",
    "fl_l1_ssim_forward", "fl_l1_ssim_backward", "up_unproject_forward", "up_unproject_backward", "up_unproject_forward_dev", "up_unproject_backward_dev", "cs_forward", "cs_backward",
    "cv_build_forward", "cv_build_backward", "cs_lookup_forward", "cs_lookup_backward", "cu_upsample_forward", "cu_upsample_backward", "cu_upsample_scratch_bytes",
)

GPSGS_OK, GPSGS_E_INVALID, GPSGS_E_WORKSPACE, GPSGS_E_LAUNCH, GPSGS_E_NO_DEVICE, GPSGS_E_INTERNAL = 0, -1, -2, -3, -4, -5
_ERR = {-1: "invalid argument", -2: "workspace too small", -3: "HIP launch failed", -4: "no HIP device",
        -5: "internal self-check failed (debug mode: inconsistent bin lists)"}
GSR_FLAG_DEBUG = 1
GSR_FLAG_NO_LARGE_SORT = 4
GSR_FLAG_TIMING = 2
GSR_FLAG_COMPOSITE_TILES = 8
GSR_FLAG_NO_COLOR_GRAD = 256
GSR_FLAG_WAVE_PRIORITY = 512
STAGES = ("preprocess", "scan", "scatter", "sort", "composite_fwd", "composite_bwd", "preprocess_bwd")


class GsrStrided(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("pixel_stride", C.c_int64), ("channel_stride", C.c_int64)]


class GsrHeader(C.Structure):
    _fields_ = [("num_rendered", C.c_uint64), ("overflow", C.c_uint32), ("max_tile_count", C.c_uint32),
                ("num_busy_wgs", C.c_uint32), ("num_slots", C.c_uint32), ("num_points", C.c_uint32), ("row_overflow", C.c_uint32),
                ("reserved", C.c_uint32 * 8)]


class GsrViewExt(C.Structure):
    """Optional extras of one view (include/gpsgs.h): device pointer to the {begin, end} row range, work-order hint, and (ABI 3) the SH-colour /
    precomputed-covariance inputs of the upstream interface with their gradient outputs."""
    _fields_ = [("row_range", C.c_void_p), ("order_hint", C.c_uint32), ("sh_degree", C.c_uint32), ("sh_coeffs", C.c_uint32), ("bin_capacity", C.c_uint32),
                ("shs", C.c_void_p), ("campos", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("dL_dsh", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("reserved", C.c_uint32 * 4)]


_lib = None


def lib():
    """Load libgpsgs_hip.so once.  Raises ImportError (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gps_gaussian_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gps-gaussian_amd/csrc`). There is no CPU fallback for this path." % LIB_PATH)
    l = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint
    l.gpsgs_abi_version.restype = i32
    l.gpsgs_abi_version.argtypes = []
    l.gpsgs_build_info.restype = C.c_char_p
    l.gpsgs_build_info.argtypes = []
    l.gsr_debug_set_wg_trace.restype = i32
    l.gsr_debug_set_wg_trace.argtypes = [vp]
    l.gpsgs_measure_sclk.restype = i32
    l.gpsgs_measure_sclk.argtypes = [vp, C.POINTER(C.c_double), vp]
    l.gsr_workspace_bytes.restype = sz
    l.gsr_workspace_bytes.argtypes = [i32, i32, i32, i64]
    l.gsr_workspace_bytes_forward_only.restype = sz
    l.gsr_workspace_bytes_forward_only.argtypes = [i32, i32, i32, i64]
    l.gsr_workspace_bytes_ex.restype = sz
    l.gsr_workspace_bytes_ex.argtypes = [i32, i32, i32, i64, u32, i32]
    l.gsr_direct_lists_ok.restype = i32
    l.gsr_direct_lists_ok.argtypes = [i32, i32, u32]
    l.gsr_forward.restype = i32
    l.gsr_forward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, f32, vp, vp, vp, vp, vp, vp, sz, i64, u32, vp]
    l.gsr_forward_notify.restype = i32
    l.gsr_forward_notify.argtypes = l.gsr_forward.argtypes + [vp, u32]
    l.gsr_forward_ex.restype = i32
    l.gsr_forward_ex.argtypes = l.gsr_forward_notify.argtypes + [C.POINTER(GsrViewExt)]
    l.gsr_backward.restype = i32
    l.gsr_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                               vp, vp, sz, i64, u32, vp]
    l.gsr_backward_ex.restype = i32
    l.gsr_backward_ex.argtypes = l.gsr_backward.argtypes + [C.POINTER(GsrViewExt)]
    l.gsr_copy_header_async.restype = i32
    l.gsr_copy_header_async.argtypes = [vp, vp, vp]
    l.gsr_read_header.restype = i32
    l.gsr_read_header.argtypes = [vp, C.POINTER(GsrHeader), vp]
    l.gsr_export_state.restype = i32
    l.gsr_export_state.argtypes = [vp, i32, i32, i32, i64, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    l.gsr_debug_count_records.restype = i32
    l.gsr_debug_count_records.argtypes = [vp, sz, i32, i32, i32, i64, u32, vp, vp]
    l.gsr_mark_visible.restype = i32
    l.gsr_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    l.gsr_selftest.restype = i32
    l.gsr_selftest.argtypes = [vp, vp]
    l.gsr_timing_read.restype = i32
    l.gsr_timing_read.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    l.gsr_pack_scratch_bytes.restype = sz
    l.gsr_pack_scratch_bytes.argtypes = [i32, i32, i32]
    sp = C.POINTER(GsrStrided)
    l.gsr_pack_views.restype = i32
    l.gsr_pack_views.argtypes = [i32, i32, i32, sp, sp, sp, sp, sp, sp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    pp = C.POINTER(C.c_void_p)
    l.gsr_pack_views_backward.restype = i32
    l.gsr_pack_views_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, pp, pp, pp, pp, pp, vp]
    l.fl_scratch_bytes.restype = sz
    l.fl_scratch_bytes.argtypes = [i32, i32, i32]
    l.fl_l1_ssim_forward.restype = i32
    l.fl_l1_ssim_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    l.fl_l1_ssim_backward.restype = i32
    l.fl_l1_ssim_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    l.up_unproject_forward.restype = i32
    l.up_unproject_forward.argtypes = [i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    l.up_unproject_backward.restype = i32
    l.up_unproject_backward.argtypes = [i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp]
    l.up_unproject_forward_dev.restype = i32
    l.up_unproject_forward_dev.argtypes = [i32, i32, vp, vp, i64, vp, vp, vp, vp, vp]
    l.up_unproject_backward_dev.restype = i32
    l.up_unproject_backward_dev.argtypes = [i32, i32, vp, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp]
    l.cs_forward.restype = i32
    l.cs_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    l.cs_backward.restype = i32
    l.cs_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    l.cv_build_forward.restype = i32
    l.cv_build_forward.argtypes = [vp, vp, pp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.cv_build_backward.restype = i32
    l.cv_build_backward.argtypes = [vp, vp, pp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.cs_lookup_forward.restype = i32
    l.cs_lookup_forward.argtypes = [pp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.cs_lookup_backward.restype = i32
    l.cs_lookup_backward.argtypes = [vp, vp, pp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.cu_upsample_forward.restype = i32
    l.cu_upsample_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    l.cu_upsample_backward.restype = i32
    l.cu_upsample_backward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    l.cu_upsample_scratch_bytes.restype = sz
    l.cu_upsample_scratch_bytes.argtypes = [i32, i32, i32, i32]
    if l.gpsgs_abi_version() != 4:
        raise ImportError("gps_gaussian_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = l
    return l


def check(rc, what):
    if rc != 0:
        raise RuntimeError("gps_gaussian_amd: %s failed: %s (%d)" % (what, _ERR.get(rc, "unknown"), rc))


def measure_sclk_mhz(device=None):
    """Shader clock (MHz) under a few milliseconds of chip-filling VALU load (gpsgs_measure_sclk); synchronises the current stream."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    scratch = torch.zeros((3,), dtype=torch.int64, device=dev)
    mhz = C.c_double(0.0)
    with torch.cuda.device(dev):
        check(lib().gpsgs_measure_sclk(scratch.data_ptr(), C.byref(mhz), torch.cuda.current_stream(dev).cuda_stream), "gpsgs_measure_sclk")
    return float(mhz.value)


def timing_read():
    """Per-stage {name: (total_ms, launches)} of the calls made with GSR_FLAG_TIMING since the last read."""
    ms = (C.c_float * len(STAGES))()
    n = (C.c_int * len(STAGES))()
    check(lib().gsr_timing_read(ms, n), "gsr_timing_read")
    return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(STAGES)}
