"""Drop-in replacement for the `diff_gaussian_rasterization` Python module on MI355X.

Mirrors the interface the reference binds at /root/reference/gaussian_renderer/__init__.py:14,36-62:
`GaussianRasterizationSettings` (12-field NamedTuple, no `antialiasing`), `GaussianRasterizer(raster_settings)` whose
call returns the 2-tuple `(color[3,H,W], radii[P])`, and the autograd contract of upstream's `_RasterizeGaussians`
(gradients for means3D, means2D, colors_precomp, opacities, scales, rotations; None for the rest) -- SURVEY.md section 8b.

All arithmetic happens in libgpsgs_hip.so (hand-written gfx950 kernels) through the C-ABI of include/gpsgs.h.
There is NO eager/CPU fallback: non-GPU tensors or a missing library raise.

Host-side design notes (MI355X-first, not a translation of upstream's C++ glue):
  * one workspace tensor per forward replaces upstream's three resizable byte buffers and travels to backward in ctx;
  * upstream blocks on a D2H copy of `num_rendered` in the middle of every forward to size its sort buffers.  Here the
    whole forward is enqueued against an instance capacity learnt from previous calls; the kernels record the R they
    needed + an overflow flag in the workspace header.  `GPSGS_CHECK=sync` (default) has the binning scan kernel store
    that header straight into pinned host memory (gsr_forward_notify) and spins on it: the host learns R ~40 us into the
    forward, while scatter / sort / compositing are still running, so the check costs no GPU idle time; on overflow it
    transparently re-runs with a larger capacity, so results are always exact.  `GPSGS_CHECK=deferred` never blocks:
    the header lands in pinned memory and is examined on the next call into this module; an overflow then raises
    (capacity grows for later calls).  `GPSGS_CHECK=none` never looks at the header at all: the call sequence is then a pure
    stream of kernel launches + one memset, which is what a HIP graph capture (torch.cuda.graph) needs; the caller owns the
    capacity question (run the step once eagerly in sync mode first: the learnt capacity is reused), an overflowing view is
    simply not rendered (the workspace header says so).
"""
import ctypes as C
import os
import threading
import time
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _capi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_extra_flags = 0  # bench.py sets GSR_FLAG_TIMING here to bracket every kernel with hipEvents


def set_stage_timing(on, stage=None):
    """Turn hipEvent bracketing of the kernels on/off for subsequent calls (read with _capi.timing_read()).
    stage: name from _capi.STAGES to bracket only that kernel (2 events per call instead of ~20)."""
    global _extra_flags
    if not on:
        _extra_flags = 0
    elif stage is None:
        _extra_flags = _capi.GSR_FLAG_TIMING
    else:
        _extra_flags = _capi.GSR_FLAG_TIMING | ((_capi.STAGES.index(stage) + 1) << 4)


# ---- capacity policy --------------------------------------------------------------------------------------------
_MIN_CAP = 1 << 16
_early_notify = os.environ.get("GPSGS_EARLY_NOTIFY", "1") != "0"  # 0: sync mode waits for the whole forward (event after a header copy)
_state = {}  # device index -> dict(ratio=instances per Gaussian seen so far, pending=[(event, pinned_header, P)])
_lock = threading.Lock()  # the per-device capacity state and the pinned header rings may be used from several host threads


def _dev_state(dev):
    st = _state.get(dev.index)
    if st is None:
        with _lock:
            st = _state.setdefault(dev.index, dict(ratio=4.0, floor=_MIN_CAP, pending=[]))
    return st


def _capacity_for(st, P):
    # generous: 288 GB of HBM3E means 12 B/instance of slack is free; 2x the worst ratio seen, never below the floor
    return int(min(max(st["floor"], int(P * st["ratio"] * 2.0) + 4096), 0x7fffffff))


def _composite_flag():
    """GPSGS_COMPOSITE=valu: the compositing kernels that evaluate every (pixel, splat) exponent on the vector ALUs (gsr_composite.hip).
    GPSGS_COMPOSITE=tiles: the kernels that take the exponents from bf16 matrix-core tiles (gsr_composite_tiles.hip; exact split
    evaluation, same results within rounding)."""
    m = os.environ.get("GPSGS_COMPOSITE", _DEFAULT_FAMILY)
    if m not in ("valu", "tiles"):
        raise ValueError("GPSGS_COMPOSITE must be 'valu' or 'tiles'")
    return _capi.GSR_FLAG_COMPOSITE_TILES if m == "tiles" else 0


_DEFAULT_FAMILY = "tiles"


def _check_mode():
    m = os.environ.get("GPSGS_CHECK", "sync")
    if m not in ("sync", "deferred", "none"):
        raise ValueError("GPSGS_CHECK must be 'sync', 'deferred' or 'none'")
    return m


def _drain_pending(st, block=False):
    keep = []
    for ev, hdr, P in st["pending"]:
        if block:
            ev.synchronize()
        if ev.query():
            R, overflow, need = _decode(hdr)
            _learn(st, R, need, P)
            if overflow:
                st["pending"] = []
                raise RuntimeError(
                    "gps_gaussian_amd: a previous rasteriser call (GPSGS_CHECK=deferred) needed %d instances, more than its "
                    "capacity; that image was not rendered. Capacity has been raised; re-run, or use GPSGS_CHECK=sync." % R)
        else:
            keep.append((ev, hdr, P))
    st["pending"] = keep


def _decode(hdr):
    """hdr: int64[4] view of the first 32 header bytes -> (R, overflow, capacity needed = max(R, gradient-record slots))."""
    R = int(hdr[0])
    overflow = int(hdr[1]) & 0xffffffff
    slots = (int(hdr[2]) >> 32) & 0xffffffff
    return R, overflow, max(R, slots)


def _learn(st, R, need, P):
    with _lock:  # read-modify-write of monotone maxima
        st["last_R"] = R
        if P > 0:
            st["ratio"] = max(st["ratio"], need / P)
        st["floor"] = max(st["floor"], min(int(need * 1.25) + 4096, 0x7fffffff))


def _ptr(t):
    return t.data_ptr() if t is not None else None  # ctypes converts a Python int to the void* argument


class _HeaderRing:
    """Pinned 32-byte header slots + reusable events (allocating pinned memory / events per call costs tens of us)."""

    def __init__(self, n=64):
        self.buf = torch.zeros((n, 4), dtype=torch.int64).pin_memory()  # 32-byte header prefix per slot
        self.np = self.buf.numpy()  # same memory; plain numpy scalars are much cheaper to read than 0-d tensors
        self.base = self.buf.data_ptr()
        self.events = [torch.cuda.Event() for _ in range(n)]
        self.n, self.i = n, 0

        self.np32 = self.np.view("uint32")  # [n, 8]; word 7 of a slot is the early-notification sequence number
        self.seq = 0

    def next(self):
        with _lock:
            i = self.i
            self.i = (i + 1) % self.n
        return self.np[i], C.c_void_p(self.base + 32 * i), self.events[i]

    def next_notify(self):
        """-> (int64[4] header view, uint32[8] view, pointer, sequence number the device will store in word 7)."""
        with _lock:  # slot and sequence number are handed out together: two host threads never share either
            i = self.i
            self.i = (i + 1) % self.n
            seq = self.seq = self.seq % 0x7fffffff + 1  # never 0, never equal to what the slot holds from its previous use
        return self.np[i], self.np32[i], C.c_void_p(self.base + 32 * i), seq


def _wait_notify(w32, seq, cur_stream):
    """Spin until the device has stored `seq` (gsr_forward_notify): typically the preprocess + scan time, ~40 us."""
    n = 0
    while int(w32[7]) != seq:
        n += 1
        if n & 0x3ff == 0:  # ~every 100 us of spinning (the store normally lands within the first ~40 us)
            if cur_stream.query():  # the stream drained without the store: surface the device error
                if int(w32[7]) == seq:
                    break
                torch.cuda.synchronize()
                raise RuntimeError("gps_gaussian_amd: the rasteriser forward finished without publishing its header")
            if n & 0x3fff == 0:
                time.sleep(0)  # a very long wait (> ~1.5 ms: the stream is backed up): give the core to another thread of this rank


# ---- several forwards in flight (pts2render: one stream per sample of the batch) ---------------------------------------------------
_deferred = None  # list collecting the capacity checks of forwards enqueued inside `defer_capacity_checks()`


class defer_capacity_checks:
    """Context: sync-mode forwards enqueued inside it do not wait for their capacity notification one by one; all of them are checked
    (and, on overflow, re-rendered in place) when the context exits.  Lets a caller enqueue independent views on several streams back
    to back so that they run concurrently, with the same exactness guarantee as the blocking form.  Their images must not be consumed
    before the context has exited."""

    def __enter__(self):
        global _deferred
        self.prev, _deferred = _deferred, []
        return self

    def __exit__(self, et, ev, tb):
        global _deferred
        todo, _deferred = _deferred, self.prev
        if et is None:
            for finish in todo:
                finish()
        return False


_rings = {}


def _ring(dev, kind="notify"):
    """Pinned header slots of a device.  Two rings: "notify" (slots the scan kernel writes directly, reused as soon as the host has seen
    the sequence word) and "copy" (targets of header copies that may stay pending across calls in deferred mode) -- a slot of one
    kind can so never be handed out while the other path still has it in flight."""
    r = _rings.get((dev.index, kind))
    if r is None:
        ring = _HeaderRing()  # (allocates pinned memory: outside the lock)
        with _lock:
            r = _rings.setdefault((dev.index, kind), ring)
    return r


def _prep(t, name, shape_tail, device):
    # fast path (what the reference's callers pass): fp32, contiguous, on the device, right shape -- used as is (only its address is needed)
    if (isinstance(t, torch.Tensor) and t.dtype is torch.float32 and t.device == device and t.is_contiguous()
            and (shape_tail is None or (t.dim() == 2 and tuple(t.shape[1:]) == shape_tail)) and (shape_tail != (4,) or t.data_ptr() % 16 == 0)):
        return t
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if t.device != device:
        raise RuntimeError("gps_gaussian_amd: %s is on %s but means3D is on %s (no CPU path exists)" % (name, t.device, device))
    if t.requires_grad:
        t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    if shape_tail == (4,) and t.data_ptr() % 16:  # only the quaternions are read as float4 (row slices of packed buffers stay aligned)
        t = t.clone()
    if shape_tail is not None and (t.dim() < 1 or tuple(t.shape[1:]) != shape_tail):
        raise RuntimeError("%s must have dimensions (num_points, %s)" % (name, ", ".join(map(str, shape_tail))))
    return t


def _cam(t, n, device):
    # H1: in training the camera tensors arrive as pinned CPU tensors, in the test scripts as GPU tensors
    if t.device != device or t.dtype != torch.float32 or t.requires_grad or not t.is_contiguous():
        t = t.detach().to(device=device, dtype=torch.float32, non_blocking=True).contiguous()
    if t.numel() != n:
        raise RuntimeError("camera tensor must have %d elements" % n)
    return t


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def _device_guard(dev):
    # switching the current device costs ~10 us; skip it when the tensors already live on the current device
    return _NOGUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _forward_impl(ctx, means3D, colors_precomp, opacities, scales, rotations, raster_settings, needs_grad, out_color=None):
    """One view's forward through the C-ABI (capacity policy, early notification, overflow repair).  `ctx` is any attribute holder: the
    autograd ctx of _RasterizeGaussians, or a plain namespace when a caller drives several views itself (render_api._RenderBatch).
    Leaves on it: raster_settings, cap, family, saved = (m3, col, opa, sca, rot, view, proj, bg, radii, ws) and, inside
    defer_capacity_checks(), ws_box.  out_color: optional preallocated contiguous fp32 [3,H,W] the image is written into.
    -> (color, radii)"""
    rs = raster_settings
    lib = _capi.lib()
    if not means3D.is_cuda:
        raise RuntimeError("gps_gaussian_amd: rasteriser inputs must live on a GPU (no CPU fallback)")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    m3 = _prep(means3D, "means3D", (3,), dev)
    col = _prep(colors_precomp, "colors_precomp", (3,), dev)
    opa = _prep(opacities, "opacities", None, dev).reshape(-1)
    sca = _prep(scales, "scales", (3,), dev)
    rot = _prep(rotations, "rotations", (4,), dev)
    if not (col.shape[0] == opa.shape[0] == sca.shape[0] == rot.shape[0] == P):
        raise RuntimeError("all per-Gaussian inputs must have num_points rows")
    view = _cam(rs.viewmatrix, 16, dev)
    proj = _cam(rs.projmatrix, 16, dev)
    bg = _cam(rs.bg, 3, dev)
    family = _composite_flag()
    flags = (_capi.GSR_FLAG_DEBUG if rs.debug else 0) | _extra_flags | family
    mode = _check_mode()
    st = _dev_state(dev)
    if mode != "none" and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("gps_gaussian_amd: the capacity check reads a header back on the host and cannot run under graph capture; "
                           "warm up eagerly, then capture with GPSGS_CHECK=none")
    with _device_guard(dev):
        if mode != "none":
            _drain_pending(st)
        cur_stream = torch.cuda.current_stream(dev)
        stream = cur_stream.cuda_stream
        ring = _ring(dev)
        if out_color is None:
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        else:
            color = out_color
            if (color.dtype is not torch.float32 or color.device != dev or tuple(color.shape) != (3, H, W) or not color.is_contiguous()):
                raise RuntimeError("gps_gaussian_amd: out_color must be a contiguous fp32 [3, H, W] tensor on the inputs' device")
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cap = _capacity_for(st, P)
        # inference (no input needs a gradient): skip the backward tail of the workspace (37 B per instance slot)
        ws_bytes = lib.gsr_workspace_bytes if needs_grad else lib.gsr_workspace_bytes_forward_only
        while True:
            nbytes = ws_bytes(P, W, H, cap)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            if mode == "sync" and P > 0 and _early_notify:
                # the device publishes the instance count to pinned memory right after the binning scan; the host
                # checks capacity while scatter / sort / compositing are still running (no GPU idle time)
                hdr, w32, hdr_ptr, seq = ring.next_notify()
                # the same header carries the longest bin list: as long as none has exceeded 1024 entries on this device,
                # the (then idle, ~5 us) large-list sort launch is left out -- a surprise is reported like an overflow
                skip_large = not st.get("big_bins", False)
                flags = (flags & ~_capi.GSR_FLAG_NO_LARGE_SORT) | (_capi.GSR_FLAG_NO_LARGE_SORT if skip_large else 0)
                rc = lib.gsr_forward_notify(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot),
                                            float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy), _ptr(view),
                                            _ptr(proj), _ptr(bg), _ptr(color), _ptr(radii), _ptr(ws), nbytes, cap, flags,
                                            stream, hdr_ptr, seq)
                _capi.check(rc, "gsr_forward_notify")
                if _deferred is not None:
                    # checked when the enclosing defer_capacity_checks() exits (several views in flight); an overflow is repaired
                    # there, in place: same output tensors, a larger workspace in ctx.ws_box
                    box = [ws, cap]
                    ctx.ws_box = box

                    def finish(hdr=hdr, w32=w32, seq=seq, flags=flags):
                        while True:
                            _wait_notify(w32, seq, cur_stream)
                            R, overflow, need = _decode(hdr)
                            _learn(st, R, need, P)
                            if int(w32[3]) > 768:
                                st["big_bins"] = True
                            if not overflow:
                                return
                            if box[1] >= 0x7fffffff:
                                raise RuntimeError("gps_gaussian_amd: this view needs %d (Gaussian, bin) instances, more than the 2^31 - 1 the "
                                                   "workspace layout can address" % R)
                            with torch.cuda.stream(cur_stream):
                                box[1] = _capacity_for(st, P)
                                nb = ws_bytes(P, W, H, box[1])
                                box[0] = torch.empty((nb,), dtype=torch.uint8, device=dev)
                                hdr, w32, hdr_ptr2, seq = ring.next_notify()
                                fl = (flags & ~_capi.GSR_FLAG_NO_LARGE_SORT) | (0 if st.get("big_bins", False) else _capi.GSR_FLAG_NO_LARGE_SORT)
                                _capi.check(lib.gsr_forward_notify(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot),
                                                                   float(rs.scale_modifier), float(rs.tanfovx), float(rs.tanfovy), _ptr(view),
                                                                   _ptr(proj), _ptr(bg), _ptr(color), _ptr(radii), _ptr(box[0]), nb, box[1], fl,
                                                                   cur_stream.cuda_stream, hdr_ptr2, seq), "gsr_forward_notify")

                    _deferred.append(finish)
                    break
                _wait_notify(w32, seq, cur_stream)
                R, overflow, need = _decode(hdr)
                _learn(st, R, need, P)
                if int(w32[3]) > 768:
                    st["big_bins"] = True  # sticky, with margin: from now on the large-list sort is always launched
                if not overflow:
                    break
                if cap >= 0x7fffffff:
                    raise RuntimeError("gps_gaussian_amd: this view needs %d (Gaussian, bin) instances, more than the 2^31 - 1 the workspace "
                                       "layout can address" % R)
                cap = _capacity_for(st, P)  # the in-flight kernels of the failed attempt exit at once on the overflow flag
                continue
            flags &= ~_capi.GSR_FLAG_NO_LARGE_SORT  # only the early-notification path can verify that shortcut
            rc = lib.gsr_forward(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot), float(rs.scale_modifier),
                                 float(rs.tanfovx), float(rs.tanfovy), _ptr(view), _ptr(proj), _ptr(bg), _ptr(color),
                                 _ptr(radii), _ptr(ws), nbytes, cap, flags, stream)
            _capi.check(rc, "gsr_forward")
            if P == 0 or mode == "none":
                break
            cring = _ring(dev, "copy")
            if mode == "deferred" and len(st["pending"]) >= cring.n - 1:
                _drain_pending(st, block=True)  # never reuse a pinned slot that is still in flight
            hdr, hdr_ptr, ev = cring.next()
            _capi.check(lib.gsr_copy_header_async(_ptr(ws), hdr_ptr, stream), "gsr_copy_header_async")
            ev.record(cur_stream)
            if mode == "deferred":
                st["pending"].append((ev, hdr, P))
                break
            ev.synchronize()
            R, overflow, need = _decode(hdr)
            _learn(st, R, need, P)
            if not overflow:
                break
            if cap >= 0x7fffffff:
                raise RuntimeError("gps_gaussian_amd: this view needs %d (Gaussian, bin) instances, more than the 2^31 - 1 the workspace "
                                   "layout can address" % R)
            cap = _capacity_for(st, P)  # grown by _learn; re-run the whole (cheap) forward
    ctx.raster_settings = rs
    ctx.cap = cap
    ctx.family = family  # the backward must repeat the forward's per-pixel decisions: same kernel family
    ctx.saved = (m3, col, opa, sca, rot, view, proj, bg, radii, ws)
    return color, radii


def _backward_impl(ctx, saved, grad_out_color, arena):
    """One view's backward through the C-ABI.  saved: the tuple _forward_impl left in ctx.saved; arena: optional five preallocated
    gradient tensors (means3D, colours, opacities, scales, rotations).  -> (d_m3, d_m2, d_col, d_op, d_sc, d_rot)"""
    rs = ctx.raster_settings
    lib = _capi.lib()
    m3, col, opa, sca, rot, view, proj, bg, radii, ws = saved
    cap = ctx.cap
    box = getattr(ctx, "ws_box", None)
    if box is not None:  # forward ran inside defer_capacity_checks(): the workspace may have been replaced by the overflow repair
        ws, cap = box
    dev = m3.device
    P = m3.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    g = grad_out_color.detach().to(dtype=torch.float32).contiguous()  # H3: may arrive non-contiguous
    with _device_guard(dev):
        st = _dev_state(dev)
        if _check_mode() != "none":
            _drain_pending(st, block=(_check_mode() == "deferred"))
        stream = torch.cuda.current_stream(dev).cuda_stream
        if arena is not None and all(a.dtype == torch.float32 and a.is_contiguous() and a.device == dev and tuple(a.shape) == (P, c)
                                     for a, c in zip(arena, (3, 3, 1, 3, 4))):
            d_m3, d_col, d_op, d_sc, d_rot = arena
            d_m2 = torch.empty((P, 3), dtype=torch.float32, device=dev)
        else:
            # one allocation, six contiguous gradient arrays carved out of it (quaternion gradient first: it is stored as float4)
            buf = torch.empty((P * 17,), dtype=torch.float32, device=dev)
            d_rot = buf[:4 * P].view(P, 4)
            d_m3 = buf[4 * P:7 * P].view(P, 3)
            d_m2 = buf[7 * P:10 * P].view(P, 3)
            d_col = buf[10 * P:13 * P].view(P, 3)
            d_sc = buf[13 * P:16 * P].view(P, 3)
            d_op = buf[16 * P:].view(P, 1)
        if P > 0:
            rc = lib.gsr_backward(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot), float(rs.scale_modifier),
                                  float(rs.tanfovx), float(rs.tanfovy), _ptr(view), _ptr(proj), _ptr(bg), _ptr(radii), _ptr(g),
                                  _ptr(d_m3), _ptr(d_m2), _ptr(d_col), _ptr(d_op), _ptr(d_sc), _ptr(d_rot), _ptr(ws),
                                  ws.numel(), cap, (_capi.GSR_FLAG_DEBUG if rs.debug else 0) | _extra_flags | ctx.family, stream)
            _capi.check(rc, "gsr_backward")
    return d_m3, d_m2, d_col, d_op, d_sc, d_rot


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena=None):
        # grad_arena (optional, internal to pts2render): five preallocated fp32 tensors [P,3],[P,3],[P,1],[P,3],[P,4] -- row slices
        # of batch-wide buffers -- that the backward writes dL/d(means3D, colours, opacities, scales, rotations) into instead of
        # fresh allocations, so that the batch's gradients arrive already concatenated (render_api._SplitRows)
        ctx.grad_arena = grad_arena
        color, radii = _forward_impl(ctx, means3D, colors_precomp, opacities, scales, rotations, raster_settings, any(ctx.needs_input_grad))
        ctx.save_for_backward(*ctx.saved)
        ctx.saved = None
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # otherwise autograd fills a zero int32 [P] "gradient" for radii on every backward
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        if grad_out_color is None:  # the image did not take part in the loss
            return (None,) * 10
        d_m3, d_m2, d_col, d_op, d_sc, d_rot = _backward_impl(ctx, ctx.saved_tensors, grad_out_color, ctx.grad_arena)
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        return d_m3, d_m2, None, d_col, d_op, d_sc, d_rot, None, None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, grad_arena)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                grad_arena=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is not None:
            raise NotImplementedError(
                "gps_gaussian_amd: spherical-harmonics colours are outside the GPS-Gaussian hot path (the reference always "
                "passes colors_precomp, gaussian_renderer/__init__.py:54-62)")
        if cov3D_precomp is not None:
            raise NotImplementedError(
                "gps_gaussian_amd: precomputed 3D covariances are outside the GPS-Gaussian hot path (the reference always "
                "passes scales+rotations, gaussian_renderer/__init__.py:54-62)")
        return rasterize_gaussians(means3D, means2D, None, colors_precomp, opacities, scales, rotations, None,
                                   self.raster_settings, grad_arena)


def last_stats(device=None):
    """Capacity-policy state (instances-per-Gaussian ratio seen, capacity floor) for diagnostics."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return dict(_dev_state(dev), pending=len(_dev_state(dev)["pending"]))


def export_state(ws, P, W, H, cap):
    """Debug/parity helper: unpack a forward's workspace into tensors: depth, xy, conic_opacity, rect (the 8x8-BIN rect
    bx0,by0,bx1,by1 each Gaussian is listed in), ranges [bx*by, 2] (per-bin list range), point_list, final_T, n_contrib."""
    lib = _capi.lib()
    dev = ws.device
    hdr = ws[:16].view(torch.int64).cpu()
    R = int(hdr[0])
    bx, by = ((W + 7) // 8 + 3) // 4 * 4, (H + 7) // 8   # padded bin grid (gsr_common.h)
    out = dict(
        depth=torch.empty(P, device=dev), xy=torch.empty(P, 2, device=dev), conic_opacity=torch.empty(P, 4, device=dev),
        rect=torch.empty(P, 4, dtype=torch.int32, device=dev), ranges=torch.empty(bx * by, 2, dtype=torch.int64, device=dev),
        point_list=torch.empty(max(cap, 1), dtype=torch.int32, device=dev), final_T=torch.empty(H, W, device=dev),
        n_contrib=torch.empty(H, W, dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        rc = lib.gsr_export_state(_ptr(ws), P, W, H, cap, _ptr(out["depth"]), _ptr(out["xy"]), _ptr(out["conic_opacity"]),
                                  _ptr(out["rect"]), _ptr(out["ranges"]), _ptr(out["point_list"]), _ptr(out["final_T"]),
                                  _ptr(out["n_contrib"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _capi.check(rc, "gsr_export_state")
    torch.cuda.synchronize(dev)
    out["point_list"] = out["point_list"][:R]
    out["num_rendered"] = R
    out["bx"], out["by"] = bx, by
    out["overflow"] = int(hdr[1]) & 0xffffffff
    return out
