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
    stream of kernel launches only, which is what a HIP graph capture (torch.cuda.graph) needs; the caller owns the
    capacity question (run the step once eagerly in sync mode first: the learnt capacity is reused), an overflowing view is
    simply not rendered (the workspace header says so).
"""
import ctypes as C
import os
import sys
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


_DIRECT_CAP = 1024  # entries per bin of direct lists (GSR_DIRECT_MAX_CAP: one wave sorts a bin's list)


def _bin_capacity(st, W, H):
    """Per-bin capacity of DIRECT bin lists for the next view on this device, or 0 = scanned lists (include/gpsgs.h GsrViewExt.bin_capacity).
    Direct lists drop two dependent launches from the forward (scan -> scatter) but a bin holds at most `capacity` entries: they are used until
    this device sees bin lists longer than 768 entries (`big_bins`, _note_longest -- the untrained-heads regime, lists of 1,000-3,000), and for images
    of up to 65,536 bins (2048^2).  A view that turns out to need longer lists is reported like an overflow and repaired by the callers' re-run
    loops with scanned lists.  GPSGS_LISTS=scanned / direct forces the form (direct: still subject to the size limits)."""
    m = os.environ.get("GPSGS_LISTS", "auto")
    if m not in ("auto", "scanned", "direct"):
        raise ValueError("GPSGS_LISTS must be 'auto', 'scanned' or 'direct'")
    if m == "scanned" or (m == "auto" and st.get("big_bins", False)):
        return 0
    key = ("direct_ok", W, H)
    ok = st.get(key)
    if ok is None:
        ok = st[key] = bool(_capi.lib().gsr_direct_lists_ok(W, H, _DIRECT_CAP))
    return _DIRECT_CAP if ok else 0


def _composite_flag():
    """GPSGS_COMPOSITE=valu: the compositing kernels that evaluate every (pixel, splat) exponent on the vector ALUs (gsr_composite.hip).
    GPSGS_COMPOSITE=tiles: the kernels that take the exponents from bf16 matrix-core tiles (gsr_composite_tiles.hip; exact split
    evaluation, same results within rounding)."""
    m = os.environ.get("GPSGS_COMPOSITE", _DEFAULT_FAMILY)
    if m not in ("valu", "tiles"):
        raise ValueError("GPSGS_COMPOSITE must be 'valu' or 'tiles'")
    return _capi.GSR_FLAG_COMPOSITE_TILES if m == "tiles" else 0


_DEFAULT_FAMILY = "tiles"


def _wave_priority_flag(st=None, stream=None, deterministic=False):
    """GSR_FLAG_WAVE_PRIORITY (include/gpsgs.h: hardware wave priorities in the tile compositing kernels; results unchanged) for a view that has
    the chip to itself.  GPSGS_WAVE_PRIORITY=0 / 1 forces it off / on; default (unset): on, unless this device saw rasteriser calls on MORE THAN ONE
    stream within the last 50 ms -- user code that spreads views over streams overlaps their kernels, where the scheme costs ~2 % of the
    aggregate rate instead of gaining 4 - 8 %.  deterministic (graph capture, GPSGS_CHECK=none): the wall-clock history is
    not consulted -- a captured graph, and a run that is meant to be reproducible launch for launch, must not pick a kernel variant from timing
    (ADVICE r05): on, unless the environment says otherwise."""
    env = os.environ.get("GPSGS_WAVE_PRIORITY")
    if env == "0":
        return 0
    if env is None and not deterministic and st is not None and stream is not None:
        now = time.perf_counter()
        with _lock:  # forwards may run from several host threads
            seen = st.setdefault("streams", {})
            seen[stream] = now
            if len(seen) > 1:
                for k in [k for k, t in seen.items() if now - t > 0.05]:
                    del seen[k]
            several = len(seen) > 1
        if several:
            return 0
    return _capi.GSR_FLAG_WAVE_PRIORITY


def _check_mode():
    m = os.environ.get("GPSGS_CHECK", "sync")
    if m not in ("sync", "deferred", "none"):
        raise ValueError("GPSGS_CHECK must be 'sync', 'deferred' or 'none'")
    return m


def _drain_pending(st, block=False):
    """Look at the capacity results of earlier GPSGS_CHECK=deferred forwards.  Two kinds of entries: ("note", ...) -- the binning scan stores the
    header straight into a pinned slot ~40 us into the forward (round 4: deferred mode uses the early notification too; until then it copied the
    header at the END of the forward, so the blocking drain in the backward waited for the whole forward and deferred ran SLOWER than sync) -- and
    ("ev", ...) -- a header copy + event at the end of the forward (GPSGS_EARLY_NOTIFY=0)."""
    keep = []
    with _lock:  # the swap is atomic with respect to other threads' drains and appends (an autograd thread's backward drains too)
        pending, st["pending"] = st["pending"], []
    try:
        for i, ent in enumerate(pending):
            if ent[0] == "note":
                _, (slot, hdr, w32, seq), stream, P, dev_index = ent
                if int(w32[7]) != seq:
                    if not block:
                        keep.append(ent)
                        continue
                    try:
                        _wait_notify(w32, seq, stream)
                    except Exception:
                        _rings[(dev_index, "notify")].release(slot)
                        keep.extend(pending[i + 1:])  # the entries behind this one stay pending: their slots and capacity results are not lost
                        raise
                R, overflow, need = _decode(hdr)
                npts, longest = int(w32[6]), int(w32[3])
                _rings[(dev_index, "notify")].release(slot)
                _learn(st, R, need, P if P is not None else npts, longest)
                _note_longest(st, longest)
            else:
                _, ev, hdr, P = ent
                if block:
                    ev.synchronize()
                if not ev.query():
                    keep.append(ent)
                    continue
                R, overflow, need = _decode(hdr)
                _learn(st, R, need, P if P is not None else int(hdr[3]) & 0xffffffff, (int(hdr[1]) >> 32) & 0xffffffff)
            if overflow:
                keep.extend(pending[i + 1:])  # (their slots are looked at -- and released -- by the next call)
                raise RuntimeError(
                    "gps_gaussian_amd: a previous rasteriser call (GPSGS_CHECK=deferred) was not rendered: it needed %d (Gaussian, bin) instances / "
                    "gradient slots -- more than the capacity of its workspace -- and its longest bin list was %d entries. The capacity has been raised; re-run, "
                    "or use GPSGS_CHECK=sync." % (need, longest if ent[0] == "note" else (int(hdr[1]) >> 32) & 0xffffffff))
    finally:
        with _lock:
            st["pending"] = keep + st["pending"]


def _decode(hdr):
    """hdr: int64[4] view of the first 32 header bytes -> (R, overflow, capacity needed = max(R, gradient-record slots))."""
    R = int(hdr[0])
    overflow = int(hdr[1]) & 0xffffffff
    slots = (int(hdr[2]) >> 32) & 0xffffffff
    return R, overflow, max(R, slots)


def _learn(st, R, need, P, longest=None):
    with _lock:  # read-modify-write of monotone maxima
        st["last_R"] = R
        if P is not None and P > 0:
            st["ratio"] = max(st["ratio"], need / P)
            st["last_points"] = P
        st["floor"] = max(st["floor"], min(int(need * 1.25) + 4096, 0x7fffffff))
        if longest is not None:
            st["longest"] = longest  # the next view's work-order hint (GsrViewExt.order_hint): longest bin list of this one


def _note_longest(st, longest):
    """`big_bins` -- this device renders views with long bin lists (the untrained-heads regime): the large-list sort launches are issued and the lists are
    SCANNED, not direct.  Set by one list beyond 768 entries (three quarters of what one sorting wave / one direct bin holds); cleared again after eight
    views in a row whose longest list stayed below 512 (stage-2 training leaves the regime as the scales shrink: the fast path must come back)."""
    with _lock:
        if longest > 768:
            st["big_bins"], st["short_streak"] = True, 0
        elif longest <= 512 and st.get("big_bins", False):
            st["short_streak"] = st.get("short_streak", 0) + 1
            if st["short_streak"] >= 8:
                st["big_bins"], st["short_streak"] = False, 0
        else:
            st["short_streak"] = 0


def _ptr(t):
    return t.data_ptr() if t is not None else None  # ctypes converts a Python int to the void* argument


class _HeaderRing:
    """Pinned 32-byte header slots + reusable events (allocating pinned memory / events per call costs tens of us).

    Two uses: `next()` hands out slots round-robin for header COPIES (the caller bounds the number in flight: _drain_pending), and
    `acquire_notify()` / `release()` hand out slots the scan kernel writes directly (gsr_forward_notify) with IN-FLIGHT ACCOUNTING: a
    slot goes back to the free list only after the host has seen its sequence word, and when every slot is out (a batch of more forwards
    than slots inside one defer_capacity_checks(), nested callers, many threads) the ring GROWS by another pinned chunk instead of
    reissuing a slot that is still being polled."""

    def __init__(self, n=64):
        self.n, self.i = n, 0
        self.chunks = []  # (int64 [n,4] view, uint32 [n,8] view: word 7 of a slot = sequence number, base address, the pinned tensor)
        self.free = []
        self.seq = 0
        self._grow()
        self.np, self.base, self.events = self.chunks[0][0], self.chunks[0][2], [torch.cuda.Event() for _ in range(n)]

    def next(self):
        with _lock:
            i = self.i
            self.i = (i + 1) % self.n
        return self.np[i], C.c_void_p(self.base + 32 * i), self.events[i]

    def acquire_notify(self):
        """-> (slot handle, int64[4] header view, uint32[8] view, pointer, sequence number the device will store in word 7)."""
        while True:
            with _lock:  # slot and sequence number are handed out together: two host threads never share either
                if self.free:
                    c, k = slot = self.free.pop()
                    seq = self.seq = self.seq % 0x7fffffff + 1  # never 0, never equal to what the slot holds from its previous use
                    arr, a32, base, _ = self.chunks[c]
                    return slot, arr[k], a32[k], C.c_void_p(base + 32 * k), seq
            self._grow()

    def _grow(self):
        buf = torch.zeros((self.n, 4), dtype=torch.int64).pin_memory()  # 32-byte header prefix per slot (allocates pinned memory: outside the lock)
        arr = buf.numpy()  # same memory; plain numpy scalars are much cheaper to read than 0-d tensors
        with _lock:
            c = len(self.chunks)
            self.chunks.append((arr, arr.view("uint32"), buf.data_ptr(), buf))
            self.free.extend((c, k) for k in reversed(range(self.n)))

    def release(self, slot):
        with _lock:
            self.free.append(slot)

    def in_flight(self):
        with _lock:
            return len(self.chunks) * self.n - len(self.free)


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
_tls = threading.local()  # .deferred: list collecting the capacity checks of forwards this THREAD enqueued inside `defer_capacity_checks()`
# (per thread: the autograd engine runs backwards -- and recomputed forwards -- on its own threads, and multi-threaded hosts are supported)


def _deferred_list():
    return getattr(_tls, "deferred", None)


class defer_capacity_checks:
    """Context: sync-mode forwards enqueued inside it do not wait for their capacity notification one by one; all of them are checked
    (and, on overflow, re-rendered in place) when the context exits.  Lets a caller enqueue independent views on several streams back
    to back so that they run concurrently, with the same exactness guarantee as the blocking form.  Their images must not be consumed
    before the context has exited."""

    def __enter__(self):
        self.prev, _tls.deferred = _deferred_list(), []
        return self

    def __exit__(self, et, ev, tb):
        todo, _tls.deferred = _tls.deferred, self.prev
        # EVERY finish() runs, whatever the others do: each one waits for its header (capacity is learnt, the pinned slot returns to the ring,
        # an overflowed forward is repaired).  Stopping at the first failure would leak the remaining slots for good.  The first
        # error is re-raised afterwards -- unless an exception is already on its way out of the block, which then wins.
        first = None
        for finish in todo:
            try:
                finish()
            except Exception as e:  # noqa: BLE001
                if first is None:
                    first = e
        if et is None and first is not None:
            raise first
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


class _Rows:
    """A view whose Gaussians are a ROW RANGE of batch-wide arrays (GsrViewExt.row_range): `offsets` is an int32 device tensor, the view's
    rows are [offsets[index], offsets[index + 1]); `capacity` bounds their number (it sizes the workspace and the launches).  The host
    never learns the count: the kernels read the range from the device."""

    __slots__ = ("offsets", "index", "capacity", "ptr")

    def __init__(self, offsets, index, capacity):
        if offsets.dtype is not torch.int32 or not offsets.is_cuda or not offsets.is_contiguous() or offsets.numel() < index + 2:
            raise RuntimeError("gps_gaussian_amd: row offsets must be a contiguous int32 GPU tensor with index + 2 entries")
        self.offsets, self.index, self.capacity = offsets, int(index), int(capacity)
        self.ptr = offsets.data_ptr() + 4 * self.index


def _ext(rows, hint, appear=None, bin_cap=0):
    """appear: None, or (shs, sh_degree, campos, cov3D_precomp, dL_dsh, dL_dcov3D) tensors / None -- the SH-colour and precomputed-covariance
    inputs (and, for the backward, their gradient arrays)."""
    e = _capi.GsrViewExt()
    e.row_range = rows.ptr if rows is not None else None
    e.order_hint = int(hint) & 0xffffffff
    e.bin_capacity = int(bin_cap)
    if appear is not None:
        shs, deg, campos, cov, d_sh, d_cov = appear
        if shs is not None:
            e.shs, e.sh_degree, e.sh_coeffs, e.campos = shs.data_ptr(), int(deg), int(shs.shape[1]), campos.data_ptr()
            e.dL_dsh = d_sh.data_ptr() if d_sh is not None else None
        if cov is not None:
            e.cov3D_precomp = cov.data_ptr()
            e.dL_dcov3D = d_cov.data_ptr() if d_cov is not None else None
    return e


def _too_many(R):
    return RuntimeError("gps_gaussian_amd: this view needs %d (Gaussian, bin) instances, more than the 2^31 - 1 the workspace layout can address" % R)


def _forward_impl(ctx, means3D, colors_precomp, opacities, scales, rotations, raster_settings, needs_grad, out_color=None, rows=None,
                  radii_out=None, shs=None, cov3D_precomp=None):
    """One view's forward through the C-ABI (capacity policy, early notification, overflow repair).  `ctx` is any attribute holder: the
    autograd ctx of _RasterizeGaussians, or a plain namespace when a caller drives several views itself (render_api._RenderBatch).
    Leaves on it: raster_settings, cap, family, extra_flags, rows, saved = (m3, col, opa, sca, rot, view, proj, bg, radii, ws, sh, cov, campos)
    (col / sca / rot / sh / cov / campos: None where the other form of the input was given) and, inside defer_capacity_checks(), ws_box.
    shs [P, M, 3] (M <= 16, evaluated up to raster_settings.sh_degree towards raster_settings.campos) INSTEAD of colors_precomp;
    cov3D_precomp [P, 6] INSTEAD of scales + rotations.  out_color: optional preallocated contiguous fp32 [3,H,W] the image is written into.
    rows (a _Rows): the five inputs are batch-wide packed arrays, this view is the row range rows.offsets[rows.index : rows.index + 2] of
    them (read on the DEVICE), radii_out the batch-wide int32 radii array.  -> (color, radii)"""
    rs = raster_settings
    lib = _capi.lib()
    if not means3D.is_cuda:
        raise RuntimeError("gps_gaussian_amd: rasteriser inputs must live on a GPU (no CPU fallback)")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    N = means3D.shape[0]                      # rows of the arrays
    P = N if rows is None else rows.capacity  # Gaussians of the view, or their upper bound
    H, W = int(rs.image_height), int(rs.image_width)
    m3 = _prep(means3D, "means3D", (3,), dev)
    opa = _prep(opacities, "opacities", None, dev).reshape(-1)
    col = sca = rot = sh = cov = campos = None
    if shs is None:
        col = _prep(colors_precomp, "colors_precomp", (3,), dev)
    else:
        sh = _prep(shs, "shs", None, dev)
        deg = int(rs.sh_degree)
        if sh.dim() != 3 or sh.shape[2] != 3 or sh.shape[1] > 16 or not 0 <= deg <= 3 or (deg + 1) ** 2 > sh.shape[1]:
            raise RuntimeError("shs must have dimensions (num_points, M, 3) with (sh_degree + 1)^2 <= M <= 16 and sh_degree in 0..3")
        campos = _cam(rs.campos, 3, dev)
    if cov3D_precomp is None:
        sca = _prep(scales, "scales", (3,), dev)
        rot = _prep(rotations, "rotations", (4,), dev)
    else:
        cov = _prep(cov3D_precomp, "cov3D_precomp", (6,), dev)
    if not all(t is None or t.shape[0] == N for t in (col, opa, sca, rot, sh, cov)):
        raise RuntimeError("all per-Gaussian inputs must have num_points rows")
    appear = (sh, int(rs.sh_degree), campos, cov, None, None) if (sh is not None or cov is not None) else None
    view = _cam(rs.viewmatrix, 16, dev)
    proj = _cam(rs.projmatrix, 16, dev)
    bg = _cam(rs.bg, 3, dev)
    # the compositing family and, for a view rendered on its own (the GaussianRasterizer module: no row range), the wave-priority scheme --
    # it pays when the view's kernels have the chip to themselves; a batch spreads its views over several streams and leaves it off
    st = _dev_state(dev)
    mode = _check_mode()
    family = _composite_flag() | (_wave_priority_flag(st, torch._C._cuda_getCurrentRawStream(dev.index), deterministic=(mode == "none")) if rows is None else 0)
    extra = _extra_flags  # read ONCE per view and carried to its backward in ctx (the backward runs on an autograd thread)
    base_flags = (_capi.GSR_FLAG_DEBUG if rs.debug else 0) | extra | family
    if mode != "none" and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("gps_gaussian_amd: the capacity check reads a header back on the host and cannot run under graph capture; "
                           "warm up eagerly, then capture with GPSGS_CHECK=none")
    with _device_guard(dev):
        if mode != "none":
            _drain_pending(st)
        cur_stream = torch.cuda.current_stream(dev)
        stream = cur_stream.cuda_stream
        if out_color is None:
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        else:
            color = out_color
            if (color.dtype is not torch.float32 or color.device != dev or tuple(color.shape) != (3, H, W) or not color.is_contiguous()):
                raise RuntimeError("gps_gaussian_amd: out_color must be a contiguous fp32 [3, H, W] tensor on the inputs' device")
        if rows is None:
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
        else:
            radii = radii_out
            if radii is None or radii.dtype is not torch.int32 or radii.device != dev or radii.numel() != N or not radii.is_contiguous():
                raise RuntimeError("gps_gaussian_amd: a row-range view needs radii_out: contiguous int32 [rows of the arrays] on the inputs' device")
        # inference (no input needs a gradient): skip the backward tail of the workspace (37 B per instance slot)
        fwd_only = 0 if needs_grad else 1
        # a row-range view's P is only a bound: the instance capacity follows the Gaussian counts seen so far on this device
        p_est = P if rows is None else min(P, int(st.get("last_points", P) * 1.25) + 4096)
        early = mode in ("sync", "deferred") and P > 0 and _early_notify
        ring = _ring(dev) if early else None

        scanned_only = [False]

        def launch(cap):
            """Enqueue the whole forward against an instance capacity.  -> (workspace, bytes, notify slot or None, per-bin capacity of its lists)"""
            # direct lists unless this device sees long bin lists -- or THIS view has just shown one (a repair run; also under GPSGS_LISTS=direct)
            bin_cap = 0 if scanned_only[0] else _bin_capacity(st, W, H)
            nbytes = lib.gsr_workspace_bytes_ex(P, W, H, cap, bin_cap, fwd_only)
            st["last_ws_bytes"] = nbytes  # reported by last_stats(): what one view in flight holds (forward-only workspaces are ~3x smaller)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            flags, note = base_flags, None
            hdr_ptr, seq = None, 0
            if early:
                # the device publishes the instance count to pinned memory right after the binning scan; the host checks capacity while
                # scatter / sort / compositing are still running (no GPU idle time).  The same header carries the longest bin list: as
                # long as none has exceeded 1024 entries on this device, the (then idle, ~5 us) large-list sort launch is left out -- a
                # surprise is reported like an overflow
                slot, hdr, w32, hdr_ptr, seq = ring.acquire_notify()
                note = (slot, hdr, w32, seq)
                # (only where a repair loop exists: sync mode, also inside defer_capacity_checks().  Deferred mode cannot re-run the view, so it always
                #  launches the large-list sorts -- they return at once when the view's longest list is short: ADVICE r04)
                if mode == "sync" and not st.get("big_bins", False):
                    flags |= _capi.GSR_FLAG_NO_LARGE_SORT
            ext = _ext(rows, st.get("longest", 0), appear, bin_cap)  # work order: longest lists first, relative to the longest list seen on this device
            rc = lib.gsr_forward_ex(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot), float(rs.scale_modifier), float(rs.tanfovx),
                                    float(rs.tanfovy), _ptr(view), _ptr(proj), _ptr(bg), _ptr(color), _ptr(radii), _ptr(ws), nbytes, cap, flags,
                                    stream, hdr_ptr, seq, C.byref(ext))
            if rc != 0 and note is not None:
                ring.release(note[0])
            _capi.check(rc, "gsr_forward_ex")
            return ws, nbytes, note, bin_cap

        def settle(note, cap, bin_cap):
            """Wait for a notification and learn from it.  -> (overflowed, R)"""
            slot, hdr, w32, seq = note
            try:
                _wait_notify(w32, seq, cur_stream)
                R, overflow, need = _decode(hdr)
                npts, longest = int(w32[6]), int(w32[3])
            finally:
                ring.release(slot)  # the host has seen the sequence word (or gave up): the slot may be reissued
            _learn(st, R, need, npts if rows is not None else P, longest)
            _note_longest(st, longest)  # with margin: from now on the large-list sorts are launched and the lists are scanned
            if overflow and rows is not None and npts > P:
                raise RuntimeError("gps_gaussian_amd: the row range holds %d Gaussians, more than the capacity of %d rows the view was sized for" % (npts, P))
            if overflow and bin_cap and longest > bin_cap:
                scanned_only[0] = True  # a list that does not fit a direct bin: the repair run uses scanned lists (and _note_longest has set big_bins)
            elif overflow and cap >= 0x7fffffff:
                raise _too_many(R)
            return bool(overflow), R

        cap = _capacity_for(st, p_est)
        box = None
        dump = os.environ.get("GPSGS_DUMP_INPUTS")  # debugging aid: the inputs of the LAST forward entered, as an .npz (synchronises)
        if dump and rows is None:
            import numpy as _np
            _np.savez(dump, means3D=m3.cpu().numpy(), opacities=opa.cpu().numpy(),
                      **{k: v.cpu().numpy() for k, v in (("colors", col), ("scales", sca), ("rotations", rot), ("shs", sh), ("cov3D_precomp", cov)) if v is not None},
                      view=view.cpu().numpy(), proj=proj.cpu().numpy(), bg=bg.cpu().numpy(), W=W, H=H,
                      tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy), scale_modifier=float(rs.scale_modifier), cap=cap, needs_grad=bool(needs_grad))
        while True:
            ws, nbytes, note, bin_cap = launch(cap)
            if note is not None:
                if _deferred_list() is not None:
                    # checked when the enclosing defer_capacity_checks() exits (several views in flight); an overflow is repaired there, in
                    # place: same output tensors, a larger workspace in ctx.ws_box
                    box = [ws, cap, bin_cap]

                    def finish(note=note):
                        with torch.cuda.stream(cur_stream):
                            while settle(note, box[1], box[2])[0]:
                                box[1] = _capacity_for(st, p_est)  # the in-flight kernels of the failed attempt exit at once on the overflow flag
                                box[0], _, note, box[2] = launch(box[1])

                    _deferred_list().append(finish)
                    break
                if mode == "deferred":
                    # never blocks: the notification (it lands ~40 us into this forward) is looked at by the next call on this device
                    with _lock:
                        st["pending"].append(("note", note, cur_stream, P if rows is None else None, dev.index))
                    break
                if not settle(note, cap, bin_cap)[0]:
                    break
                cap = _capacity_for(st, p_est)  # grown by _learn; the in-flight kernels of the failed attempt exit at once on the overflow flag
                continue
            if P == 0 or mode == "none":
                break
            cring = _ring(dev, "copy")
            if mode == "deferred" and len(st["pending"]) >= cring.n - 1:
                _drain_pending(st, block=True)  # never reuse a pinned slot that is still in flight
            hdr, hdr_ptr, ev = cring.next()
            _capi.check(lib.gsr_copy_header_async(_ptr(ws), hdr_ptr, stream), "gsr_copy_header_async")
            ev.record(cur_stream)
            if mode == "deferred":
                with _lock:
                    st["pending"].append(("ev", ev, hdr, P if rows is None else None))
                break
            ev.synchronize()
            R, overflow, need = _decode(hdr)
            npts = int(hdr[3]) & 0xffffffff
            longest = (int(hdr[1]) >> 32) & 0xffffffff
            _learn(st, R, need, npts if rows is not None else P, longest)
            _note_longest(st, longest)
            if not overflow:
                break
            if bin_cap and longest > bin_cap:
                scanned_only[0] = True
                continue  # re-run with scanned lists
            if rows is not None and npts > P:
                raise RuntimeError("gps_gaussian_amd: the row range holds %d Gaussians, more than the capacity of %d rows the view was sized for" % (npts, P))
            if cap >= 0x7fffffff:
                raise _too_many(R)
            cap = _capacity_for(st, p_est)  # grown by _learn; re-run the whole (cheap) forward
    ctx.raster_settings = rs
    ctx.cap = cap
    ctx.bin_cap = bin_cap
    ctx.family = family  # the backward must repeat the forward's per-pixel decisions: same kernel family
    ctx.extra_flags = extra
    ctx.rows = rows
    if box is not None:
        ctx.ws_box = box
    ctx.saved = (m3, col, opa, sca, rot, view, proj, bg, radii, ws, sh, cov, campos)
    return color, radii


def _backward_impl(ctx, saved, grad_out_color, arena, color_grad=True):
    """One view's backward through the C-ABI.  color_grad=False: the caller does not need dL/dcolours (GSR_FLAG_NO_COLOR_GRAD: the tile
    family leaves the colour sums out; the returned colour gradient is zeros / not meaningful).  saved: the tuple _forward_impl left in ctx.saved; arena: optional five preallocated
    gradient tensors (means3D, colours, opacities, scales, rotations) -- for a row-range view (ctx.rows) they are REQUIRED and batch-wide,
    the view's rows of them are written.  -> (d_m3, d_m2, d_col, d_op, d_sc, d_rot, d_sh, d_cov); d_sh / d_cov are None unless the forward was given
    SH coefficients / precomputed covariances (d_sc, d_rot are then not meaningful)."""
    rs = ctx.raster_settings
    lib = _capi.lib()
    m3, col, opa, sca, rot, view, proj, bg, radii, ws = saved[:10]
    sh, cov, campos = saved[10:13] if len(saved) >= 13 else (None, None, None)
    cap = ctx.cap
    rows = getattr(ctx, "rows", None)
    box = getattr(ctx, "ws_box", None)
    bin_cap = getattr(ctx, "bin_cap", 0)
    if box is not None:  # forward ran inside defer_capacity_checks(): the workspace may have been replaced by the overflow repair
        ws, cap, bin_cap = box
    dev = m3.device
    N = m3.shape[0]
    P = N if rows is None else rows.capacity
    H, W = int(rs.image_height), int(rs.image_width)
    g = grad_out_color  # H3: may arrive non-contiguous (or in another dtype); the common case -- fp32, contiguous -- is used as it is
    if g.dtype is not torch.float32 or not g.is_contiguous() or g.requires_grad:
        g = g.detach().to(dtype=torch.float32).contiguous()
    with _device_guard(dev):
        st = _dev_state(dev)
        if _check_mode() != "none":
            _drain_pending(st, block=(_check_mode() == "deferred"))
        stream = torch._C._cuda_getCurrentRawStream(dev.index)  # the raw handle: torch.cuda.current_stream() builds a Stream object (~3 us)
        if arena is not None and all(a.dtype == torch.float32 and a.is_contiguous() and a.device == dev and tuple(a.shape) == (N, c)
                                     for a, c in zip(arena[:5], (3, 3, 1, 3, 4))):
            d_m3, d_col, d_op, d_sc, d_rot = arena[:5]
            d_m2 = arena[5] if len(arena) > 5 else torch.empty((N, 3), dtype=torch.float32, device=dev)
        elif rows is not None:
            raise RuntimeError("gps_gaussian_amd: a row-range view needs its batch-wide gradient arrays")
        else:
            # one allocation, six contiguous gradient arrays carved out of it (quaternion gradient first: it is stored as float4)
            # (ONE split + six views: the slice-then-view form was twelve tensor operations, ~25 us of host time on the path between the capacity
            #  notification and the backward's launch -- the stretch that decides whether a slower host keeps the GPU fed, profiles/r05_host_timeline.md)
            buf = torch.empty((P * 17,), dtype=torch.float32, device=dev)
            b_rot, b_m3, b_m2, b_col, b_sc, b_op = buf.split_with_sizes((4 * P, 3 * P, 3 * P, 3 * P, 3 * P, P))
            d_rot, d_m3, d_m2 = b_rot.view(P, 4), b_m3.view(P, 3), b_m2.view(P, 3)
            d_col, d_sc, d_op = b_col.view(P, 3), b_sc.view(P, 3), b_op.view(P, 1)
        d_sh = torch.empty_like(sh) if sh is not None else None
        d_cov = torch.empty_like(cov) if cov is not None else None
        if P > 0:
            ext = _ext(rows, 0, (sh, int(rs.sh_degree), campos, cov, d_sh, d_cov) if (sh is not None or cov is not None) else None, bin_cap)
            rc = lib.gsr_backward_ex(P, W, H, _ptr(m3), _ptr(col), _ptr(opa), _ptr(sca), _ptr(rot), float(rs.scale_modifier),
                                     float(rs.tanfovx), float(rs.tanfovy), _ptr(view), _ptr(proj), _ptr(bg), _ptr(radii), _ptr(g),
                                     _ptr(d_m3), _ptr(d_m2), _ptr(d_col), _ptr(d_op), _ptr(d_sc), _ptr(d_rot), _ptr(ws),
                                     ws.numel(), cap, (_capi.GSR_FLAG_DEBUG if rs.debug else 0) | getattr(ctx, "extra_flags", _extra_flags) | ctx.family
                                     | (0 if color_grad else _capi.GSR_FLAG_NO_COLOR_GRAD), stream, C.byref(ext))
            _capi.check(rc, "gsr_backward_ex")
    return d_m3, d_m2, d_col, d_op, d_sc, d_rot, d_sh, d_cov


# ---- the compiled host path (csrc/host_shim.cpp -> lib/_gpsgs_fast.so) -----------------------------------------------------------------------------
# The call shape the reference uses, in sync mode, goes through a C++ torch::autograd::Function that performs the same steps as _forward_impl /
# _backward_impl against the same C-ABI; everything else -- and every overflow -- takes the Python path below.  The capacity POLICY stays here.
_fast = None   # the module, or False when it is switched off / not built


def _fast_module():
    """lib/_gpsgs_fast.so, or False.  GPSGS_FAST=0 switches it off (the Python host path is then used for everything); on a machine where the
    library was built the module must load -- a missing or stale one is an ImportError, not a silent fallback."""
    global _fast
    if os.environ.get("GPSGS_FAST", "1") == "0":
        return False
    if _fast is None:
        if True:
            import importlib.util
            path = os.path.join(os.path.dirname(_capi.LIB_PATH), "_gpsgs_fast.so")
            if not os.path.exists(path):
                raise ImportError("gps_gaussian_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` (or set GPSGS_FAST=0 "
                                  "to use the Python host path)." % path)
            _capi.lib()  # libgpsgs_hip.so first: the extension links against it
            spec = importlib.util.spec_from_file_location("_gpsgs_fast", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != _capi.lib().gpsgs_abi_version():
                raise ImportError("gps_gaussian_amd: %s was built against another ABI version" % path)
            _fast = mod
    return _fast


def _fast_forward(means3D, means2D, opacities, colors_precomp, scales, rotations, rs):
    """-> (color, radii) through the compiled host path, or None: not its call shape / mode, or the view overflowed (the caller then takes the
    Python path, which repairs it).  Policy decisions (capacity, list form, flags) are taken here, from the same per-device state."""
    F = _fast_module()
    env = os.environ
    if (not F or rs.debug or _extra_flags or _deferred_list() is not None or env.get("GPSGS_CHECK", "sync") != "sync" or not _early_notify
            or env.get("GPSGS_DUMP_INPUTS") or not isinstance(means3D, torch.Tensor) or not means3D.is_cuda or torch.cuda.is_current_stream_capturing()):
        return None  # (under graph capture the Python path says what to do instead of spinning on a header that never arrives)
    dev = means3D.device
    st = _dev_state(dev)
    if st["pending"]:
        return None
    H, W = int(rs.image_height), int(rs.image_width)
    P = means3D.shape[0]
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    flags = _composite_flag() | _wave_priority_flag(st, stream)
    if not st.get("big_bins", False):
        flags |= _capi.GSR_FLAG_NO_LARGE_SORT
    cap = _capacity_for(st, P)
    bin_cap = _bin_capacity(st, W, H)
    out = F.rasterize(means3D, means2D, colors_precomp, opacities, scales, rotations, rs.bg, rs.viewmatrix, rs.projmatrix, H, W, float(rs.tanfovx),
                      float(rs.tanfovy), float(rs.scale_modifier), cap, bin_cap, st.get("longest", 0), flags)
    if out is None:
        return None
    R, overflow, longest, slots, npts, nbytes, _wait_us = F.last_header()
    st["last_ws_bytes"] = nbytes
    _learn(st, R, max(R, slots), P, longest)
    _note_longest(st, longest)
    if overflow:
        return None  # nothing was rendered (the kernels exit on the flag): the Python path re-renders with what was just learnt
    if _debug_keep_ws:
        _tls.last_ws = dict(ws=out[2], cap=cap, bin_cap=bin_cap)
    return out[0], out[1]


_debug_keep_ws = False  # tests: both host paths leave {ws, cap, bin_cap} of the last forward in _tls.last_ws (export_state needs them)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena=None):
        # grad_arena (optional, internal to pts2render): five preallocated fp32 tensors [P,3],[P,3],[P,1],[P,3],[P,4] -- row slices
        # of batch-wide buffers -- that the backward writes dL/d(means3D, colours, opacities, scales, rotations) into instead of
        # fresh allocations, so that the batch's gradients arrive already concatenated (render_api._SplitRows)
        ctx.grad_arena = grad_arena
        # stage 2 never differentiates the colours (they are input pixels, lib/GaussianRender.py:30-31): the backward then skips their sums
        ctx.color_grad = bool(ctx.needs_input_grad[3]) or sh is not None  # (dL/dsh is formed from dL/dcolour)
        color, radii = _forward_impl(ctx, means3D, colors_precomp, opacities, scales, rotations, raster_settings, any(ctx.needs_input_grad),
                                     shs=sh, cov3D_precomp=cov3Ds_precomp)
        if _debug_keep_ws:
            _tls.last_ws = dict(ws=ctx.saved[9], cap=ctx.cap, bin_cap=ctx.bin_cap)
        ctx.save_for_backward(*ctx.saved)
        ctx.saved = None
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # otherwise autograd fills a zero int32 [P] "gradient" for radii on every backward
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        if grad_out_color is None:  # the image did not take part in the loss
            return (None,) * 10
        saved = ctx.saved_tensors
        d_m3, d_m2, d_col, d_op, d_sc, d_rot, d_sh, d_cov = _backward_impl(ctx, saved, grad_out_color, ctx.grad_arena, ctx.color_grad)
        has_sh, has_cov = saved[10] is not None, saved[11] is not None
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        return (d_m3, d_m2, d_sh, (d_col if ctx.color_grad and not has_sh else None), d_op, (None if has_cov else d_sc), (None if has_cov else d_rot),
                d_cov, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, grad_arena)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        acc = sys.modules.get("gps_gaussian_amd.accelerate")  # present only when GPSGS_ACCELERATE is set (opt-in, accelerate.py)
        if acc is not None and acc._armed:
            acc.late_apply()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                grad_arena=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is None and cov3D_precomp is None and grad_arena is None:
            # the reference's call shape (gaussian_renderer/__init__.py:54-62): the compiled host path, when it applies
            out = _fast_forward(means3D, means2D, opacities, colors_precomp, scales, rotations, self.raster_settings)
            if out is not None:
                return out
        # shs [P, M, 3] are evaluated up to raster_settings.sh_degree towards raster_settings.campos (upstream computeColorFromSH); cov3D_precomp
        # [P, 6] replaces scale + rotation.  The reference passes neither (gaussian_renderer/__init__.py:54-62) but constructs the settings
        # with sh_degree = 3 and campos (:46-47): both inputs are part of the module it imports.
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, grad_arena)


    def markVisible(self, positions):
        """Upstream GaussianRasterizer.markVisible: boolean [P] mask of the points that pass the frustum test of the preprocess -- view-space
        z > 0.2, the only test upstream's in_frustum() applies (the reference never calls it; part of the class it imports,
        /root/reference/gaussian_renderer/__init__.py:14).  Not differentiable, like upstream (torch.no_grad)."""
        rs = self.raster_settings
        if not isinstance(positions, torch.Tensor) or not positions.is_cuda:
            raise RuntimeError("gps_gaussian_amd: markVisible needs a GPU tensor of positions (no CPU fallback)")
        if positions.dim() != 2 or positions.shape[1] != 3:
            raise RuntimeError("positions must have dimensions (num_points, 3)")
        with torch.no_grad():
            dev = positions.device
            pos = _prep(positions, "positions", (3,), dev)
            view, proj = _cam(rs.viewmatrix, 16, dev), _cam(rs.projmatrix, 16, dev)
            present = torch.empty((pos.shape[0],), dtype=torch.bool, device=dev)
            with _device_guard(dev):
                rc = _capi.lib().gsr_mark_visible(pos.shape[0], _ptr(pos), _ptr(view), _ptr(proj), _ptr(present), torch._C._cuda_getCurrentRawStream(dev.index))
            _capi.check(rc, "gsr_mark_visible")
        return present


def last_stats(device=None):
    """Capacity-policy state (instances-per-Gaussian ratio seen, capacity floor, bytes of the last workspace = what one view in flight holds)
    for diagnostics."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return dict(_dev_state(dev), pending=len(_dev_state(dev)["pending"]))


def export_state(ws, P, W, H, cap, bin_cap=0):
    """Debug/parity helper: unpack a forward's workspace into tensors: depth, xy, conic_opacity, rect (the 8x8-BIN rect
    bx0,by0,bx1,by1 each Gaussian is listed in), ranges [bx*by, 2] (per-bin list range), point_list, final_T, n_contrib.
    bin_cap: the per-bin capacity the forward used (ctx.bin_cap; 0 = scanned lists) -- with direct lists point_list holds every bin's segment
    (bins x bin_cap entries, indexed by `ranges`)."""
    lib = _capi.lib()
    dev = ws.device
    hdr = ws[:16].view(torch.int64).cpu()
    R = int(hdr[0])
    bx, by = ((W + 7) // 8 + 3) // 4 * 4, (H + 7) // 8   # padded bin grid (gsr_common.h)
    out = dict(
        depth=torch.empty(P, device=dev), xy=torch.empty(P, 2, device=dev), conic_opacity=torch.empty(P, 4, device=dev),
        rect=torch.empty(P, 4, dtype=torch.int32, device=dev), ranges=torch.empty(bx * by, 2, dtype=torch.int64, device=dev),
        point_list=torch.empty(max(bx * by * bin_cap if bin_cap else cap, 1), dtype=torch.int32, device=dev), final_T=torch.empty(H, W, device=dev),
        n_contrib=torch.empty(H, W, dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        rc = lib.gsr_export_state(_ptr(ws), P, W, H, cap, int(bin_cap), _ptr(out["depth"]), _ptr(out["xy"]), _ptr(out["conic_opacity"]),
                                  _ptr(out["rect"]), _ptr(out["ranges"]), _ptr(out["point_list"]), _ptr(out["final_T"]),
                                  _ptr(out["n_contrib"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _capi.check(rc, "gsr_export_state")
    torch.cuda.synchronize(dev)
    if not bin_cap:
        out["point_list"] = out["point_list"][:R]
    out["num_rendered"] = R
    out["bx"], out["by"] = bx, by
    out["overflow"] = int(hdr[1]) & 0xffffffff
    return out
