"""RasterSession -- a torch-autograd-free host for the rasteriser C-ABI (include/gpsgs.h), for callers that drive forward and
backward themselves (inference servers, C++/other-language hosts mirror it 1:1 -- tests/capi_host/capi_host_smoke.cpp -- and
bench.py's headline step).

What it is NOT: a different code path.  It enqueues exactly the kernels `rasterizer._RasterizeGaussians` enqueues, through the same
two C-ABI entry points (gsr_forward_notify, gsr_backward), with the same exact capacity policy (the binning scan publishes the
instance count to pinned host memory; an overflow is repaired by a transparent re-run).  What it leaves out is what PyTorch adds
around them per call: the autograd Function / engine round trip (~120 us per forward+backward on an idle MI355X host, measured
round 3; tools/host_time.py measures the module paths today), tensor allocations (outputs, workspace and gradient buffers are allocated once per (P, W, H)) and
argument normalisation.  Host cost per forward+backward: two ctypes calls (~35 us, almost all of it the HIP launches themselves).

Semantics follow the reference call at /root/reference/gaussian_renderer/__init__.py:36-62 (precomputed colours, scales + rotations).
"""
import ctypes as C

import torch

from . import _capi
from . import rasterizer as RZ


class RasterSession:
    def __init__(self, P, width, height, device, training=True, wave_priority=True):
        self.P, self.W, self.H = int(P), int(width), int(height)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("gps_gaussian_amd: RasterSession needs a GPU device (no CPU path exists)")
        self.training = bool(training)
        # GSR_FLAG_WAVE_PRIORITY (include/gpsgs.h): on for a session that renders on its own; a caller that keeps several sessions in
        # flight on different streams passes False (the kernels of different views then share the SIMDs and the scheme costs ~2 %)
        self.wave_priority = bool(wave_priority)
        self.lib = _capi.lib()
        d, f32 = self.dev, torch.float32
        self.color = torch.empty((3, self.H, self.W), dtype=f32, device=d)
        self.radii = torch.empty((self.P,), dtype=torch.int32, device=d)
        self.ws, self.cap, self.nbytes = None, 0, 0
        self.allocations = 0  # workspace (re)allocations so far: 1 in steady state
        self._in, self._pending, self._cur = None, None, None
        self._ws_streams = []  # streams that have launched on the current workspace (see _ensure_ws)
        if self.training:
            # one buffer, six contiguous gradient arrays carved out of it
            P_ = max(self.P, 1)
            self._gbuf = torch.empty((P_ * 17 + 64,), dtype=f32, device=d)
            o, self.grads = 0, {}
            for name, c in (("rotations", 4), ("means3D", 3), ("means2D", 3), ("colors", 3), ("scales", 3), ("opacities", 1)):
                o = (o + 3) // 4 * 4  # 16-byte alignment (the quaternion gradient is stored as float4)
                self.grads[name] = self._gbuf[o:o + self.P * c].view(self.P, c)
                o += P_ * c
        self.bin_cap = 0  # per-bin capacity of the current workspace's lists (0 = scanned lists; rasterizer._bin_capacity)
        self._scanned_only = False  # the current view has shown a list too long for direct bins: its repair run uses scanned lists

    def _ensure_ws(self, cap, stream, bin_cap=0):
        """The workspace for `cap` instances, allocated on `stream` -- the stream its kernels are launched on.  (allocated under PyTorch's
        ambient current stream, the block belonged to THAT stream's pool while its kernels ran on `stream`; on replacement it went straight back to the
        pool and a later ambient-stream allocation could reuse it under the in-flight forward / backward.)  A replaced workspace is handed back with
        record_stream() for every stream the session ever launched on it, so the allocator waits for those kernels before reusing the block."""
        if self.ws is None or cap != self.cap or bin_cap != self.bin_cap:
            old = self.ws
            if old is not None:
                for st in self._ws_streams:
                    old.record_stream(st)
            self.nbytes = self.lib.gsr_workspace_bytes_ex(self.P, self.W, self.H, cap, bin_cap, 0 if self.training else 1)
            self.bin_cap = bin_cap
            with torch.cuda.stream(stream):
                self.ws = torch.empty((self.nbytes,), dtype=torch.uint8, device=self.dev)
            self._ws_streams = [stream]
            self.cap = cap
            self.allocations += 1
        elif not any(st is stream or st == stream for st in self._ws_streams):
            self._ws_streams.append(stream)

    @staticmethod
    def _chk(t, n, name):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("gps_gaussian_amd: RasterSession expects %s as a contiguous fp32 GPU tensor of %d elements" % (name, n))
        return t.data_ptr()

    def forward(self, means3D, colors, opacities, scales, rotations, viewmatrix, projmatrix, bg, tanfovx, tanfovy, scale_modifier=1.0, stream=None):
        """-> (color[3,H,W], radii[P]) (the session's own buffers: valid until the next forward).  Enqueues on `stream` (a
        torch.cuda.Stream), default: PyTorch's current stream."""
        self.forward_begin(means3D, colors, opacities, scales, rotations, viewmatrix, projmatrix, bg, tanfovx, tanfovy, scale_modifier, stream)
        return self.forward_end()

    def forward_begin(self, means3D, colors, opacities, scales, rotations, viewmatrix, projmatrix, bg, tanfovx, tanfovy, scale_modifier=1.0, stream=None):
        """First half of forward(): enqueue the forward kernels and return at once.  forward_end() then waits for the capacity
        notification (and repairs an overflow).  Several sessions on several streams can so be started back to back -- independent views
        rendered concurrently -- before the host waits for any of them.  stream: the torch.cuda.Stream to enqueue on (the session only
        hands its raw handle to the C-ABI and owns all its buffers, so PyTorch's current stream need not be switched: ~8 us of host time per
        `with torch.cuda.stream(...)` saved, three times per view); default: the current stream.  The INPUTS (and, for backward(), dL_dpix)
        must already be ordered against that stream by the caller -- the session never inserts a wait; its own workspace is allocated on, and
        tied to, the streams it launches on."""
        P, W, H, lib = self.P, self.W, self.H, self.lib
        ptrs = (self._chk(means3D, 3 * P, "means3D"), self._chk(colors, 3 * P, "colors"), self._chk(opacities, P, "opacities"),
                self._chk(scales, 3 * P, "scales"), self._chk(rotations, 4 * P, "rotations"))
        cam = (self._chk(viewmatrix, 16, "viewmatrix"), self._chk(projmatrix, 16, "projmatrix"), self._chk(bg, 3, "bg"))
        fl = (float(scale_modifier), float(tanfovx), float(tanfovy))
        family = RZ._composite_flag() | (RZ._wave_priority_flag() if self.wave_priority else 0)
        self._in = (ptrs, fl, cam, family, (means3D, colors, opacities, scales, rotations, viewmatrix, projmatrix, bg))  # keeps the inputs alive
        st = RZ._dev_state(self.dev)
        self._cur = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self._scanned_only = False
        self._enqueue(max(self.cap, RZ._capacity_for(st, P)))

    def _enqueue(self, cap):
        P, W, H, lib = self.P, self.W, self.H, self.lib
        ptrs, fl, cam, family, _ = self._in
        st = RZ._dev_state(self.dev)
        stream = self._cur.cuda_stream
        flags = RZ._extra_flags | family
        self._ensure_ws(cap, self._cur, 0 if (P == 0 or self._scanned_only) else RZ._bin_capacity(st, W, H))
        if P == 0:
            _capi.check(lib.gsr_forward(P, W, H, *ptrs, *fl, *cam, self.color.data_ptr(), self.radii.data_ptr(), self.ws.data_ptr(),
                                        self.nbytes, cap, flags, stream), "gsr_forward")
            self._pending = None
            return
        ring = RZ._ring(self.dev)
        slot, hdr, w32, hdr_ptr, seq = ring.acquire_notify()
        skip_large = not st.get("big_bins", False)
        f = (flags & ~_capi.GSR_FLAG_NO_LARGE_SORT) | (_capi.GSR_FLAG_NO_LARGE_SORT if skip_large else 0)
        ext = RZ._ext(None, st.get("longest", 0), None, self.bin_cap)  # work order: longest lists first, relative to the longest list seen on this device
        rc = lib.gsr_forward_ex(P, W, H, *ptrs, *fl, *cam, self.color.data_ptr(), self.radii.data_ptr(), self.ws.data_ptr(),
                                self.nbytes, cap, f, stream, hdr_ptr, seq, C.byref(ext))
        if rc != 0:
            ring.release(slot)
        _capi.check(rc, "gsr_forward_ex")
        self._pending = (slot, hdr, w32, seq)

    def forward_end(self):
        """Second half of forward(): wait until the binning scan has published the instance count (typically well before the forward has
        finished), re-render with a larger workspace if it overflowed.  -> (color, radii)"""
        st = RZ._dev_state(self.dev)
        while self._pending is not None:
            slot, hdr, w32, seq = self._pending
            try:
                RZ._wait_notify(w32, seq, self._cur)
                R, overflow, need = RZ._decode(hdr)
                longest = int(w32[3])
            finally:
                RZ._ring(self.dev).release(slot)
            RZ._learn(st, R, need, self.P, longest)
            RZ._note_longest(st, longest)
            if not overflow:
                self._pending = None
                break
            if self.bin_cap and longest > self.bin_cap:
                self._scanned_only = True  # a list that does not fit a direct bin: the re-run below uses scanned lists
            elif self.cap >= 0x7fffffff:
                raise RuntimeError("gps_gaussian_amd: this view needs %d (Gaussian, bin) instances, more than the 2^31 - 1 the workspace layout can address" % R)
            self._enqueue(RZ._capacity_for(st, self.P))  # a larger workspace (allocated on the view's own stream); the in-flight kernels of the failed attempt exit at once on the overflow flag
        return self.color, self.radii

    def backward(self, dL_dpix, color_grad=True, stream=None):
        """dL_dpix[3,H,W] fp32 contiguous -> dict of gradient tensors (the session's own buffers: valid until the next backward).
        color_grad=False: dL/dcolours is not needed (stage 2: the colours are input pixels) -- GSR_FLAG_NO_COLOR_GRAD, its buffer holds zeros."""
        if not self.training:
            raise RuntimeError("gps_gaussian_amd: this RasterSession was created with training=False (no backward workspace)")
        if self._in is None:
            raise RuntimeError("gps_gaussian_amd: RasterSession.backward() without a forward()")
        ptrs, fl, cam, family, _keep = self._in
        g = self._chk(dL_dpix, 3 * self.W * self.H, "dL_dpix")
        G = self.grads
        bstream = stream if stream is not None else torch.cuda.current_stream(self.dev)
        if not any(st is bstream or st == bstream for st in self._ws_streams):
            self._ws_streams.append(bstream)  # (the caller orders this stream against the forward's; the workspace must outlive both)
        if self.P > 0:
            ext = RZ._ext(None, 0, None, self.bin_cap)
            rc = self.lib.gsr_backward_ex(self.P, self.W, self.H, *ptrs, *fl, *cam, self.radii.data_ptr(), g, G["means3D"].data_ptr(),
                                          G["means2D"].data_ptr(), G["colors"].data_ptr(), G["opacities"].data_ptr(), G["scales"].data_ptr(),
                                          G["rotations"].data_ptr(), self.ws.data_ptr(), self.nbytes, self.cap,
                                          RZ._extra_flags | family | (0 if color_grad else _capi.GSR_FLAG_NO_COLOR_GRAD),
                                          bstream.cuda_stream, C.byref(ext))
            _capi.check(rc, "gsr_backward_ex")
        return G
