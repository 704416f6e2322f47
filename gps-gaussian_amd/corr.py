"""Drop-in replacement for the `corr_sampler` extension of RAFT-Stereo on MI355X.

Interface bound by the reference at /root/reference/core/corr.py:22,28:
    corr, = corr_sampler.forward(volume[N,H1,W1,W2], coords[N,1,H1,W1], radius)          -> corr[N,2r+1,H1,W1]
    grad_volume, = corr_sampler.backward(volume, coords, grad_output[N,2r+1,H1,W1], radius)
Both return 1-tuples.  fp32 and fp16 volumes (stage 2 runs the update block under AMP).  No CPU fallback.
"""
import ctypes as C
import sys

import torch

from . import _capi

_DT = {torch.float32: 0, torch.float16: 1}


def _args(volume, coords):
    if not volume.is_cuda:
        raise RuntimeError("gps_gaussian_amd: corr_sampler inputs must live on a GPU (no CPU fallback)")
    if volume.dim() != 4 or coords.dim() != 4 or coords.shape[1] < 1:
        raise RuntimeError("volume must be [N,H1,W1,W2] and coords [N,1,H1,W1]")
    if volume.dtype not in _DT:
        raise RuntimeError("corr_sampler supports float32 and float16 volumes, got %s" % volume.dtype)
    N, H1, W1, W2 = volume.shape
    if coords.shape[0] != N or tuple(coords.shape[2:]) != (H1, W1) or coords.device != volume.device:
        raise RuntimeError("coords shape %s does not match volume %s" % (tuple(coords.shape), tuple(volume.shape)))
    # what core/corr.py:46-49 passes -- `coords[:, [0]] / 2**i`: fp32, contiguous, ONE channel -- is used as is ([N,1,H1,W1] and [N,H1,W1]
    # are the same memory); anything else is brought into that form
    if coords.dtype is torch.float32 and coords.shape[1] == 1 and coords.is_contiguous():
        c = coords
    else:
        c = coords[:, 0].to(dtype=torch.float32).contiguous()
    return N, H1, W1, W2, c


_lib = None


def _call(fn, what, dev, *args):
    """One C-ABI call on the current stream of `dev`; the device guard (~10 us) is taken only when `dev` is not the current device."""
    if torch.cuda.current_device() == dev.index:
        rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream)
    else:
        with torch.cuda.device(dev):
            rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        _capi.check(rc, what)


def forward(volume, coords, radius):
    global _lib
    lib = _lib
    if lib is None:
        lib = _lib = _capi.lib()
        acc = sys.modules.get("gps_gaussian_amd.accelerate")  # first sampler call of the process: the opt-in hook's safety net (accelerate.py)
        if acc is not None and acc._armed:
            acc.late_apply()
    N, H1, W1, W2, c = _args(volume, coords)
    v = volume if volume.is_contiguous() else volume.contiguous()
    out = torch.empty((N, 2 * radius + 1, H1, W1), dtype=v.dtype, device=v.device)
    # plain integers: ctypes converts them to the void* / int arguments declared in _capi (boxing them in c_void_p costs ~1 us each)
    _call(lib.cs_forward, "cs_forward", v.device, v.data_ptr(), c.data_ptr(), out.data_ptr(), N, H1, W1, W2, int(radius), _DT[v.dtype])
    return (out,)


def backward(volume, coords, grad_output, radius):
    global _lib
    lib = _lib or _capi.lib()
    _lib = lib
    N, H1, W1, W2, c = _args(volume, coords)
    g = grad_output
    if g.dtype is not volume.dtype or not g.is_contiguous():
        g = g.to(dtype=volume.dtype).contiguous()
    if tuple(g.shape) != (N, 2 * radius + 1, H1, W1):
        raise RuntimeError("grad_output must be [N,2r+1,H1,W1]")
    gv = torch.empty((N, H1, W1, W2), dtype=volume.dtype, device=volume.device)
    _call(lib.cs_backward, "cs_backward", volume.device, c.data_ptr(), g.data_ptr(), gv.data_ptr(), N, H1, W1, W2, int(radius), _DT[volume.dtype])
    return (gv,)


# ---- the sampler's neighbours inside RAFT-Stereo -------------------------------------------------
def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr_array(tensors):
    return (C.c_void_p * 4)(*[C.c_void_p(t.data_ptr()) if t is not None else None for t in list(tensors) + [None] * (4 - len(tensors))])


class _BuildPyramid(torch.autograd.Function):
    """fmap1[N,D,H,W1], fmap2[N,D,H,W2] -> `levels` volumes [N,H,W1,W2>>l] (cv_build_forward / cv_build_backward)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, levels):
        lib = _capi.lib()
        if not fmap1.is_cuda:
            raise RuntimeError("gps_gaussian_amd: correlation volume inputs must live on a GPU (no CPU fallback)")
        if fmap1.dim() != 4 or fmap2.dim() != 4 or fmap1.shape[:3] != fmap2.shape[:3]:
            raise RuntimeError("fmap1 / fmap2 must be [N,D,H,W1] / [N,D,H,W2]")
        if fmap1.dtype not in _DT or fmap2.dtype != fmap1.dtype:
            raise RuntimeError("correlation volume supports float32 and float16 feature maps of one dtype")
        if not 1 <= levels <= 4:
            raise RuntimeError("1..4 pyramid levels")
        f1, f2 = fmap1.contiguous(), fmap2.contiguous()
        N, D, H, W1 = f1.shape
        W2 = f2.shape[3]
        pyr = [torch.empty((N, H, W1, W2 >> l), dtype=f1.dtype, device=f1.device) for l in range(levels)]
        with torch.cuda.device(f1.device):
            rc = lib.cv_build_forward(C.c_void_p(f1.data_ptr()), C.c_void_p(f2.data_ptr()), _ptr_array(pyr), N, D, H, W1, W2, levels,
                                      _DT[f1.dtype], _stream(f1.device))
        _capi.check(rc, "cv_build_forward")
        ctx.save_for_backward(f1, f2)
        ctx.levels = levels
        ctx.set_materialize_grads(False)
        return tuple(pyr)

    @staticmethod
    def backward(ctx, *grads):
        lib = _capi.lib()
        f1, f2 = ctx.saved_tensors
        N, D, H, W1 = f1.shape
        W2 = f2.shape[3]
        gs = [g.to(dtype=f1.dtype).contiguous() if g is not None else None for g in grads]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if all(g is None for g in gs) or not (need1 or need2):
            return None, None, None
        g1 = torch.empty_like(f1) if need1 else None
        g2 = torch.empty_like(f2) if need2 else None
        with torch.cuda.device(f1.device):
            rc = lib.cv_build_backward(C.c_void_p(f1.data_ptr()), C.c_void_p(f2.data_ptr()), _ptr_array(gs),
                                       C.c_void_p(g1.data_ptr()) if need1 else None, C.c_void_p(g2.data_ptr()) if need2 else None, N, D, H, W1,
                                       W2, ctx.levels, _DT[f1.dtype], _stream(f1.device))
        _capi.check(rc, "cv_build_backward")
        return g1, g2, None


class _LookupPyramid(torch.autograd.Function):
    """(coords[N,1,H,W1], radius, *volumes) -> [N, levels*(2r+1), H, W1] (cs_lookup_forward / cs_lookup_backward)."""

    @staticmethod
    def forward(ctx, coords, radius, *pyr):
        lib = _capi.lib()
        v0 = pyr[0]
        N, H1, W1, W2, c = _args(v0, coords)
        for l, v in enumerate(pyr):
            if tuple(v.shape) != (N, H1, W1, W2 >> l) or v.dtype != v0.dtype or not v.is_contiguous():
                raise RuntimeError("pyramid level %d must be a contiguous [N,H1,W1,W2>>%d] tensor of the level-0 dtype" % (l, l))
        out = torch.empty((N, len(pyr) * (2 * radius + 1), H1, W1), dtype=v0.dtype, device=v0.device)
        with torch.cuda.device(v0.device):
            rc = lib.cs_lookup_forward(_ptr_array(pyr), C.c_void_p(c.data_ptr()), C.c_void_p(out.data_ptr()), N, H1, W1, W2, len(pyr),
                                       int(radius), _DT[v0.dtype], _stream(v0.device))
        _capi.check(rc, "cs_lookup_forward")
        ctx.save_for_backward(c)
        ctx.meta = (N, H1, W1, W2, len(pyr), int(radius), v0.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _capi.lib()
        c, = ctx.saved_tensors
        N, H1, W1, W2, levels, radius, dt = ctx.meta
        g = grad_out.to(dtype=dt).contiguous()
        gp = [torch.empty((N, H1, W1, W2 >> l), dtype=dt, device=g.device) for l in range(levels)]
        with torch.cuda.device(g.device):
            rc = lib.cs_lookup_backward(C.c_void_p(c.data_ptr()), C.c_void_p(g.data_ptr()), _ptr_array(gp), N, H1, W1, W2, levels, radius,
                                        _DT[dt], _stream(g.device))
        _capi.check(rc, "cs_lookup_backward")
        return (None, None) + tuple(gp)


class CorrBlockFast1D:
    """Mirror of the reference's CorrBlockFast1D (/root/reference/core/corr.py:31-61): same constructor and call, but the volume and
    its pyramid come from ONE fused kernel and every call samples all levels in ONE launch (the reference: einsum + divide +
    3 avg_pool2d, then 4 sampler launches + torch.cat per GRU iteration)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        pyr = _BuildPyramid.apply(fmap1, fmap2, num_levels)
        self.volumes = list(pyr)
        # same shapes the reference exposes: [batch, h1, w1, 1, w2 >> i]
        self.corr_pyramid = [v.unsqueeze(3) for v in self.volumes]

    def __call__(self, coords):
        return _LookupPyramid.apply(coords[:, [0]], self.radius, *self.volumes)

    @staticmethod
    def corr(fmap1, fmap2):
        v, = _BuildPyramid.apply(fmap1, fmap2, 1)
        return v.unsqueeze(3)


class _ConvexUpsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, mask, factor):
        lib = _capi.lib()
        if not flow.is_cuda:
            raise RuntimeError("gps_gaussian_amd: upsample_flow inputs must live on a GPU (no CPU fallback)")
        N, Cc, H, W = flow.shape
        if mask.shape[0] != N or mask.shape[1] != 9 * factor * factor or tuple(mask.shape[2:]) != (H, W) or not 1 <= Cc <= 2:
            raise RuntimeError("flow must be [N,C<=2,H,W] and mask [N,9*factor^2,H,W]")
        fl = flow.detach().to(torch.float32).contiguous()
        mk = mask.detach().to(torch.float32).contiguous()  # AMP hands over an fp16 mask; the softmax is evaluated in fp32
        out = torch.empty((N, Cc, H * factor, W * factor), dtype=torch.float32, device=flow.device)
        with torch.cuda.device(flow.device):
            rc = lib.cu_upsample_forward(C.c_void_p(fl.data_ptr()), C.c_void_p(mk.data_ptr()), C.c_void_p(out.data_ptr()), N, Cc, H, W, int(factor),
                                         _stream(flow.device))
        _capi.check(rc, "cu_upsample_forward")
        ctx.save_for_backward(fl, mk)
        ctx.factor = int(factor)
        ctx.dtypes = (flow.dtype, mask.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _capi.lib()
        fl, mk = ctx.saved_tensors
        N, Cc, H, W = fl.shape
        g = grad_out.to(torch.float32).contiguous()
        gf = torch.empty_like(fl) if ctx.needs_input_grad[0] else None
        gm = torch.empty_like(mk) if ctx.needs_input_grad[1] else None
        if gf is None and gm is None:
            return None, None, None
        scratch = torch.empty((lib.cu_upsample_scratch_bytes(N, Cc, H, W),), dtype=torch.uint8, device=fl.device) if gf is not None else None
        with torch.cuda.device(fl.device):
            rc = lib.cu_upsample_backward(C.c_void_p(fl.data_ptr()), C.c_void_p(mk.data_ptr()), C.c_void_p(g.data_ptr()),
                                          C.c_void_p(gf.data_ptr()) if gf is not None else None,
                                          C.c_void_p(gm.data_ptr()) if gm is not None else None,
                                          C.c_void_p(scratch.data_ptr()) if scratch is not None else None, N, Cc, H, W, ctx.factor,
                                          _stream(fl.device))
        _capi.check(rc, "cu_upsample_backward")
        return (gf.to(ctx.dtypes[0]) if gf is not None else None, gm.to(ctx.dtypes[1]) if gm is not None else None, None)


def upsample_flow(flow, mask, factor):
    """RAFTStereoHuman.upsample_flow (/root/reference/core/raft_stereo_human.py:69-81) with factor = 2 ** n_downsample:
    [N,D,H,W] -> [N,D,factor*H,factor*W] by the learned convex combination of the 3x3 coarse neighbours."""
    return _ConvexUpsample.apply(flow, mask, factor)


class CorrSampler(torch.autograd.Function):
    """Mirror of the reference's autograd wrapper around the sampler extension (/root/reference/core/corr.py:17-29)."""

    @staticmethod
    def forward(ctx, volume, coords, radius):
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        corr, = forward(volume, coords, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        volume, coords = ctx.saved_tensors
        grad_volume, = backward(volume, coords, grad_output.contiguous(), ctx.radius)
        return grad_volume, None, None
