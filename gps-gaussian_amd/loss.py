"""Fused L1 + SSIM loss for stage-2 training on MI355X (the step right after the rasteriser).

Same definitions as /root/reference/lib/loss.py:36-37 (`l1_loss`) and :40-83 (`ssim`: 11x11 Gaussian window, sigma 1.5, zero
padding, mean over batch/channels/pixels); train_stage2.py:70-72 combines them as 0.8 * L1 + 0.2 * (1 - SSIM).
One forward launch pair and one backward launch replace ~30 eager kernels each way.  Gradient flows to `pred` only (the
ground-truth image has no gradient in the reference either).  No CPU fallback."""
import ctypes as C

import threading

import torch

from . import _capi


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        lib = _capi.lib()
        if not pred.is_cuda:
            raise RuntimeError("gps_gaussian_amd: loss inputs must live on a GPU (no CPU fallback)")
        if pred.shape != gt.shape or pred.dim() < 2:
            raise RuntimeError("pred and gt must have the same [..., H, W] shape")
        p = pred.detach().float().contiguous()
        g = gt.detach().to(device=pred.device, dtype=torch.float32).contiguous()
        H, W = p.shape[-2], p.shape[-1]
        planes = p.numel() // (H * W)
        dev = p.device
        need = ctx.needs_input_grad[0]
        maps = [torch.empty_like(p) for _ in range(3)] if need else [None, None, None]
        scratch = torch.empty((lib.fl_scratch_bytes(planes, H, W) // 4 + 1,), dtype=torch.float32, device=dev)
        out2 = torch.empty((2,), dtype=torch.float32, device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.fl_l1_ssim_forward(ptr(p), ptr(g), planes, H, W, ptr(maps[0]), ptr(maps[1]), ptr(maps[2]), ptr(scratch), ptr(out2),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "fl_l1_ssim_forward")
        if need:
            ctx.save_for_backward(p, g, *maps)
        ctx.dims = (planes, H, W, pred.shape, pred.dtype)
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        lib = _capi.lib()
        p, g, m1, m2, m3 = ctx.saved_tensors
        planes, H, W, shape, dtype = ctx.dims
        dev = p.device
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        gout = torch.stack([(g_l1 if g_l1 is not None else zero).float().reshape(()), (g_ssim if g_ssim is not None else zero).float().reshape(())])
        d = torch.empty_like(p)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.fl_l1_ssim_backward(ptr(p), ptr(g), ptr(m1), ptr(m2), ptr(m3), planes, H, W, ptr(gout), ptr(d),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "fl_l1_ssim_backward")
        return d.view(shape).to(dtype), None


def l1_and_ssim(pred, gt):
    """Returns (mean |pred - gt|, mean SSIM) as 0-d tensors, both differentiable w.r.t. pred."""
    return _L1SSIM.apply(pred, gt)


_shared = threading.local()  # .entry = (weakref pred, pred._version, weakref gt, gt._version, grad mode, (l1, ssim)) of THIS THREAD's latest fused forward


def l1_and_ssim_shared(pred, gt):
    """l1_and_ssim for callers that ask for the two numbers in two calls on the SAME tensors (train_stage2.py:70-71: `l1_loss(render, gt)`
    then `ssim(render, gt)`): the second call gets the other output of the first call's fused forward -- one forward and one backward launch
    pair instead of two.  Same tensor OBJECTS, unchanged since (version counters) and the same grad mode, or it is computed afresh.
    SINGLE USE: the pair is handed out once more and then dropped -- a third evaluation of the same unchanged tensors (e.g. after a
    backward has freed the first pair's graph) computes afresh instead of returning outputs whose graph is gone, and nothing keeps the autograd
    graph and its saved SSIM maps alive between iterations.  Per host thread (two threads computing losses must not cross-talk)."""
    import weakref

    s = getattr(_shared, "entry", None)
    _shared.entry = None
    if (s is not None and s[0]() is pred and s[2]() is gt and s[1] == pred._version and s[3] == gt._version
            and s[4] == torch.is_grad_enabled()):
        return s[5]
    out = _L1SSIM.apply(pred, gt)
    _shared.entry = (weakref.ref(pred), pred._version, weakref.ref(gt), gt._version, torch.is_grad_enabled(), out)
    return out


def l1_loss(network_output, gt):
    return l1_and_ssim(network_output, gt)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11 or not size_average:
        raise NotImplementedError("gps_gaussian_amd: only the configuration the reference uses (window 11, size_average) is fused")
    return l1_and_ssim(img1, img2)[1]


def stage2_photometric_loss(pred, gt, w_l1=0.8, w_ssim=0.2):
    """train_stage2.py:70-72: 0.8 * L1 + 0.2 * (1 - SSIM), from ONE fused forward."""
    l1, s = l1_and_ssim(pred, gt)
    return w_l1 * l1 + w_ssim * (1.0 - s)
