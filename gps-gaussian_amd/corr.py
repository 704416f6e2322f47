"""Drop-in replacement for the `corr_sampler` extension of RAFT-Stereo on MI355X.

Interface bound by the reference at /root/reference/core/corr.py:22,28:
    corr, = corr_sampler.forward(volume[N,H1,W1,W2], coords[N,1,H1,W1], radius)          -> corr[N,2r+1,H1,W1]
    grad_volume, = corr_sampler.backward(volume, coords, grad_output[N,2r+1,H1,W1], radius)
Both return 1-tuples.  fp32 and fp16 volumes (stage 2 runs the update block under AMP).  No CPU fallback.
"""
import ctypes as C

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
    if coords.shape[0] != N or tuple(coords.shape[2:]) != (H1, W1):
        raise RuntimeError("coords shape %s does not match volume %s" % (tuple(coords.shape), tuple(volume.shape)))
    c = coords[:, 0].to(dtype=torch.float32).contiguous()
    return N, H1, W1, W2, c


def forward(volume, coords, radius):
    lib = _capi.lib()
    N, H1, W1, W2, c = _args(volume, coords)
    v = volume.contiguous()
    out = torch.empty((N, 2 * radius + 1, H1, W1), dtype=v.dtype, device=v.device)
    with torch.cuda.device(v.device):
        rc = lib.cs_forward(C.c_void_p(v.data_ptr()), C.c_void_p(c.data_ptr()), C.c_void_p(out.data_ptr()), N, H1, W1, W2,
                            int(radius), _DT[v.dtype], C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream))
    _capi.check(rc, "cs_forward")
    return (out,)


def backward(volume, coords, grad_output, radius):
    lib = _capi.lib()
    N, H1, W1, W2, c = _args(volume, coords)
    g = grad_output.to(dtype=volume.dtype).contiguous()
    if tuple(g.shape) != (N, 2 * radius + 1, H1, W1):
        raise RuntimeError("grad_output must be [N,2r+1,H1,W1]")
    gv = torch.empty((N, H1, W1, W2), dtype=volume.dtype, device=volume.device)
    with torch.cuda.device(volume.device):
        rc = lib.cs_backward(C.c_void_p(c.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(gv.data_ptr()), N, H1, W1, W2,
                             int(radius), _DT[volume.dtype], C.c_void_p(torch.cuda.current_stream(volume.device).cuda_stream))
    _capi.check(rc, "cs_backward")
    return (gv,)
