"""Fused disparity -> inverse depth -> world points.

`flow2gsparms_geometry(data_view)` does, in one kernel each way, what /root/reference/lib/network.py:66-69 does with
flow2depth (lib/utils.py:113-120), depth2pc (:88-110) and `depth != 0`: it returns (depth [B,1,S,S], xyz [B,S*S,3],
pts_valid [B,S*S]) and is differentiable w.r.t. the predicted flow.  Camera parameters are read on the host (a few dozen
floats per sample; in the reference they come from the data loader on the host anyway).  No CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _capi


def _host(t, shape):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32).reshape(shape)
    return a, a.ctypes.data_as(C.c_void_p)


class _Unproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, mask, ref_intr, intr, extr, tf_x):
        lib = _capi.lib()
        if not flow.is_cuda:
            raise RuntimeError("gps_gaussian_amd: unproject inputs must live on a GPU (no CPU fallback)")
        B, _, S, S2_ = flow.shape
        if S != S2_:
            raise RuntimeError("depth2pc assumes square maps")
        dev = flow.device
        f = flow.detach().float().contiguous()
        m = mask.detach().to(device=dev, dtype=torch.float32).contiguous()
        cams = (_host(ref_intr, (B, 9)), _host(intr, (B, 9)), _host(extr[:, :3, :4], (B, 12)), _host(tf_x, (B,)))
        depth = torch.empty((B, 1, S, S), dtype=torch.float32, device=dev)
        xyz = torch.empty((B, S * S, 3), dtype=torch.float32, device=dev)
        valid = torch.empty((B, S * S), dtype=torch.bool, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.up_unproject_forward(B, S, p(f), p(m), m.stride(0), cams[0][1], cams[1][1], cams[2][1], cams[3][1], p(depth), p(xyz),
                                          p(valid), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "up_unproject_forward")
        ctx.save_for_backward(depth, m)
        ctx.cams = cams
        ctx.mark_non_differentiable(valid)
        return depth, xyz, valid

    @staticmethod
    def backward(ctx, g_depth, g_xyz, _g_valid):
        lib = _capi.lib()
        depth, m = ctx.saved_tensors
        B, _, S, _ = depth.shape
        dev = depth.device
        cams = ctx.cams
        gd = g_depth.float().contiguous() if g_depth is not None else None
        gx = g_xyz.float() if g_xyz is not None else None
        d_flow = torch.empty((B, 1, S, S), dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        st = (gx.stride(0), gx.stride(1), gx.stride(2)) if gx is not None else (0, 0, 0)
        with torch.cuda.device(dev):
            rc = lib.up_unproject_backward(B, S, p(depth), p(m), m.stride(0), cams[0][1], cams[1][1], cams[2][1], cams[3][1], p(gd), p(gx),
                                           st[0], st[1], st[2], p(d_flow), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "up_unproject_backward")
        return d_flow, None, None, None, None, None


def unproject(flow_pred, mask, ref_intr, intr, extr, tf_x):
    return _Unproject.apply(flow_pred, mask, ref_intr, intr, extr, tf_x)


def flow2gsparms_geometry(view):
    """view: the reference's per-view dict with 'flow_pred','mask','ref_intr','intr','extr','Tf_x'.
    Fills and returns view['depth'], view['xyz'], view['pts_valid'] like lib/network.py:66-69."""
    view['depth'], view['xyz'], view['pts_valid'] = unproject(view['flow_pred'], view['mask'], view['ref_intr'], view['intr'],
                                                                view['extr'], view['Tf_x'])
    return view
