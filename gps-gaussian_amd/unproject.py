"""Fused disparity -> inverse depth -> world points.

`flow2gsparms_geometry(data_view)` does, in one kernel each way, what /root/reference/lib/network.py:66-69 does with
flow2depth (lib/utils.py:113-120), depth2pc (:88-110) and `depth != 0`: it returns (depth [B,1,S,S], xyz [B,S*S,3],
pts_valid [B,S*S]) and is differentiable w.r.t. the predicted flow.  The camera parameters (31 floats per sample) STAY ON THE DEVICE: they
are gathered into one [B,31] tensor by a torch.cat and the kernels read them from there (up_unproject_*_dev) -- the reference's scripts
keep them on the GPU (train_stage2.py:154-156), and reading them back (round 3) put a device synchronisation in the middle of every network forward.
No CPU fallback."""
import ctypes as C

import torch

from . import _capi


def _cams(ref_intr, intr, extr, tf_x, B, dev):
    """[B,31] fp32 on the device: ref_intr 3x3 | intr 3x3 | extr rows 0..2 (3x4) | Tf_x.  Host tensors are copied over (no sync either)."""
    parts = (ref_intr.reshape(B, 9), intr.reshape(B, 9), extr[:, :3, :4].reshape(B, 12), tf_x.reshape(B, 1))
    return torch.cat([p.detach().to(device=dev, dtype=torch.float32, non_blocking=True) for p in parts], dim=1).contiguous()


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
        cams = _cams(ref_intr, intr, extr, tf_x, B, dev)
        depth = torch.empty((B, 1, S, S), dtype=torch.float32, device=dev)
        xyz = torch.empty((B, S * S, 3), dtype=torch.float32, device=dev)
        valid = torch.empty((B, S * S), dtype=torch.bool, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.up_unproject_forward_dev(B, S, p(f), p(m), m.stride(0), p(cams), p(depth), p(xyz), p(valid),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "up_unproject_forward_dev")
        ctx.save_for_backward(depth, m, cams)
        ctx.mark_non_differentiable(valid)
        return depth, xyz, valid

    @staticmethod
    def backward(ctx, g_depth, g_xyz, _g_valid):
        lib = _capi.lib()
        depth, m, cams = ctx.saved_tensors
        B, _, S, _ = depth.shape
        dev = depth.device
        gd = g_depth.float().contiguous() if g_depth is not None else None
        gx = g_xyz.float() if g_xyz is not None else None
        d_flow = torch.empty((B, 1, S, S), dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        st = (gx.stride(0), gx.stride(1), gx.stride(2)) if gx is not None else (0, 0, 0)
        with torch.cuda.device(dev):
            rc = lib.up_unproject_backward_dev(B, S, p(depth), p(m), m.stride(0), p(cams), p(gd), p(gx), st[0], st[1], st[2], p(d_flow),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "up_unproject_backward_dev")
        return d_flow, None, None, None, None, None


def unproject(flow_pred, mask, ref_intr, intr, extr, tf_x):
    return _Unproject.apply(flow_pred, mask, ref_intr, intr, extr, tf_x)


def flow2gsparms_geometry(view):
    """view: the reference's per-view dict with 'flow_pred','mask','ref_intr','intr','extr','Tf_x'.
    Fills and returns view['depth'], view['xyz'], view['pts_valid'] like lib/network.py:66-69."""
    view['depth'], view['xyz'], view['pts_valid'] = unproject(view['flow_pred'], view['mask'], view['ref_intr'], view['intr'],
                                                                view['extr'], view['Tf_x'])
    return view
