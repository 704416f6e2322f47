"""Fused mask-compaction + pack of the per-pixel Gaussian maps: the step immediately before the rasteriser.

Mirrors what /root/reference/lib/GaussianRender.py:15-34 does per sample with 10 boolean-mask gathers (a device sync each),
5 concats and an affine, as ONE autograd op over the whole batch (3 small kernel launches, no sync inside).  Row order
is the reference's: sample, then view (lmain, rmain), then valid pixels in raster order.
"""
import ctypes as C

import torch

from . import _capi

VIEWS = ("lmain", "rmain")


def _strided(t, kind):
    """Element strides (batch, pixel, channel) of the layouts the reference produces."""
    s = _capi.GsrStrided()
    s.ptr = t.data_ptr()
    if kind == "valid":      # [B, S2] bool
        s.batch_stride, s.pixel_stride, s.channel_stride = t.stride(0), t.stride(1), 0
    elif kind == "xyz":      # [B, S2, 3], usually a permuted view of [B, 3, S2]
        s.batch_stride, s.pixel_stride, s.channel_stride = t.stride(0), t.stride(1), t.stride(2)
    else:                    # [B, C, H, W] with pixels contiguous
        if t.stride(3) != 1 or t.stride(2) != t.shape[3]:
            raise RuntimeError("map must have contiguous pixels")
        s.batch_stride, s.pixel_stride, s.channel_stride = t.stride(0), 1, t.stride(1)
    return s


class _PackViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_views, *tensors):
        # tensors: for each view: xyz, img, rot_maps, scale_maps, opacity_maps, pts_valid
        lib = _capi.lib()
        per = 6
        views = [tensors[i * per:(i + 1) * per] for i in range(n_views)]
        xyz0 = views[0][0]
        if not xyz0.is_cuda:
            raise RuntimeError("gps_gaussian_amd: pack_views inputs must live on a GPU (no CPU fallback)")
        dev = xyz0.device
        B, S2 = xyz0.shape[0], xyz0.shape[1]
        keep = []

        def f32(t):
            t = t.detach()
            if t.dtype != torch.float32:
                t = t.float()
            keep.append(t)
            return t

        arrs = {k: (_capi.GsrStrided * n_views)() for k in ("valid", "xyz", "img", "rot", "scale", "opacity")}
        for v, (xyz, img, rot, scale, opa, valid) in enumerate(views):
            val = valid.detach()
            if val.dtype != torch.bool and val.dtype != torch.uint8:
                val = val != 0
            keep.append(val)
            arrs["valid"][v] = _strided(val.view(torch.uint8) if val.dtype == torch.bool else val, "valid")
            arrs["xyz"][v] = _strided(f32(xyz), "xyz")
            for name, t in (("img", img), ("rot", rot), ("scale", scale), ("opacity", opa)):
                t = f32(t)
                if t.stride(-1) != 1 or t.stride(-2) != t.shape[-1]:
                    t = t.contiguous()
                    keep.append(t)
                arrs[name][v] = _strided(t, "map")
        cap = B * n_views * S2
        out = [torch.empty((cap, c), dtype=torch.float32, device=dev) for c in (3, 3, 4, 3, 1)]
        row_of_pixel = torch.empty((B, n_views, S2), dtype=torch.int32, device=dev)
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        scratch = torch.empty((lib.gsr_pack_scratch_bytes(B, n_views, S2) // 4 + 1,), dtype=torch.int32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.gsr_pack_views(B, n_views, S2, arrs["valid"], arrs["xyz"], arrs["img"], arrs["rot"], arrs["scale"], arrs["opacity"],
                                    p(out[0]), p(out[1]), p(out[2]), p(out[3]), p(out[4]), p(row_of_pixel), p(offsets), p(scratch),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "gsr_pack_views")
        ctx.save_for_backward(row_of_pixel)
        ctx.meta = (n_views, B, S2, [tuple(t.shape) for t in tensors])
        # the packed colours are differentiable only if some image is (in stage 2 none is: the rasteriser backward then skips dL/dcolour)
        img_needs = any(ctx.needs_input_grad[1 + v * per + 1] for v in range(n_views))
        if img_needs:
            ctx.mark_non_differentiable(offsets)
        else:
            ctx.mark_non_differentiable(offsets, out[1])
        return (*out, offsets)

    @staticmethod
    def backward(ctx, g_xyz, g_rgb, g_rot, g_scale, g_op, _g_offsets):
        lib = _capi.lib()
        (row_of_pixel,) = ctx.saved_tensors
        n_views, B, S2, shapes = ctx.meta
        dev = row_of_pixel.device
        per = 6
        needs = ctx.needs_input_grad[1:]
        grads = [None] * (n_views * per)
        ptrs = {k: (C.c_void_p * n_views)() for k in ("xyz", "img", "rot", "scale", "op")}
        for v in range(n_views):
            for j, (name, shape_of) in enumerate((("xyz", (B, S2, 3)), ("img", None), ("rot", None), ("scale", None), ("op", None))):
                if needs[v * per + j]:
                    shp = shape_of or shapes[v * per + j]
                    g = torch.empty(shp, dtype=torch.float32, device=dev)
                    grads[v * per + j] = g
                    ptrs[name][v] = g.data_ptr()
                else:
                    ptrs[name][v] = None
        cg = lambda t: None if t is None else t.contiguous()
        g_xyz, g_rgb, g_rot, g_scale, g_op = map(cg, (g_xyz, g_rgb, g_rot, g_scale, g_op))
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.gsr_pack_views_backward(B, n_views, S2, p(row_of_pixel), p(g_xyz), p(g_rgb), p(g_rot), p(g_scale), p(g_op),
                                             ptrs["xyz"], ptrs["img"], ptrs["rot"], ptrs["scale"], ptrs["op"],
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _capi.check(rc, "gsr_pack_views_backward")
        return (None, *grads)


def pack_views(data, views=VIEWS):
    """data: the reference's nested dict (data[view]['xyz','img','rot_maps','scale_maps','opacity_maps','pts_valid']).
    Returns (xyz, rgb, rot, scale, opacity, offsets): packed rows for the whole batch (capacity B*len(views)*S2 rows; only
    the first offsets[-1] are defined) and the int32 device tensor offsets[B+1]."""
    flat = []
    for v in views:
        d = data[v]
        flat += [d['xyz'], d['img'], d['rot_maps'], d['scale_maps'], d['opacity_maps'], d['pts_valid']]
    return _PackViews.apply(len(views), *flat)
