"""render(data, idx, ...) and pts2render(data, bg_color) with the reference's signatures and semantics.

Host-side mirror of /root/reference/gaussian_renderer/__init__.py:17-67 (`render`) and
/root/reference/lib/GaussianRender.py:6-40 (`pts2render`).  The reference's own two files also run unmodified against
the drop-in `diff_gaussian_rasterization` shim; these mirrors exist so that callers (bench, tests, the DDP launcher)
do not need /root/reference on the path, and they read the camera scalars without per-sample device syncs when the
tensors already live on the host.
"""
import math
import os

import torch

from . import rasterizer as _RZ
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

import threading

_tls = threading.local()  # per host thread: device index -> side streams (the samples of a batch are independent views and render concurrently)


def _streams(dev, n):
    pools = getattr(_tls, "pools", None)
    if pools is None:
        pools = _tls.pools = {}
    pool = pools.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


_bg_cache = {}


def _bg_tensor(bg_color, dev):
    """The background colour on the device.  torch.tensor(list, device=cuda) -- what the reference does per render() call,
    gaussian_renderer/__init__.py:23 -- is a SYNCHRONOUS copy from pageable memory: it blocks the host until the stream has drained
    (measured 0.35 ms per stage-2 iteration, and it stops the host from running ahead of the GPU).  The handful of distinct colours a
    run uses are kept on the device instead."""
    if isinstance(bg_color, torch.Tensor):
        return bg_color.to(device=dev, dtype=torch.float32)
    key = (tuple(float(c) for c in bg_color), dev.index)
    t = _bg_cache.get(key)
    if t is None:
        if len(_bg_cache) > 64:
            _bg_cache.clear()
        t = _bg_cache[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
    return t


def _scalar(x):
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


class _SplitRows(torch.autograd.Function):
    """packed[rows, C] -> row chunks, like torch.split -- but if the chunk gradients come back as the matching consecutive row
    slices of ONE buffer (the rasteriser backward was handed such slices: grad_arena), that buffer IS the gradient of `packed`
    and nothing is copied.  Otherwise falls back to what SplitBackward does: one concatenation."""

    @staticmethod
    def forward(ctx, packed, sizes, arena):
        ctx.sizes = sizes
        ctx.arena = arena
        ctx.shape = tuple(packed.shape)
        ctx.set_materialize_grads(False)
        return tuple(c.view_as(c) for c in packed.detach().split(sizes))

    @staticmethod
    def backward(ctx, *grads):
        arena = ctx.arena
        if arena is not None:
            off, ok = 0, True
            for g, n in zip(grads[:-1], ctx.sizes[:-1]):  # the last chunk is the unused tail of the packed capacity
                if n and (g is None or g.data_ptr() != arena.data_ptr() + off * arena.stride(0) * 4 or g.shape[0] != n):
                    ok = False
                    break
                off += n
            if ok:
                return arena, None, None
        parts = [g if g is not None else torch.zeros((n,) + ctx.shape[1:], dtype=torch.float32, device=arena.device if arena is not None else None)
                 for g, n in zip(grads, ctx.sizes)]
        return torch.cat(parts, dim=0), None, None


class _Holder:
    """Attribute holder handed to rasterizer._forward_impl / _backward_impl in place of an autograd ctx (one per view of a batch)."""
    pass


class _RenderBatch(torch.autograd.Function):
    """All samples of a pts2render batch as ONE autograd node: B raster forwards enqueued back to back on B HIP streams (the views
    are independent and one view leaves the chip under-occupied: DESIGN.md section 4), their exact capacity checks collected and run
    once all are in flight, the images written straight into one [B,3,H,W] tensor; the backward does the same with the B raster
    backwards, each writing its rows of six batch-wide gradient buffers.  Replaces, per iteration, 5 split nodes + B rasteriser
    nodes + one concatenation (and their Python), which is what kept the per-sample form host-bound (tools/stage2_ab.py).

    The HOST NEVER LEARNS how many valid pixels a sample has: every view is handed the packed batch-wide arrays plus a DEVICE pointer
    to its {begin, end} row range (GsrViewExt.row_range; the pack kernels' offsets tensor) and a row capacity; the kernels read the range
    themselves.  The reference learns those counts through ten boolean-mask gathers per sample, each a device sync
    (lib/GaussianRender.py:15-34); the round-2 form of this class still read the B + 1 offsets back (one sync per batch).
    Numerically it IS the per-sample path: the same kernels on the same rows (tests/test_gpu_pack.py compares the bits)."""

    @staticmethod
    def forward(ctx, xyz, rgb, rot, scale, opacity, offsets, settings, cap_rows):
        # xyz .. opacity: packed [N, C] fp32 (pack.pack_views); offsets: B + 1 row offsets, int32 ON THE DEVICE; settings: B
        # GaussianRasterizationSettings; cap_rows: upper bound of a sample's rows (views x pixels)
        bs = len(settings)
        dev = xyz.device
        H, W = int(settings[0].image_height), int(settings[0].image_width)
        out = torch.empty((bs, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((xyz.shape[0],), dtype=torch.int32, device=dev)  # batch-wide, like the inputs
        cur = torch.cuda.current_stream(dev)
        # one HIP stream per sample -- except under graph capture (GPSGS_CHECK=none), where everything stays on the capturing stream
        side = _streams(dev, bs) if (bs > 1 and not torch.cuda.is_current_stream_capturing()) else [cur] * bs
        needs = any(ctx.needs_input_grad[:5])
        views = []
        with _RZ.defer_capacity_checks():
            for i in range(bs):
                h = _Holder()
                if side[i] is not cur:
                    side[i].wait_stream(cur)
                with torch.cuda.stream(side[i]):
                    _RZ._forward_impl(h, xyz, rgb, opacity, scale, rot, settings[i], needs, out_color=out[i],
                                      rows=_RZ._Rows(offsets, i, cap_rows), radii_out=radii)
                views.append(h)
        for i in range(bs):
            if side[i] is not cur:
                cur.wait_stream(side[i])
        # the packed inputs go through save_for_backward (autograd then notices an in-place change between forward and backward); the
        # per-view holders keep only what is not an input: camera matrices, background, radii, workspace.  (pack_views hands out
        # contiguous fp32, which _forward_impl uses as is; should an input ever have been converted on the way in, the holder keeps the
        # converted tensors instead.)
        packed = (xyz, rgb, opacity, scale, rot)  # order of _forward_impl's saved tuple: m3, col, opa, sca, rot
        for h in views:
            same = all(sv.data_ptr() == t.data_ptr() and sv.numel() == t.numel() for sv, t in zip(h.saved[:5], packed))
            h.tail, h.own = h.saved[5:], (None if same else h.saved[:5])
            h.saved = None
        ctx.save_for_backward(xyz, rgb, rot, scale, opacity)
        ctx.views, ctx.side = views, side
        ctx.color_grad = bool(ctx.needs_input_grad[1])  # False in stage 2: pack_views marks rgb non-differentiable when no image needs a gradient
        ctx.shapes = tuple(tuple(t.shape) for t in (xyz, rgb, rot, scale, opacity))
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return (None,) * 8
        views, side = ctx.views, ctx.side
        xyz, rgb, rot, scale, opacity = ctx.saved_tensors
        dev = gout.device
        g = gout.detach().to(dtype=torch.float32).contiguous()
        # one gradient buffer per packed tensor (+ one for the unused screen-space gradient); every view's backward writes its own
        # rows, rows behind offsets[-1] (the unused tail of the packed capacity) are never read by the pack backward
        d_xyz, d_rgb, d_rot, d_scale, d_op = (torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes)
        d_m2 = torch.empty(ctx.shapes[0], dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        for i, h in enumerate(views):
            if side[i] is not cur:
                side[i].wait_stream(cur)
            ins = h.own if h.own is not None else (xyz, rgb, opacity.reshape(-1), scale, rot)
            with torch.cuda.stream(side[i]):  # (a workspace replaced by the overflow repair is picked up from h.ws_box in there)
                _RZ._backward_impl(h, tuple(ins) + tuple(h.tail), g[i], (d_xyz, d_rgb, d_op, d_scale, d_rot, d_m2), ctx.color_grad)
        for i in range(len(views)):
            if side[i] is not cur:
                cur.wait_stream(side[i])
        return d_xyz, (d_rgb if ctx.color_grad else None), d_rot, d_scale, d_op, None, None, None


def render(data, idx, pts_xyz, pts_rgb, rotations, scales, opacity, bg_color, grad_arena=None):
    """Render one novel view.  Same arguments and return value as the reference's render(): returns image [3,H,W].
    (grad_arena: internal, see pts2render.)"""
    nv = data['novel_view']
    bg = _bg_tensor(bg_color, pts_xyz.device)
    screenspace_points = torch.zeros_like(pts_xyz, dtype=torch.float32, requires_grad=True, device=pts_xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    raster_settings = GaussianRasterizationSettings(
        image_height=int(nv['height'][idx]), image_width=int(nv['width'][idx]),
        tanfovx=math.tan(_scalar(nv['FovX'][idx]) * 0.5), tanfovy=math.tan(_scalar(nv['FovY'][idx]) * 0.5),
        bg=bg, scale_modifier=1.0, viewmatrix=nv['world_view_transform'][idx], projmatrix=nv['full_proj_transform'][idx],
        sh_degree=3, campos=nv['camera_center'][idx], prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    rendered_image, _ = rasterizer(means3D=pts_xyz, means2D=screenspace_points, shs=None, colors_precomp=pts_rgb,
                                   opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None, grad_arena=grad_arena)
    return rendered_image


def _settings(nv, idx, bg, view=None, proj=None):
    return GaussianRasterizationSettings(
        image_height=int(nv['height'][idx]), image_width=int(nv['width'][idx]),
        tanfovx=math.tan(_scalar(nv['FovX'][idx]) * 0.5), tanfovy=math.tan(_scalar(nv['FovY'][idx]) * 0.5),
        bg=bg, scale_modifier=1.0, viewmatrix=(nv['world_view_transform'] if view is None else view)[idx],
        projmatrix=(nv['full_proj_transform'] if proj is None else proj)[idx],
        sh_degree=3, campos=nv['camera_center'][idx], prefiltered=False, debug=False)


def _to_device_once(t, dev):
    """[B,4,4] camera matrices of the whole batch in ONE copy (H1: in training they arrive as pinned CPU tensors; per-sample slices
    would each cost their own small H2D copy)."""
    if t.device != dev or t.dtype != torch.float32 or t.requires_grad or not t.is_contiguous():
        t = t.detach().to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
    return t


def pts2render(data, bg_color):
    """Same contract as the reference's pts2render(): writes data['novel_view']['img_pred'] = [B,3,H,W].

    The flatten / mask-gather / concat / rgb-affine of lib/GaussianRender.py:15-34 runs as one fused op for the whole batch
    (pack.py: 3 launches, no sync) instead of 10 boolean-index gathers + syncs per sample, and the B + 1 row offsets STAY ON THE
    DEVICE: the B renders run as ONE autograd node with the samples on B HIP streams, every view reading its row range of the packed
    arrays from device memory (_RenderBatch) -- with GPSGS_CHECK=none the whole pack -> render -> loss -> backward chain is launches only
    and can be captured into ONE HIP graph (tests/test_gpu_pack.py).  GPSGS_PTS2RENDER=loop restores the literal per-sample loop of
    render() calls, which reads the offsets back (also taken when the samples differ in image size)."""
    from .pack import pack_views

    bs = data['lmain']['img'].shape[0]
    xyz, rgb, rot, scale, opacity, offsets = pack_views(data)
    nv = data['novel_view']
    dev = xyz.device
    sizes_hw = {(int(nv['height'][i]), int(nv['width'][i])) for i in range(bs)}
    if os.environ.get("GPSGS_PTS2RENDER", "batch") != "loop" and len(sizes_hw) == 1:
        bg = _bg_tensor(bg_color, dev)
        view, proj = _to_device_once(nv['world_view_transform'], dev), _to_device_once(nv['full_proj_transform'], dev)
        settings = [_settings(nv, i, bg, view, proj) for i in range(bs)]
        nv['img_pred'] = _RenderBatch.apply(xyz, rgb, rot, scale, opacity, offsets, settings, xyz.shape[0] // bs)  # no read-back of the offsets
        return data
    return _pts2render_loop(data, bg_color, (xyz, rgb, rot, scale, opacity), offsets.tolist())


def _pts2render_loop(data, bg_color, packed, offs):
    """The per-sample form: one render() (one rasteriser autograd node) per sample, on the current stream."""
    bs = data['lmain']['img'].shape[0]
    xyz, rgb, rot, scale, opacity = packed
    # ONE split per packed tensor (its backward is one concatenation of the per-sample gradients); B Python slices would make
    # autograd zero-fill and add a full-size gradient per sample and tensor (measured: 0.6 ms of a 4.1 ms stage-2 iteration)
    sizes = [offs[i + 1] - offs[i] for i in range(bs)] + [xyz.shape[0] - offs[bs]]
    if any(t.requires_grad for t in packed):
        # gradient arenas: one buffer per packed tensor; every sample's rasteriser backward writes its rows in place and the
        # split's backward hands the whole buffer on (no per-sample allocation, no concatenation).  Rows behind offs[bs] are
        # never read by the pack backward.
        arenas = [torch.empty(t.shape, dtype=torch.float32, device=t.device) for t in packed]
        a_parts = [a.split(sizes) for a in arenas]
    else:
        arenas, a_parts = [None] * 5, None
    parts = [_SplitRows.apply(t, sizes, a) for t, a in zip(packed, arenas)]
    out = []
    dev = xyz.device
    cur = torch.cuda.current_stream(dev)
    # GPSGS_PTS2RENDER_STREAMS=1: each sample on its own HIP stream also in this form (measured no gain: through B autograd nodes the
    # batch loop is host-bound, tools/stage2_ab.py -- the reason _RenderBatch exists)
    concurrent = bs > 1 and not torch.cuda.is_current_stream_capturing() and os.environ.get("GPSGS_PTS2RENDER_STREAMS", "0") == "1"
    side = _streams(dev, bs) if concurrent else [cur] * bs
    with _RZ.defer_capacity_checks():
        for i in range(bs):
            # arena order expected by the rasteriser: means3D, colours, opacities, scales, rotations
            ga = (a_parts[0][i], a_parts[1][i], a_parts[4][i], a_parts[3][i], a_parts[2][i]) if a_parts is not None else None
            if side[i] is not cur:
                side[i].wait_stream(cur)
            with torch.cuda.stream(side[i]):
                out.append(render(data, i, parts[0][i], parts[1][i], parts[2][i], parts[3][i], parts[4][i], bg_color=bg_color,
                                  grad_arena=ga).unsqueeze(0))
    for i in range(bs):
        if side[i] is not cur:
            cur.wait_stream(side[i])
            out[i].record_stream(cur)
    data['novel_view']['img_pred'] = torch.cat(out, dim=0)
    return data


def pts2render_unfused(data, bg_color):
    """Literal mirror of the reference's pts2render (per-sample torch mask-gathers); kept as the comparison baseline."""
    bs = data['lmain']['img'].shape[0]
    out = []
    for i in range(bs):
        parts = [[], [], [], [], []]
        for view in ('lmain', 'rmain'):
            d = data[view]
            valid = d['pts_valid'][i, :]
            maps = (d['xyz'][i], d['img'][i].permute(1, 2, 0).reshape(-1, 3), d['rot_maps'][i].permute(1, 2, 0).reshape(-1, 4),
                    d['scale_maps'][i].permute(1, 2, 0).reshape(-1, 3), d['opacity_maps'][i].permute(1, 2, 0).reshape(-1, 1))
            for lst, m in zip(parts, maps):
                lst.append(m[valid])
        xyz, rgb, rot, scale, opacity = (torch.cat(p, dim=0) for p in parts)
        rgb = rgb * 0.5 + 0.5
        out.append(render(data, i, xyz, rgb, rot, scale, opacity, bg_color=bg_color).unsqueeze(0))
    data['novel_view']['img_pred'] = torch.cat(out, dim=0)
    return data
