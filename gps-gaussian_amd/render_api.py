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

_stream_pool = {}  # device index -> side streams: the samples of a batch are independent views and render concurrently


def _streams(dev, n):
    pool = _stream_pool.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


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


def render(data, idx, pts_xyz, pts_rgb, rotations, scales, opacity, bg_color, grad_arena=None):
    """Render one novel view.  Same arguments and return value as the reference's render(): returns image [3,H,W].
    (grad_arena: internal, see pts2render.)"""
    nv = data['novel_view']
    bg = torch.tensor(bg_color, dtype=torch.float32, device=pts_xyz.device)
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


def pts2render(data, bg_color):
    """Same contract as the reference's pts2render(): writes data['novel_view']['img_pred'] = [B,3,H,W].

    The flatten / mask-gather / concat / rgb-affine of lib/GaussianRender.py:15-34 runs as one fused op for the whole batch
    (pack.py: 3 launches, no sync) instead of 10 boolean-index gathers + syncs per sample; the only host read is the B+1
    row offsets that give every sample's tensors their exact shape."""
    from .pack import pack_views

    bs = data['lmain']['img'].shape[0]
    xyz, rgb, rot, scale, opacity, offsets = pack_views(data)
    offs = offsets.tolist()
    # ONE split per packed tensor (its backward is one concatenation of the per-sample gradients); B Python slices would make
    # autograd zero-fill and add a full-size gradient per sample and tensor (measured: 0.6 ms of a 4.1 ms stage-2 iteration)
    sizes = [offs[i + 1] - offs[i] for i in range(bs)] + [xyz.shape[0] - offs[bs]]
    packed = (xyz, rgb, rot, scale, opacity)
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
    # One view's kernels leave the chip under-occupied (~5 one-wave compositing work items per SIMD at 1024^2 / 600k Gaussians), and the
    # samples of a batch are independent: with GPSGS_PTS2RENDER_STREAMS=1 each is enqueued on its own HIP stream so that they render
    # concurrently (the reference loops over them on one stream, lib/GaussianRender.py:9); their capacity checks are collected and run
    # after all of them are in flight, and autograd runs every sample's backward on the stream of its forward.  Off by default: through
    # PyTorch's autograd the batch loop is host-bound (tools/stage2_ab.py: 2.73 ms vs 2.79 ms per iteration of 4 pairs), so the overlap
    # that the C-ABI sessions get (bench.py --inflight: +27 %) does not materialise here.
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
