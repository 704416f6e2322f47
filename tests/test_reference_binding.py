"""CPU (build container only): the reference's OWN glue files, unmodified, bind to the drop-in modules.

Skipped where /root/reference does not exist (the GPU box).  Each check runs in a fresh interpreter whose sys.path holds
gps-gaussian_amd/dropin followed by /root/reference -- exactly the integration INTEGRATION.md describes -- imports the reference's
real modules and drives them with CPU tensors (`device="cuda"` requests are redirected to the CPU: there is no GPU here):

  * /root/reference/lib/GaussianRender.py::pts2render -> /root/reference/gaussian_renderer/__init__.py::render ->
    `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer` must resolve to
    gps_gaussian_amd.rasterizer, the 12-field keyword construction and the 8-keyword call must be accepted, the call must arrive at
    `_RasterizeGaussians.forward` with exactly the tensors of tests/golden/pts2render_golden.npz (a fixture produced by the same
    reference function), and it must stop there, at the "must live on a GPU" check -- there is no CPU path to fall into;
  * /root/reference/core/corr.py::CorrBlockFast1D.__call__ -> CorrSampler.apply -> `corr_sampler.forward(volume, coords, radius)`
    must resolve to gps_gaussian_amd.corr.forward with the [N,H1,W1,W2] / [N,1,H1,W1] / int arguments and stop at the same kind of check.
"""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import GOLDEN, ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")

_PRELUDE = """
import os, sys
import numpy as np, torch
ROOT, REF, GOLDEN = %r, %r, %r
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "gps-gaussian_amd", "dropin"))   # the whole integration: this directory ahead on sys.path
# there is no GPU in this container: requests for device="cuda" land on the CPU (the reference hard-codes them)
def _cpu(kw):
    if str(kw.get("device", "")).startswith("cuda"):
        kw["device"] = "cpu"
    return kw
for _name in ("tensor", "zeros_like", "zeros", "ones"):
    _orig = getattr(torch, _name)
    setattr(torch, _name, (lambda f: (lambda *a, **k: f(*a, **_cpu(k))))(_orig))
torch.Tensor.cuda = lambda self, *a, **k: self
"""


def _run(body):
    code = (_PRELUDE % (ROOT, REF, GOLDEN)) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_reference_pts2render_and_render_bind_to_the_dropin_rasteriser():
    out = _run("""
        import lib.GaussianRender as GR                 # the reference's file, unmodified
        import gaussian_renderer as GRR                 # ditto; imports diff_gaussian_rasterization
        import diff_gaussian_rasterization as DGR
        import gps_gaussian_amd.rasterizer as RZ
        assert GR.__file__.startswith(REF) and GRR.__file__.startswith(REF)
        assert DGR.GaussianRasterizer is RZ.GaussianRasterizer and GRR.GaussianRasterizer is RZ.GaussianRasterizer
        assert DGR.GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                             "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
        gold = np.load(os.path.join(GOLDEN, "pts2render_golden.npz"))
        B, side = 2, 12
        data = {v: {k: torch.from_numpy(gold["%s_%s" % (v, k)]) for k in ("img", "xyz", "pts_valid", "rot_maps", "scale_maps", "opacity_maps")}
                for v in ("lmain", "rmain")}
        data["novel_view"] = dict(FovX=torch.tensor([0.9, 0.8]), FovY=torch.tensor([0.7, 0.6]), width=torch.tensor([24, 24]), height=torch.tensor([20, 20]),
                                  world_view_transform=torch.eye(4)[None].repeat(B, 1, 1), full_proj_transform=torch.eye(4)[None].repeat(B, 1, 1),
                                  camera_center=torch.zeros(B, 3))
        seen = []
        real_fwd = RZ._RasterizeGaussians.forward
        def spy(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena=None):
            seen.append(dict(means3D=means3D, means2D=means2D, sh=sh, colors=colors_precomp, opacities=opacities, scales=scales,
                             rotations=rotations, cov=cov3Ds_precomp, rs=raster_settings))
            return real_fwd(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_arena)
        RZ._RasterizeGaussians.forward = staticmethod(spy)
        try:
            GR.pts2render(data, [0.1, 0.2, 0.3])
            raise SystemExit("the drop-in rendered on the CPU: there must be no CPU path")
        except RuntimeError as e:
            assert "must live on a GPU" in str(e), e
        assert len(seen) == 1                              # sample 0 reached _RasterizeGaussians.forward, which refused the CPU tensors
        a = seen[0]
        for k, g in (("means3D", "xyz"), ("colors", "rgb"), ("rotations", "rot"), ("scales", "scale"), ("opacities", "opacity")):
            np.testing.assert_array_equal(a[k].detach().numpy(), gold["out0_" + g], err_msg=k)
        assert a["sh"] is None and a["cov"] is None and tuple(a["means2D"].shape) == tuple(gold["out0_xyz"].shape) and a["means2D"].requires_grad
        rs = a["rs"]
        assert (rs.image_height, rs.image_width, rs.sh_degree, rs.prefiltered, rs.debug, rs.scale_modifier) == (20, 24, 3, False, False, 1.0)
        assert abs(rs.tanfovx - np.tan(0.45)) < 1e-6 and abs(rs.tanfovy - np.tan(0.35)) < 1e-6
        np.testing.assert_allclose(rs.bg.numpy(), [0.1, 0.2, 0.3])
        assert tuple(rs.viewmatrix.shape) == (4, 4) and tuple(rs.projmatrix.shape) == (4, 4) and tuple(rs.campos.shape) == (3,)
        # the two argument checks upstream's Python wrapper makes, with upstream's messages
        r = DGR.GaussianRasterizer(raster_settings=rs)
        for kw in (dict(shs=None, colors_precomp=None), dict(shs=torch.zeros(1, 16, 3), colors_precomp=a["colors"])):
            try:
                r(means3D=a["means3D"], means2D=a["means2D"], opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"], cov3D_precomp=None, **kw)
                raise SystemExit("accepted an invalid SH / colour combination")
            except Exception as e:
                assert "excatly one of either SHs or precomputed colors" in str(e), e
        print("BOUND pts2render")
    """)
    assert "BOUND pts2render" in out


def test_reference_corr_block_binds_to_the_dropin_sampler():
    out = _run("""
        import core.corr as RC                          # the reference's file, unmodified; `import corr_sampler` inside a try/except
        import corr_sampler
        import gps_gaussian_amd.corr as MC
        assert RC.__file__.startswith(REF) and RC.corr_sampler is corr_sampler and corr_sampler.forward is MC.forward and corr_sampler.backward is MC.backward
        seen = []
        real = MC.forward
        def spy(volume, coords, radius):
            seen.append((tuple(volume.shape), volume.dtype, tuple(coords.shape), coords.dtype, radius))
            return real(volume, coords, radius)
        corr_sampler.forward = spy
        B, D, H, W = 2, 16, 5, 40
        blk = RC.CorrBlockFast1D(torch.randn(B, D, H, W), torch.randn(B, D, H, W), num_levels=4, radius=4)
        assert [tuple(v.shape) for v in blk.corr_pyramid] == [(B, H, W, 1, W >> i) for i in range(4)]
        try:
            blk(torch.rand(B, 2, H, W) * W)
            raise SystemExit("the drop-in sampled on the CPU: there must be no CPU path")
        except RuntimeError as e:
            assert "must live on a GPU" in str(e), e
        assert seen == [((B, H, W, W), torch.float32, (B, 1, H, W), torch.float32, 4)], seen
        print("BOUND corr_sampler")
    """)
    assert "BOUND corr_sampler" in out
