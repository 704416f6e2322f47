"""Generates the golden fixtures in this directory by IMPORTING the reference (read-only at /root/reference).

Run here (the build container) only:   python tests/golden/make_golden.py
/root/reference does not exist on the GPU box; tests read the committed .npz files, never the reference.

Fixtures (all produced by the reference's own code, on CPU):
  camera_golden.npz        lib/utils.py::get_novel_calib (+ lib/graphics_utils.py) on the synthetic rig, 3 ratios x {hr, not hr}
  depth2pc_golden.npz      lib/utils.py::depth2pc on a small random inverse-depth map
  corr_sampler_golden.npz  core/corr.py::CorrBlock1D lookups (4 levels, radius 4) + autograd gradient w.r.t. each volume
  pts2render_golden.npz    lib/GaussianRender.py::pts2render's compaction: the exact arguments it hands to render()
  loss_golden.npz          lib/loss.py::l1_loss / ssim and their autograd gradients
  unproject_golden.npz     lib/utils.py::flow2depth + depth2pc and the gradient w.r.t. the flow
  corr_pyramid_golden.npz  core/corr.py::CorrBlockFast1D.__init__ (volume + average-pool pyramid) and its gradient w.r.t. both feature maps
  upsample_golden.npz      core/raft_stereo_human.py::FlowUpdateModule.upsample_flow and its gradients (flow, mask)
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

torch.manual_seed(1314)
np.random.seed(1314)
torch.Tensor.cuda = lambda self, *a, **k: self  # the reference hard-codes .cuda(); there is no GPU here

import gps_gaussian_amd  # noqa: E402
from gps_gaussian_amd import synthetic as S  # noqa: E402

# ---- camera -------------------------------------------------------------------------------------------------------
from lib.utils import get_novel_calib, depth2pc  # noqa: E402

res = 1024
(i0, e0), (i1, e1) = S.source_camera(res, 0.0), S.source_camera(res, 22.5)
cam = {}
for hr in (False, True):
    for ratio in (0.1, 0.5, 0.9):
        side = res * 2 if hr else res
        data = {"lmain": {"intr": torch.from_numpy(i0)[None], "extr": torch.from_numpy(e0)[None]},
                "rmain": {"intr": torch.from_numpy(i1)[None], "extr": torch.from_numpy(e1)[None]},
                "novel_view": {"width": torch.tensor([side]), "height": torch.tensor([side])}}
        opt = SimpleNamespace(use_hr_img=hr, znear=0.01, zfar=100.0, trans=[0.0, 0.0, 0.0], scale=1.0)
        out = get_novel_calib(data, opt, ratio=ratio)["novel_view"]
        tag = "hr%d_r%02d_" % (int(hr), int(ratio * 10))
        for k in ("FovX", "FovY", "world_view_transform", "full_proj_transform", "camera_center"):
            cam[tag + k] = out[k][0].numpy()
cam.update(intr0=i0, extr0=e0, intr1=i1, extr1=e1)
np.savez_compressed(os.path.join(HERE, "camera_golden.npz"), **cam)

# ---- depth2pc -----------------------------------------------------------------------------------------------------
rng = np.random.default_rng(7)
S_ = 24
inv = rng.uniform(0.3, 0.7, (1, 1, S_, S_)).astype(np.float32)
inv[0, 0, :3] = 0.0
k_small = i0.copy(); k_small[:2] *= S_ / res
pts = depth2pc(torch.from_numpy(inv), torch.from_numpy(e0)[None], torch.from_numpy(k_small)[None])
np.savez_compressed(os.path.join(HERE, "depth2pc_golden.npz"), inv_depth=inv[0, 0], intr=k_small, extr=e0, xyz=pts[0].numpy())

# ---- corr sampler -------------------------------------------------------------------------------------------------
from core.corr import CorrBlock1D  # noqa: E402

B, D, H, W1 = 2, 16, 5, 40
f1 = torch.randn(B, D, H, W1)
f2 = torch.randn(B, D, H, W1)
blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
vols = [v.detach().clone().requires_grad_(True) for v in blk.corr_pyramid[:4]]
blk.corr_pyramid = vols + blk.corr_pyramid[4:]
coords = torch.rand(B, 2, H, W1) * (W1 + 12) - 6.0       # includes taps left and right of the volume
coords[0, 0, 0, :4] = torch.tensor([0.0, 3.0, float(W1 - 1), -4.0])  # exact integers / borders
out = blk(coords)                                          # [B, 4*9, H, W1]
gout = torch.randn_like(out)
out.backward(gout)
cs = {"coords": coords[:, :1].numpy(), "out": out.detach().numpy(), "grad_out": gout.numpy(), "radius": np.int32(4)}
for lvl, v in enumerate(vols):
    cs["volume%d" % lvl] = v.detach().reshape(B, H, W1, -1).numpy()       # the layout corr_sampler.forward receives
    cs["grad_volume%d" % lvl] = v.grad.reshape(B, H, W1, -1).numpy()
np.savez_compressed(os.path.join(HERE, "corr_sampler_golden.npz"), **cs)

# ---- pts2render compaction ----------------------------------------------------------------------------------------
stub = types.ModuleType("diff_gaussian_rasterization")
stub.GaussianRasterizationSettings = object
stub.GaussianRasterizer = object
sys.modules["diff_gaussian_rasterization"] = stub
import lib.GaussianRender as GR  # noqa: E402

captured = []


def fake_render(data, idx, xyz, rgb, rot, scale, opacity, bg_color):
    captured.append(dict(xyz=xyz.numpy().copy(), rgb=rgb.numpy().copy(), rot=rot.numpy().copy(), scale=scale.numpy().copy(),
                         opacity=opacity.numpy().copy()))
    return torch.zeros(3, 4, 4)


GR.render = fake_render
Bp, Sp = 2, 12
data = {}
g = torch.Generator().manual_seed(5)
for view in ("lmain", "rmain"):
    data[view] = dict(img=torch.rand(Bp, 3, Sp, Sp, generator=g) * 2 - 1, xyz=torch.randn(Bp, Sp * Sp, 3, generator=g),
                      pts_valid=torch.rand(Bp, Sp * Sp, generator=g) > 0.4, rot_maps=torch.randn(Bp, 4, Sp, Sp, generator=g),
                      scale_maps=torch.rand(Bp, 3, Sp, Sp, generator=g) * 0.01, opacity_maps=torch.rand(Bp, 1, Sp, Sp, generator=g))
data["novel_view"] = {}
GR.pts2render(data, [0, 0, 0])
pk = {}
for view in ("lmain", "rmain"):
    for k, v in data[view].items():
        pk["%s_%s" % (view, k)] = v.numpy()
for i, c in enumerate(captured):
    for k, v in c.items():
        pk["out%d_%s" % (i, k)] = v
np.savez_compressed(os.path.join(HERE, "pts2render_golden.npz"), **pk)
print("golden fixtures written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))

# ---- L1 + SSIM loss (row f2) ---------------------------------------------------------------------------------------
from lib.loss import l1_loss, ssim  # noqa: E402

torch.manual_seed(77)
pred = torch.rand(2, 3, 37, 45, requires_grad=True)
gt = (pred.detach() + 0.15 * torch.randn(2, 3, 37, 45)).clamp(0, 1)
l1 = l1_loss(pred, gt)
ss = ssim(pred, gt)
g_l1, = torch.autograd.grad(l1, pred, retain_graph=True)
g_ss, = torch.autograd.grad(ss, pred)
np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), pred=pred.detach().numpy(), gt=gt.numpy(), l1=l1.item(), ssim=ss.item(),
                    grad_l1=g_l1.numpy(), grad_ssim=g_ss.numpy())
print("loss golden: l1 %.6f ssim %.6f" % (l1.item(), ss.item()))

# ---- flow2depth + depth2pc (row f3) ---------------------------------------------------------------------------------
from lib.utils import flow2depth  # noqa: E402

torch.manual_seed(11)
Bu, Su = 2, 20
ku = torch.from_numpy(i0.copy()); ku[:2] *= Su / res
intr_u = torch.stack([ku, ku * torch.tensor([[1.0, 1, 1.01], [1, 1, 1], [1, 1, 1]])])
ref_intr_u = intr_u.clone(); ref_intr_u[:, 0, 2] += torch.tensor([0.7, -0.4])
extr_u = torch.stack([torch.from_numpy(e0), torch.from_numpy(e1)])
flow_u = (torch.rand(Bu, 1, Su, Su) * 3 + 1.5).requires_grad_(True)
mask_u = (torch.rand(Bu, 3, Su, Su) > 0.3).float()
Tf_u = torch.tensor([-6.5, -7.25])
du = {"ref_intr": ref_intr_u, "intr": intr_u, "flow_pred": flow_u, "Tf_x": Tf_u, "mask": mask_u}
depth_u = flow2depth(du)
xyz_u = depth2pc(depth_u, extr_u, intr_u)
gd, gx = torch.randn_like(depth_u), torch.randn_like(xyz_u)
gflow, = torch.autograd.grad((depth_u * gd).sum() + (xyz_u * gx).sum(), flow_u)
np.savez_compressed(os.path.join(HERE, "unproject_golden.npz"), ref_intr=ref_intr_u.numpy(), intr=intr_u.numpy(), extr=extr_u.numpy(),
                    flow=flow_u.detach().numpy(), mask=mask_u.numpy(), Tf_x=Tf_u.numpy(), depth=depth_u.detach().numpy(),
                    xyz=xyz_u.detach().numpy(), valid=(depth_u != 0).view(Bu, -1).numpy(), g_depth=gd.numpy(), g_xyz=gx.numpy(),
                    g_flow=gflow.numpy())
print("unproject golden written")

# ---- correlation volume + pyramid, convex upsampling (row f4) -------------------------------------------------------
from core.corr import CorrBlockFast1D  # noqa: E402  (its constructor is pure PyTorch; only __call__ needs the CUDA sampler)
from core.raft_stereo_human import FlowUpdateModule  # noqa: E402

torch.manual_seed(23)
Nc, Dc, Hc, W1c, W2c = 2, 24, 3, 45, 52          # ragged: W1 != W2, odd widths further down the pyramid (52, 26, 13, 6)
f1c = torch.randn(Nc, Dc, Hc, W1c, requires_grad=True)
f2c = torch.randn(Nc, Dc, Hc, W2c, requires_grad=True)
blkf = CorrBlockFast1D(f1c, f2c, num_levels=4, radius=4)
pyr = [p.squeeze(3) for p in blkf.corr_pyramid]
gp = [torch.randn_like(p) for p in pyr]
gf1, gf2 = torch.autograd.grad(sum((p * g).sum() for p, g in zip(pyr, gp)), (f1c, f2c))
cp = {"fmap1": f1c.detach().numpy(), "fmap2": f2c.detach().numpy(), "grad_fmap1": gf1.numpy(), "grad_fmap2": gf2.numpy()}
for l in range(4):
    cp["pyr%d" % l] = pyr[l].detach().numpy()
    cp["grad_pyr%d" % l] = gp[l].numpy()
np.savez_compressed(os.path.join(HERE, "corr_pyramid_golden.npz"), **cp)

torch.manual_seed(29)
Nu, Hu, Wu, fu = 2, 5, 7, 4
flow_c = torch.randn(Nu, 2, Hu, Wu, requires_grad=True)
mask_c = (2.0 * torch.randn(Nu, 9 * fu * fu, Hu, Wu)).requires_grad_(True)
fake_self = SimpleNamespace(args=SimpleNamespace(n_downsample=2))
up = FlowUpdateModule.upsample_flow(fake_self, flow_c, mask_c)
gup = torch.randn_like(up)
gfl, gmk = torch.autograd.grad((up * gup).sum(), (flow_c, mask_c))
np.savez_compressed(os.path.join(HERE, "upsample_golden.npz"), flow=flow_c.detach().numpy(), mask=mask_c.detach().numpy(), factor=np.int32(fu),
                    out=up.detach().numpy(), grad_out=gup.numpy(), grad_flow=gfl.numpy(), grad_mask=gmk.numpy())
print("corr pyramid + upsample golden written")
