"""GPU (-m gpu): the reference's OWN code, unmodified, executing on the HIP kernels of an MI355X.

The reference comes from tools/refenv.reference_dir(): a CHECKOUT (--reference, $GPSGS_REFERENCE, /root/reference).  Nothing of the reference travels
to the GPU box (through round 5 a sourceless-bytecode build did: the runs of rounds 3-5 are recorded in profiles/r0[345]_full_pipeline.md,
profiles/full_pipeline.json and the parity reports; since round 6 it no longer does), so these tests SKIP there and run where a checkout and an MI355X
coexist.  Every check runs in a fresh interpreter whose sys.path holds gps-gaussian_amd/dropin ahead of the reference -- the whole integration of
INTEGRATION.md -- and calls the reference's functions on cuda:0:

  * lib/GaussianRender.py:6-40 `pts2render` -> gaussian_renderer/__init__.py:17-67 `render` -> the drop-in `diff_gaussian_rasterization`:
    image and all map gradients BIT-equal to gps_gaussian_amd.render_api.pts2render_unfused (the declared mirror) and to the fused
    pts2render, and within the north star's tolerances (RGB 1e-4, gradients 1e-3) of the CPU oracle; camera tensors pinned on the host
    (training) and on the device (test scripts);
  * core/corr.py:31-61 `CorrBlockFast1D` (-> `corr_sampler.forward/backward`, the drop-in) against core/corr.py:110-156 `CorrBlock1D`
    (the reference's pure-PyTorch implementation) evaluated beside it on the GPU, values and gradients;
  * lib/network.py `RtStereoHumanModel` with `corr_implementation: reg_cuda` (the default, the drop-in sampler) against `reg`;
  * test_view_interp.py and train_stage2.py as `__main__` (tools/run_reference.py), and tools/launch_stage2.py at world size 1 with the
    REAL networks, data set class and losses (BASELINE configs 3 and 4 at a reduced size; bench sizes: profiles/r03_full_pipeline.md).
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import refenv  # noqa: E402

REF = refenv.reference_dir()
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(REF is None, reason="no reference checkout here (nothing of the reference travels to the GPU box)")]

_PRELUDE = """
import json, math, os, sys
import numpy as np, torch
ROOT, REF = %r, %r
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refenv
refenv.activate(REF)                      # drop-in ahead of the reference on sys.path (+ yacs / cv2 stand-ins if missing)
sys.path.insert(0, os.path.join(ROOT, "tests"))
dev = torch.device("cuda:0")
"""


def _run(body, timeout=900, env=None):
    code = (_PRELUDE % (ROOT, REF)) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd="/tmp",
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    return r.stdout


@pytest.mark.parametrize("cams,bg", [("pinned", (0.0, 0.0, 0.0)), ("cuda", (0.1, 0.3, 0.2))], ids=["training-pinned-cameras-black", "test-cuda-cameras-colour-bg"])
def test_reference_pts2render_runs_on_the_hip_rasteriser(cams, bg):
    out = _run("""
        CAMS, BG = %r, %r
        import lib.GaussianRender as GR                 # the reference's files
        import gaussian_renderer as GRR
        import diff_gaussian_rasterization as DGR
        import gps_gaussian_amd.rasterizer as RZ
        from gps_gaussian_amd import render_api, synthetic as S
        from conftest import touched_by_fragile
        from oracle.gsr_oracle import OracleRasterizer
        assert os.path.dirname(GR.__file__).startswith(REF) and os.path.dirname(GRR.__file__).startswith(REF), (GR.__file__, GRR.__file__)
        assert GRR.GaussianRasterizer is RZ.GaussianRasterizer and DGR.GaussianRasterizationSettings is RZ.GaussianRasterizationSettings
        res, B = 192, 2
        samples = [S.make_stereo_sample(res, 16000 + 6000 * i, seed=77 + i, render_res=2 * res, ratio=0.3 + 0.4 * i) for i in range(B)]   # use_hr_img: render = 2 x source
        keys = ("xyz", "img", "rot_maps", "scale_maps", "opacity_maps")

        def make_data():
            data = {}
            for v in ("lmain", "rmain"):
                st = lambda k: torch.from_numpy(np.stack([s[v][k] for s in samples])).to(dev)
                data[v] = {k: st(k).requires_grad_(True) for k in keys}
                data[v]["pts_valid"] = st("pts_valid")
            nv = [s["novel_view"] for s in samples]
            cam = dict(FovX=torch.tensor([float(c["FovX"]) for c in nv]), FovY=torch.tensor([float(c["FovY"]) for c in nv]),
                       width=torch.tensor([c["width"] for c in nv]), height=torch.tensor([c["height"] for c in nv]),
                       world_view_transform=torch.from_numpy(np.stack([c["world_view_transform"] for c in nv])),
                       full_proj_transform=torch.from_numpy(np.stack([c["full_proj_transform"] for c in nv])),
                       camera_center=torch.from_numpy(np.stack([c["camera_center"] for c in nv])))
            if CAMS == "pinned":       # training: the DataLoader hands pinned host tensors, only lmain / rmain go to the GPU (train_stage2.py:155-157)
                cam = {k: (v.pin_memory() if v.dtype == torch.float32 and v.dim() > 1 else v) for k, v in cam.items()}
            else:                      # test scripts: get_novel_calib puts them on the GPU (lib/utils.py:50-54)
                cam = {k: (v.cuda() if v.dtype == torch.float32 else v) for k, v in cam.items()}
            data["novel_view"] = cam
            return data

        gout = torch.randn(B, 3, 2 * res, 2 * res, generator=torch.Generator().manual_seed(5)).to(dev)

        def run(fn):
            data = make_data()
            img = fn(data, list(BG))["novel_view"]["img_pred"]
            img.backward(gout)
            torch.cuda.synchronize()
            return img.detach(), {v: {k: data[v][k].grad.clone() for k in keys} for v in ("lmain", "rmain")}

        calls = []
        real = RZ.GaussianRasterizer.forward             # (the module's entry: behind it the compiled host path or the Python one, whichever applies)
        RZ.GaussianRasterizer.forward = lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1]
        img_r, g_r = run(GR.pts2render)                  # the REFERENCE's pts2render -> the reference's render -> HIP kernels
        RZ.GaussianRasterizer.forward = real
        assert len(calls) == B
        img_m, g_m = run(render_api.pts2render_unfused)
        img_f, g_f = run(render_api.pts2render)
        assert torch.equal(img_r, img_m) and torch.equal(img_r, img_f)
        for v in g_r:
            for k in keys:
                assert torch.equal(g_r[v][k], g_m[v][k]), ("mirror", v, k)
                assert torch.equal(g_r[v][k], g_f[v][k]), ("fused", v, k)
        # against the CPU oracle, per sample
        rep = []
        for i, s in enumerate(samples):
            g = S.compact_sample(s)
            c = s["novel_view"]
            o = OracleRasterizer("f32")
            oimg, _ = o.forward(g["means3D"], g["colors"], g["opacities"], g["scales"], g["rotations"], c["world_view_transform"], c["full_proj_transform"],
                                c["width"], c["height"], math.tan(float(c["FovX"]) * 0.5), math.tan(float(c["FovY"]) * 0.5), np.asarray(BG, np.float32))
            og = o.backward(gout[i].cpu().numpy())
            solid, touched = touched_by_fragile(o)
            err = np.abs(img_r[i].cpu().numpy() - oimg).max(0)
            assert (~solid).mean() < 5e-3 and err[solid].max() <= 1e-4, (float((~solid).mean()), float(err[solid].max()))
            assert err.max() <= 2.0 / 255 + 1e-4
            got = {}
            for name, k, C in (("means3D", "xyz", 3), ("rotations", "rot_maps", 4), ("scales", "scale_maps", 3), ("opacities", "opacity_maps", 1), ("colors", "img", 3)):
                parts = []
                for v in ("lmain", "rmain"):
                    gm = g_r[v][k][i]
                    gm = gm if k == "xyz" else gm.reshape(C, -1).T
                    parts.append(gm[torch.from_numpy(s[v]["pts_valid"]).to(dev)])
                got[name] = torch.cat(parts, 0).cpu().numpy() * (2.0 if name == "colors" else 1.0)     # rgb = img * 0.5 + 0.5 (lib/GaussianRender.py:31)
            worst = 0.0
            strict = ~touched & (o.geom()["radii"] > 0)
            for name in got:
                sc = np.abs(og[name]).max() + 1e-30
                e = np.abs(got[name] - og[name]) / (np.abs(og[name]) + 1e-3 * sc)
                assert (e > 1e-3).any(-1).mean() < 2e-3, (name, float((e > 1e-3).any(-1).mean()))
                assert strict.sum() >= 0.2 * (o.geom()["radii"] > 0).sum()
                over = int((e[strict] > 1e-3).any(-1).sum())
                assert over <= max(1, int(2e-5 * strict.sum())), (name, over)
                worst = max(worst, float(e[strict].max()))
            rep.append(dict(P=int(g["means3D"].shape[0]), rgb_max_err_solid=float(err[solid].max()), fragile_pixels=int((~solid).sum()),
                            strict_fraction=float(strict.sum() / max(1, (o.geom()["radii"] > 0).sum())), grad_max_err_strict=worst))
        print("REFERENCE_PTS2RENDER " + json.dumps(rep))
    """ % (cams, bg))
    line = [l for l in out.splitlines() if l.startswith("REFERENCE_PTS2RENDER ")]
    assert line, out[-2000:]
    rep = json.loads(line[0].split(" ", 1)[1])
    print(rep)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"test": "reference_pts2render[%s]" % cams, "samples": rep}) + "\n")


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_reference_corr_block_fast_runs_on_the_hip_sampler(dtype):
    out = _run("""
        DT = getattr(torch, %r)
        import core.corr as RC                           # the reference's file
        import corr_sampler
        import gps_gaussian_amd.corr as MC
        assert os.path.dirname(RC.__file__).startswith(REF) and RC.corr_sampler is corr_sampler and corr_sampler.forward is MC.forward
        B, D, H, W = 4, 96, 32, 64
        g = torch.Generator().manual_seed(3)
        f1 = torch.randn(B, D, H, W, generator=g).to(dev)
        f2 = torch.randn(B, D, H, W, generator=g).to(dev)
        coords = torch.stack([torch.rand(B, H, W, generator=g) * (W + 12) - 6, torch.zeros(B, H, W)], 1).to(dev)     # some taps outside the row: zero padding
        w = torch.randn(B, 36, H, W, generator=g).to(dev)

        def run(block, dt):
            a, b = f1.to(dt).requires_grad_(True), f2.to(dt).requires_grad_(True)
            out = block(a, b, num_levels=4, radius=4)(coords)
            (out.float() * w).sum().backward()
            return out.float().detach(), a.grad.float(), b.grad.float()

        calls = []
        real_f, real_b = MC.forward, MC.backward
        corr_sampler.forward = lambda *a: (calls.append("f"), real_f(*a))[1]
        corr_sampler.backward = lambda *a: (calls.append("b"), real_b(*a))[1]
        fast = run(RC.CorrBlockFast1D, DT)                # -> CorrSampler.apply -> corr_sampler.forward / backward = HIP kernels
        assert calls.count("f") == 4 and calls.count("b") == 4, calls
        slow = run(RC.CorrBlock1D, torch.float32)         # the reference's own PyTorch implementation (grid_sample), beside it on the GPU
        tol = 2e-5 if DT is torch.float32 else 2e-2
        errs = []
        for x, y in zip(fast, slow):
            s = float(y.abs().max())
            errs.append(float((x - y).abs().max()) / s)
            assert errs[-1] < tol, errs
        print("REFERENCE_CORR", errs)
    """ % dtype)
    assert "REFERENCE_CORR" in out


def test_reference_network_reg_cuda_equals_reg():
    """lib/network.py::RtStereoHumanModel (the reference's real networks, random weights) in eval mode: the default correlation
    implementation `reg_cuda` (core/raft_stereo_human.py:87-88 -> the drop-in sampler) against `reg` (pure PyTorch), then through the
    reference's flow2gsparms and pts2render: the rendered views agree."""
    out = _run("""
        from config.stereo_human_config import ConfigStereoHuman
        from lib.network import RtStereoHumanModel
        from lib.GaussianRender import pts2render
        from gps_gaussian_amd import synthetic as S
        work = refenv.make_workdir(REF, "/tmp/gpsgs_ref_net", {"raft": {"mixed_precision": False}})
        os.chdir(work)
        res = 256
        s = S.make_stereo_sample(res, 40000, seed=11, render_res=2 * res)
        imgs = {}
        for impl in ("reg_cuda", "reg"):
            c = ConfigStereoHuman(); c.load("config/stage2.yaml"); cfg = c.get_cfg()
            cfg.defrost(); cfg.raft.corr_implementation = impl; cfg.dataset.src_res = res; cfg.freeze()
            torch.manual_seed(1314)
            model = RtStereoHumanModel(cfg, with_gs_render=True).cuda().eval()
            data = {}
            for v, o in (("lmain", "rmain"), ("rmain", "lmain")):
                d = s[v]
                data[v] = dict(img=torch.from_numpy(d["img"] * d["pts_valid"].reshape(1, res, res))[None].cuda(), mask=torch.from_numpy(d["pts_valid"].reshape(1, 1, res, res).astype(np.float32)).cuda(),
                               intr=torch.from_numpy(d["intr"])[None].cuda(), ref_intr=torch.from_numpy(s[o]["intr"])[None].cuda(),
                               extr=torch.from_numpy(d["extr"])[None].cuda(), Tf_x=torch.tensor([-300.0 if v == "lmain" else 300.0]).cuda())
            nv = s["novel_view"]
            data["novel_view"] = dict(FovX=torch.tensor([float(nv["FovX"])]).cuda(), FovY=torch.tensor([float(nv["FovY"])]).cuda(), width=torch.tensor([2 * res]), height=torch.tensor([2 * res]),
                                      world_view_transform=torch.from_numpy(nv["world_view_transform"])[None].cuda(), full_proj_transform=torch.from_numpy(nv["full_proj_transform"])[None].cuda(),
                                      camera_center=torch.from_numpy(nv["camera_center"])[None].cuda())
            with torch.no_grad():
                data, _, _ = model(data, is_train=False)
                data = pts2render(data, bg_color=[0, 0, 0])
            imgs[impl] = (data["lmain"]["flow_pred"].float().clone(), data["novel_view"]["img_pred"].clone(), int(data["lmain"]["pts_valid"].sum() + data["rmain"]["pts_valid"].sum()))
        (fa, ia, na), (fb, ib, nb) = imgs["reg_cuda"], imgs["reg"]
        fe = float((fa - fb).abs().max()) / (float(fb.abs().max()) + 1e-6)
        assert fe < 1e-3, fe
        assert abs(na - nb) <= 0.001 * nb and nb > 1000, (na, nb)
        assert torch.isfinite(ia).all() and float((ia - ib).abs().mean()) < 1e-3, float((ia - ib).abs().mean())
        print("REFERENCE_NETWORK flow_rel_err=%.2e valid=%d img_mean_abs_diff=%.2e" % (fe, nb, float((ia - ib).abs().mean())))
    """)
    assert "REFERENCE_NETWORK" in out


def _tool(args, timeout=1500, env=None):
    r = subprocess.run([sys.executable] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_view_interp_script_with_the_import_hook_writes_the_same_views(tmp_path):
    """test_view_interp.py as __main__ twice -- as the reference runs it and with GPSGS_ACCELERATE=all (inference: pack, corr, upsample, unproject; no loss) --
    on the same data set and checkpoint: the JPEGs written must agree (the views are the same up to fp16 rounding inside the networks)."""
    import numpy as np
    from PIL import Image
    common = ["interp", "--res", "256", "--samples", "2", "--views", "3", "--write-images"]
    a = _tool([os.path.join(ROOT, "tools", "run_reference.py")] + common + ["--work", str(tmp_path / "w")])
    out_plain = {f: np.asarray(Image.open(tmp_path / "w" / "interp_out" / f)).astype(np.int32) for f in sorted(os.listdir(tmp_path / "w" / "interp_out"))}
    b = _tool([os.path.join(ROOT, "tools", "run_reference.py")] + common + ["--work", str(tmp_path / "w"), "--accelerate", "all"])
    out_hook = {f: np.asarray(Image.open(tmp_path / "w" / "interp_out" / f)).astype(np.int32) for f in sorted(os.listdir(tmp_path / "w" / "interp_out"))}
    assert a["accelerate"]["calls"] == {} and b["accelerate"]["calls"]["pack"] == 18 and b["accelerate"]["calls"]["loss"] == 0, (a["accelerate"], b["accelerate"])
    assert b["accelerate"]["calls"]["corr"] == 18 and b["accelerate"]["calls"]["unproject"] == 36 and b["accelerate"]["calls"]["upsample"] == 18
    assert b["accelerate"]["calls"]["loss_passthrough"] == 0 and b["accelerate"]["calls"]["unproject_passthrough"] == 0   # nothing fell through to the eager functions
    assert sorted(out_plain) == sorted(out_hook) and len(out_plain) == 6
    stats = {}
    for f in out_plain:
        d = np.abs(out_plain[f] - out_hook[f])
        stats[f] = (float((d > 16).mean()), float(np.median(d)), int(d.max()))
    print(stats)
    for f, (far, med, _) in stats.items():
        # 8-bit JPEGs of random-weight renders: pixels differ by a few levels where fp16 rounding inside the networks moved a depth or a scale
        assert far < 1e-2 and med <= 1, (f, stats[f])


def test_view_interp_script_runs_unmodified(tmp_path):
    """test_view_interp.py as __main__ (BASELINE config 3's loop at a reduced source size): the reference's loader reads the synthetic set
    from disk, its networks (random weights, loaded from the checkpoint the script asks for) regress the Gaussians, its pts2render draws
    them with the HIP rasteriser, five novel views per sample are written."""
    res = _tool([os.path.join(ROOT, "tools", "run_reference.py"), "interp", "--res", "256", "--samples", "2", "--views", "5", "--write-images", "--work", str(tmp_path / "w")])
    assert res["script_run"]["views"] == 10 and res["script_run"]["views_per_s_end_to_end"] > 0
    assert res["gaussians_per_view"]["min"] > 1000
    assert res["first_image"]["shape"] == [512, 512, 3] and res["first_image"]["nonblack_fraction"] > 0.02, res["first_image"]
    assert res["gpu_ms_per_view"]["render"] > 0 and res["gpu_ms_per_view"]["network"] > 0
    print(res)


def test_real_data_script_runs_unmodified(tmp_path):
    """test_real_data.py as __main__: the reference's third entry point -- one novel view per frame of a captured sequence, the novel camera at
    --ratio between the two source cameras (lib/utils.get_novel_calib), the reference's pts2render on the HIP rasteriser, one JPEG per frame."""
    res = _tool([os.path.join(ROOT, "tools", "run_reference.py"), "real", "--res", "256", "--samples", "3", "--ratio", "0.35", "--write-images", "--work", str(tmp_path / "w")])
    assert res["frames"] == 3 and res["frames_per_s_end_to_end"] > 0
    for im in res["images"]:
        assert im["shape"] == [512, 512, 3] and im["nonblack_fraction"] > 0.02, im
    assert sorted(os.listdir(tmp_path / "w" / "test_out")) == sorted(im["file"] for im in res["images"])
    print(res)


def test_train_stage2_script_runs_unmodified(tmp_path):
    """train_stage2.py as __main__ (one process; BASELINE config 4's iteration at a reduced size): real data set class + DataLoader workers, real
    networks under AMP, the reference's pts2render -> HIP rasteriser, L1 + SSIM, GradScaler backward through the HIP backward, AdamW."""
    res = _tool([os.path.join(ROOT, "tools", "run_reference.py"), "train", "--res", "256", "--steps", "6", "--batch", "2", "--train-samples", "2", "--work", str(tmp_path / "w")])
    # (the GradScaler skips the optimizer while it calibrates its loss scale from 65536 down: fewer AdamW steps than iterations is AMP at work)
    assert 1 <= res["optimizer_steps"] <= 6 and res["final_checkpoint_written"] and res["finite_weights"] and res["total_steps"] == 6, res
    print(res)


def test_launcher_world_1_with_the_real_networks(tmp_path):
    """tools/launch_stage2.py (the DDP entry) at world size 1 around the reference's real Trainer, networks, data set and losses; the
    timing file has one hipEvent span per iteration for the reference's own calls."""
    import make_synthetic_dataset as M
    data_root = str(tmp_path / "data")
    M.make_dataset(data_root, res=256, n_train=2, n_val=2, quiet=True)   # two validation samples: len_val = 1 batch (train_stage2.py:38)
    work = refenv.make_workdir(REF, str(tmp_path / "w"), {"stage1_ckpt": "None", "dataset": {"src_res": 256, "data_root": data_root}})
    tfile = str(tmp_path / "timing.json")
    res = _tool([os.path.join(ROOT, "tools", "launch_stage2.py"), "--reference", REF, "--workdir", work, "--steps", "5", "--exp-root", str(tmp_path / "exp"),
                 "--timing", tfile, "batch_size", "2", "record.loss_freq", "2", "record.eval_freq", "3"],   # validation INSIDE the run (after iteration 3)
                env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29553", "GPSGS_DIST_FORCE": "1"})
    # GPSGS_DIST_FORCE=1: the process group is initialised (backend nccl = RCCL) and the exchange step -- the mean all-reduce of all 5,144,408
    # gradients hooked into GradScaler.unscale_ -- is ISSUED at world size 1: RCCL runs inside the reference's own training loop
    assert res["world_size"] == 1 and res["steps"] == 5 and res["backend"] == "nccl", res
    # Trainer.run_eval ran through the launcher's eval-then-barrier wrapper (train_stage2.py:92-96, :103-139): it writes one preview image per pass
    shows = [f for dp, _, fs in os.walk(str(tmp_path / "exp")) for f in fs if dp.endswith("show")]
    assert shows == ["3.jpg"], shows
    assert res["exchange_overlapped_with_backward"] is True
    t = json.load(open(tfile))
    for k in ("network_forward", "pts2render", "loss_l1", "loss_ssim", "backward", "optimizer_step"):
        n = 6 if k in ("network_forward", "pts2render") else 5   # the validation pass after iteration 3 runs the model and pts2render once more
        assert len(t[k]) == n and min(t[k]) > 0, (k, t[k])


def test_launcher_with_stock_ddp_world_1_over_rccl_matches_the_grad_allreducer(tmp_path):
    """--ddp: the reference's model inside torch's own DistributedDataParallel(find_unused_parameters=True), world size 1 over RCCL, the real networks
    and GPSGS_ACCELERATE=all -- the first time DDP's reducer hooks meet the side streams of the batch render node (render_api._RenderBatch): the
    raster backward runs on one stream per sample while DDP's buckets fire on the autograd thread's stream.  Same seeds, same data: the final
    weights equal those of the GradAllReducer run to the reference's own run-to-run spread (eager atomics, MIOpen); validation runs inside both."""
    import torch
    import make_synthetic_dataset as M
    data_root = str(tmp_path / "data")
    M.make_dataset(data_root, res=256, n_train=2, n_val=2, quiet=True)
    outs = {}
    for tag, extra, port in (("reducer", [], "29561"), ("ddp", ["--ddp"], "29563"), ("reducer2", [], "29565")):
        work = refenv.make_workdir(REF, str(tmp_path / ("w_" + tag)), {"stage1_ckpt": "None", "dataset": {"src_res": 256, "data_root": data_root}})
        final = str(tmp_path / (tag + ".pt"))
        res = _tool([os.path.join(ROOT, "tools", "launch_stage2.py"), "--reference", REF, "--workdir", work, "--steps", "5", "--exp-root", str(tmp_path / ("exp_" + tag)),
                     "--save-final", final] + extra + ["batch_size", "2", "record.loss_freq", "2", "record.eval_freq", "3"],
                    env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "GPSGS_DIST_FORCE": "1", "GPSGS_ACCELERATE": "all"})
        assert res["world_size"] == 1 and res["steps"] == 5 and res["backend"] == "nccl" and res["ddp"] is (tag == "ddp"), res
        shows = [f for dp, _, fs in os.walk(str(tmp_path / ("exp_" + tag))) for f in fs if dp.endswith("show")]
        assert shows == ["3.jpg"], shows
        outs[tag] = torch.load(final, map_location="cpu")

    def rel(a, b):
        num = sum(float(((a[k].double() - b[k].double()) ** 2).sum()) for k in a if a[k].is_floating_point())
        den = sum(float((a[k].double() ** 2).sum()) for k in a if a[k].is_floating_point())
        return (num / den) ** 0.5
    assert set(outs["ddp"]) == set(outs["reducer"]) and not any(k.startswith("module.") for k in outs["ddp"])   # the reference's checkpoint keys
    noise, d = rel(outs["reducer"], outs["reducer2"]), rel(outs["reducer"], outs["ddp"])
    print({"ddp_vs_grad_allreducer_relative_weight_difference": d, "grad_allreducer_run_to_run": noise})
    for v in outs["ddp"].values():
        assert torch.isfinite(v).all()
    assert d <= max(5.0 * noise, 2e-3), (d, noise)


def test_train_stage2_reaches_the_fused_kernels_through_the_import_hook(tmp_path):
    """train_stage2.py as __main__, UNMODIFIED, twice with the same seeds: plain, and with GPSGS_ACCELERATE=all (gps-gaussian_amd/accelerate.py: the
    opt-in import hook rebinds pts2render / l1_loss / ssim / CorrBlockFast1D / upsample_flow / flow2depth / depth2pc to the fused kernels of SURVEY
    section 8(f) as the reference's modules load).  Every replacement must have RUN (call counters), the per-iteration losses the reference logs must
    agree, the final weights too, and `Trainer.run_eval` (train_stage2.py:103-139: eval-mode model -> pts2render under no_grad -> psnr -> cv2.imwrite)
    runs inside both (eval_freq 3)."""
    import torch
    common = ["train", "--res", "256", "--steps", "6", "--batch", "2", "--train-samples", "2", "--eval-freq", "3"]
    plain = _tool([os.path.join(ROOT, "tools", "run_reference.py")] + common + ["--work", str(tmp_path / "plain")])
    again = _tool([os.path.join(ROOT, "tools", "run_reference.py")] + common + ["--work", str(tmp_path / "again")])   # the reference's own run-to-run spread
    fused = _tool([os.path.join(ROOT, "tools", "run_reference.py")] + common + ["--work", str(tmp_path / "fused"), "--accelerate", "all"])
    assert plain["accelerate"]["rebound"] == [] and plain["accelerate"]["calls"] == {}
    acc = fused["accelerate"]
    for name in ("lib.GaussianRender.pts2render", "lib.loss.l1_loss", "lib.loss.ssim", "core.corr.CorrBlockFast1D", "core.raft_stereo_human.CorrBlockFast1D",
                 "core.raft_stereo_human.FlowUpdateModule.upsample_flow", "lib.utils.flow2depth", "lib.utils.depth2pc"):
        assert name in acc["rebound"], (name, acc["rebound"])
    # 6 training iterations + 1 validation batch at step 3: pts2render 7x; l1_loss + ssim 6x each; flow2depth twice per model forward; the convex upsample once per
    # GRU iteration in training (3 iterations: raft.train_iters) and once per validation forward
    assert acc["calls"]["pack"] == 7 and acc["calls"]["loss"] == 12 and acc["calls"]["unproject"] == 14 and acc["calls"]["upsample"] >= 6 * 3 + 1, acc["calls"]
    assert acc["calls"]["corr"] == 7, acc["calls"]   # one correlation block per model forward
    assert acc["calls"]["loss_passthrough"] == 0 and acc["calls"]["unproject_passthrough"] == 0, acc["calls"]   # no call fell through to the reference's eager functions
    for r in (plain, fused):
        assert r["total_steps"] == 6 and r["finite_weights"] and len(r["metrics"]) == 6 and len(r["evals"]) == 1 and r["evals"][0]["val_psnr"] > 0, r
    def spread(x, y):
        w = {}
        for a, b in zip(x["metrics"], y["metrics"]):
            for k in ("l1", "ssim", "train_epe"):
                w[k] = max(w.get(k, 0.0), abs(a[k] - b[k]) / max(abs(a[k]), 1e-6))
        wa = torch.load(x["final_checkpoint"], map_location="cpu")["network"]
        wb = torch.load(y["final_checkpoint"], map_location="cpu")["network"]
        num = sum(float(((wa[k].double() - wb[k].double()) ** 2).sum()) for k in wa if wa[k].is_floating_point())
        den = sum(float((wa[k].double() ** 2).sum()) for k in wa if wa[k].is_floating_point())
        return w, (num / den) ** 0.5

    noise, noise_w = spread(plain, again)      # plain vs plain: atomics in the eager backward kernels, MIOpen's algorithm choices
    worst, rel_w = spread(plain, fused)
    print({"worst_relative_loss_difference": worst, "relative_weight_difference": rel_w, "reference_run_to_run": noise, "reference_run_to_run_weights": noise_w,
           "psnr": (plain["evals"][0]["val_psnr"], fused["evals"][0]["val_psnr"]), "optimizer_steps": (plain["optimizer_steps"], fused["optimizer_steps"]),
           "calls": acc["calls"], "per_step_l1": [(round(a["l1"], 6), round(b["l1"], 6)) for a, b in zip(plain["metrics"], fused["metrics"])]})
    # AMP on both sides (fp16 feature maps, fp16 correlation volumes).  The fused volume kernel accumulates in fp32 on the matrix cores and rounds ONCE to
    # fp16 where the eager path rounds the einsum, the division by sqrt(D) and every level of the average pool separately; the fused SSIM sums in a
    # different order.  Bound: 2e-3 relative on the losses the reference logs (measured 2e-4 .. 7e-4), or 3x the reference's own run-to-run spread where
    # that is larger.  The SSIM term gets 4e-3: six optimiser steps amplify a rounding-level difference of the first step, the run-to-run spread is a
    # sample of ONE pair of runs, and SSIM's variance ratios are the most sensitive of the three (one run in ~10 lands at 2.2e-3 with l1, the end-point
    # error and the final weights inside the reference's own spread).
    for k in ("l1", "ssim", "train_epe"):
        assert worst[k] <= max(4e-3 if k == "ssim" else 2e-3, 3 * noise[k]), (k, worst, noise)
    assert plain["optimizer_steps"] == fused["optimizer_steps"] and rel_w <= max(2e-3, 3 * noise_w), (rel_w, noise_w, plain["optimizer_steps"], fused["optimizer_steps"])
    assert abs(plain["evals"][0]["val_psnr"] - fused["evals"][0]["val_psnr"]) < 0.05


def test_full_size_view_regressed_by_the_real_networks_matches_the_oracle(tmp_path):
    """BASELINE config 3 at FULL size: the reference's test_view_interp.py as __main__ on a 1024^2 stereo pair -> one 2048^2 novel view, its
    real networks (random weights: every scale at the 0.01 m clamp, ~3e7 (Gaussian, bin) instances).  The Gaussians the reference's own pts2render handed to
    the rasteriser for its last view are dumped (GPSGS_DUMP_INPUTS) and the view is rendered again here by the HIP rasteriser and by the CPU oracle: radii equal,
    RGB within 1e-4 away from branch thresholds -- the parity statement on the data the real caller produces, not on a synthetic cloud."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import hip_render, oracle_render
    dump = str(tmp_path / "last_view.npz")
    res = _tool([os.path.join(ROOT, "tools", "run_reference.py"), "interp", "--res", "1024", "--samples", "1", "--views", "2", "--work", str(tmp_path / "w")],
                env={"GPSGS_DUMP_INPUTS": dump, "MIOPEN_FIND_MODE": os.environ.get("MIOPEN_FIND_MODE", "FAST")}, timeout=1500)
    assert res["script_run"]["views"] == 2 and res["render"] == "2048x2048"
    d = np.load(dump)
    P = d["means3D"].shape[0]
    assert P > 300000 and int(d["W"]) == 2048 and int(d["H"]) == 2048
    scene = dict(means3D=d["means3D"], colors=d["colors"], opacities=d["opacities"].reshape(-1, 1), scales=d["scales"], rotations=d["rotations"],
                 view=d["view"].reshape(4, 4), proj=d["proj"].reshape(4, 4), campos=np.zeros(3, np.float32), W=2048, H=2048, tanfovx=float(d["tanfovx"]),
                 tanfovy=float(d["tanfovy"]), bg=d["bg"].astype(np.float32), scale_modifier=float(d["scale_modifier"]))
    assert float(np.median(scene["scales"])) > 0.009          # the heads really sit at their clamp
    img, radii, _, _ = hip_render(scene, debug=True)          # debug: the forward validates its bin lists (ids, order, cursors)
    o, oimg, oradii = oracle_render(scene, "f32")
    np.testing.assert_array_equal(radii, oradii)
    assert o.num_rendered > 5e6
    solid = o.fragility() > 1e-5
    err = np.abs(img - oimg).max(0)
    print({"P": P, "tile_instances": int(o.num_rendered), "median_radius_px": float(np.median(oradii[oradii > 0])), "rgb_max_err_solid": float(err[solid].max()),
           "solid_fraction": float(solid.mean()), "pixels_over_tol": int((err[solid] > 1e-4).sum())})
    # lists are thousands of splats deep: T carries ~1e-4 of accumulated rounding where it meets the 1e-4 stop threshold (see test_gpu_raster.py::
    # test_large_splats_tens_of_millions_of_instances): a handful of pixels outside the 1e-5 band may stop one splat earlier or later than the oracle
    assert solid.mean() > 0.97 and (err[solid] > 1e-4).sum() <= 1e-5 * err.size, ((err[solid] > 1e-4).sum(), float(err[solid].max()))
    assert err.max() <= 2.0 / 255 + 1e-3
