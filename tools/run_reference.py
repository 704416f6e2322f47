#!/usr/bin/env python
"""Run the reference's OWN scripts, unmodified, on the MI355X drop-in and time them (BASELINE configs 3 and 4; harness, not product).

    python tools/run_reference.py interp [--samples 2] [--views 5] [--res 1024]      test_view_interp.py   (config 3)
    python tools/run_reference.py real [--samples 2] [--ratio 0.5] [--res 1024]       test_real_data.py     (one novel view per frame)
    python tools/run_reference.py train  [--steps 12] [--batch 2] [--res 1024]       train_stage2.py       (config 4, one process)

The reference is found by tools/refenv.reference_dir(): --reference, $GPSGS_REFERENCE or /root/reference -- a CHECKOUT (nothing of the reference
travels to the GPU box: this tool runs where a checkout and an MI355X coexist).  The script is executed as `__main__` by runpy -- the same code object
`python test_view_interp.py ...` would run -- from a scratch working directory (tools/refenv.make_workdir) with
gps-gaussian_amd/dropin ahead of the reference on sys.path.  What the harness supplies, and nothing else:
  * a synthetic data set in the loader's on-disk layout (tools/make_synthetic_dataset.py) -- the reference's real
    `StereoHumanDataset` reads, rectifies and collates it;
  * randomly initialised network weights saved as the checkpoint `--ckpt_path` asks for (no pretrained weights offline);
  * stand-ins for yacs / cv2 / tensorboard when the packages are missing (tools/shims);
  * clocks: the time of every cv2.imwrite of a rendered view (interp), of every optimizer step (train), and -- in a second, INSTRUMENTED
    pass that imports the script as a module -- hipEvent brackets around the reference's `model(...)`, `pts2render(...)` and loss calls.
Prints one JSON line.
"""
import argparse
import json
import os
import runpy
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refenv  # noqa: E402


def _dataset(work, res, n_train, n_val, fill):
    import make_synthetic_dataset as M
    root = os.path.join(work, "data_%d_%d_%d_%g" % (res, n_train, n_val, fill))
    if not os.path.isdir(os.path.join(root, "val", "img")):
        M.make_dataset(root, res=res, n_train=n_train, n_val=n_val, fill=fill, quiet=True)
    return root


def _cfg(work, yaml_name="stage2.yaml"):
    from config.stereo_human_config import ConfigStereoHuman
    c = ConfigStereoHuman()
    c.load(os.path.join(work, "config", yaml_name))
    return c.get_cfg()


class _Clock:
    """hipEvent brackets around a callable; totals read after a synchronize."""

    def __init__(self, torch):
        self.torch, self.spans = torch, {}

    def wrap(self, name, fn):
        torch = self.torch

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.spans.setdefault(name, []).append((e0, e1))
            return out
        return timed

    def totals(self, skip=0):
        self.torch.cuda.synchronize()
        return {k: (round(sum(a.elapsed_time(b) for a, b in v[skip:]) / max(1, len(v[skip:])), 3), len(v[skip:])) for k, v in self.spans.items()}


def _accel_report():
    """What the opt-in hook rebound in this process and how often each replacement ran ({} when GPSGS_ACCELERATE is unset: the module is not even imported)."""
    acc = sys.modules.get("gps_gaussian_amd.accelerate")
    if acc is None:
        return {"requested": os.environ.get("GPSGS_ACCELERATE", ""), "rebound": [], "calls": {}}
    return {"requested": os.environ.get("GPSGS_ACCELERATE", ""), "rebound": sorted(acc.installed()), "calls": dict(acc.calls)}


def interp(args):
    import numpy as np
    import torch

    ref = refenv.reference_dir(args.reference)
    if ref is None:
        raise SystemExit("run_reference: no reference checkout (--reference, $GPSGS_REFERENCE, /root/reference)")
    refenv.activate(ref)
    work = os.path.abspath(args.work)
    data_root = _dataset(work, args.res, 0, args.samples, args.fill)
    refenv.make_workdir(ref, work, {"dataset": {"src_res": args.res}})
    os.chdir(work)
    cfg = _cfg(work)
    # randomly initialised weights in the checkpoint format the script loads (test_view_interp.py:61-66)
    from lib.network import RtStereoHumanModel
    torch.manual_seed(1314)
    ckpt = os.path.join(work, "random_init_stage2.pth")
    torch.save({"network": RtStereoHumanModel(cfg, with_gs_render=True).state_dict()}, ckpt)

    import cv2
    stamps = []
    real_imwrite = cv2.imwrite

    def imwrite(path, img):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        return real_imwrite(path, img) if args.write_images else True
    cv2.imwrite = imwrite

    test_root = os.path.join(data_root, "val")
    argv = ["test_view_interp", "--test_data_root", test_root, "--ckpt_path", ckpt, "--novel_view_nums", str(args.views)]
    out = {"mode": "interp", "reference": ref, "script": os.path.basename(refenv.script(ref, "test_view_interp")), "res": args.res,
           "render": "%dx%d" % (2 * args.res, 2 * args.res), "samples": args.samples, "views_per_sample": args.views}
    # ---- pass 1: the script itself, as __main__ ----------------------------------------------------------------------------------
    for rep in range(2):  # the second run is the measured one (first: MIOpen / allocator / capacity warm-up)
        del stamps[:]
        old = sys.argv
        sys.argv = argv
        t0 = time.perf_counter()
        try:
            runpy.run_path(refenv.script(ref, "test_view_interp"), run_name="__main__")
        finally:
            sys.argv = old
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    n = len(stamps)
    assert n == args.samples * args.views, (n, args.samples, args.views)
    out["script_run"] = {"views": n, "wall_s": round(wall, 3), "views_per_s_end_to_end": round(n / wall, 3),
                         "note": "whole script: model construction + checkpoint load + per-sample loader (disk, rectification on the host) + "
                                 "per-view network + render + .cpu(); the JPEG write itself is %s" % ("included" if args.write_images else "skipped")}
    # views inside one sample follow each other without the loader in between
    gaps = [stamps[i] - stamps[i - 1] for i in range(1, n) if i % args.views]
    if gaps:
        out["script_run"]["views_per_s_within_sample"] = round(1.0 / float(np.median(gaps)), 3)

    # ---- pass 2: instrumented (module import; same classes, hipEvents around the reference's own calls) ----------------------------
    import importlib
    TVI = importlib.import_module("test_view_interp")
    import lib.GaussianRender as GR
    import gaussian_renderer as GRR
    import diff_gaussian_rasterization as DGR
    import gps_gaussian_amd.rasterizer as RZ
    assert DGR.GaussianRasterizer is RZ.GaussianRasterizer and GRR.GaussianRasterizer is RZ.GaussianRasterizer
    clk = _Clock(torch)
    cfg.defrost()
    cfg.batch_size = 1
    cfg.dataset.test_data_root = test_root
    cfg.dataset.use_processed_data = False
    cfg.restore_ckpt = ckpt
    cfg.test_out_path = os.path.join(work, "interp_out")
    os.makedirs(cfg.test_out_path, exist_ok=True)
    cfg.freeze()
    r = TVI.StereoHumanRender(cfg, phase="test")
    r.model.forward = clk.wrap("network", r.model.forward)
    TVI.pts2render = clk.wrap("pts2render", TVI.pts2render)
    GR.render = clk.wrap("render", GR.render)
    Ps = []
    real_fwd = RZ.GaussianRasterizer.forward  # the module's entry: behind it the compiled host path or the Python one, whichever applies

    def spy(self, means3D, *a, **k):
        Ps.append(int(means3D.shape[0]))
        return real_fwd(self, means3D, *a, **k)
    RZ.GaussianRasterizer.forward = spy
    real_impl = RZ._forward_impl              # (with GPSGS_ACCELERATE=pack the batch node calls _forward_impl itself: row-range views)

    def spy_impl(ctx, means3D, *a, **k):
        rows = k.get("rows")
        Ps.append(int(rows.capacity if rows is not None else means3D.shape[0]))
        return real_impl(ctx, means3D, *a, **k)
    RZ._forward_impl = spy_impl
    del stamps[:]
    r.infer_static(view_select=[0, 1], novel_view_nums=args.views)
    tot = clk.totals(skip=1)
    RZ.GaussianRasterizer.forward = real_fwd
    RZ._forward_impl = real_impl
    if not Ps:
        Ps.append(0)
    out["gpu_ms_per_view"] = {k: v[0] for k, v in tot.items()}
    if "render" in tot:  # (with GPSGS_ACCELERATE=pack the reference's render() is not on the path: pts2render is the fused pack + batch node)
        out["gpu_ms_per_view"]["pack_inside_pts2render"] = round(tot["pts2render"][0] - tot["render"][0], 3)
    out["gaussians_per_view"] = {"mean": int(np.mean(Ps)), "min": int(min(Ps)), "max": int(max(Ps))}
    gpu_ms = tot["network"][0] + tot["pts2render"][0]
    out["views_per_s_gpu_side"] = round(1e3 / gpu_ms, 2)
    if "render" in tot:
        out["raster_share_of_gpu_time"] = round(tot["render"][0] / gpu_ms, 4)
    out["accelerate"] = _accel_report()
    img = None
    try:
        from PIL import Image
        files = sorted(os.listdir(cfg.test_out_path))
        if files and args.write_images:
            img = np.asarray(Image.open(os.path.join(cfg.test_out_path, files[0])))
            out["first_image"] = {"file": files[0], "shape": list(img.shape), "nonblack_fraction": round(float((img.max(-1) > 8).mean()), 4)}
    except Exception as e:  # noqa: BLE001
        out["first_image"] = {"error": repr(e)[:200]}
    print(json.dumps(out))
    return out


def real(args):
    """test_real_data.py as `__main__`: the reference's third entry point (one novel view per captured frame, `get_novel_calib` at --ratio between the
    two source cameras, `pts2render`, JPEG per frame) on the synthetic set in the loader's on-disk layout."""
    import numpy as np
    import torch

    ref = refenv.reference_dir(args.reference)
    if ref is None:
        raise SystemExit("run_reference: no reference checkout (--reference, $GPSGS_REFERENCE, /root/reference)")
    refenv.activate(ref)
    work = os.path.abspath(args.work)
    data_root = _dataset(work, args.res, 0, args.samples, args.fill)
    refenv.make_workdir(ref, work, {"dataset": {"src_res": args.res}})
    os.chdir(work)
    cfg = _cfg(work)
    from lib.network import RtStereoHumanModel
    torch.manual_seed(1314)
    ckpt = os.path.join(work, "random_init_stage2.pth")
    torch.save({"network": RtStereoHumanModel(cfg, with_gs_render=True).state_dict()}, ckpt)
    import cv2
    stamps, written = [], []
    real_imwrite = cv2.imwrite

    def imwrite(path, img):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        written.append((path, tuple(img.shape), float((np.asarray(img).max(-1) > 8).mean())))
        return real_imwrite(path, img) if args.write_images else True
    cv2.imwrite = imwrite
    argv = ["test_real_data", "--test_data_root", os.path.join(data_root, "val"), "--ckpt_path", ckpt, "--src_view", "0", "1", "--ratio", str(args.ratio)]
    for rep in range(2):  # the second run is the measured one
        del stamps[:], written[:]
        old = sys.argv
        sys.argv = argv
        t0 = time.perf_counter()
        try:
            runpy.run_path(refenv.script(ref, "test_real_data"), run_name="__main__")
        finally:
            sys.argv = old
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    out = {"mode": "real", "reference": ref, "script": os.path.basename(refenv.script(ref, "test_real_data")), "res": args.res,
           "render": "%dx%d" % (2 * args.res, 2 * args.res), "frames": len(written), "wall_s": round(wall, 3),
           "frames_per_s_end_to_end": round(len(written) / wall, 3), "ratio": args.ratio,
           "images": [{"file": os.path.basename(p_), "shape": list(sh), "nonblack_fraction": round(nb, 4)} for p_, sh, nb in written]}
    print(json.dumps(out))
    return out


def train(args):
    """train_stage2.py as __main__, one process, the reference's real data set class on the synthetic set, real networks, random init."""
    import numpy as np
    import torch

    ref = refenv.reference_dir(args.reference)
    if ref is None:
        raise SystemExit("run_reference: no reference checkout (--reference, $GPSGS_REFERENCE, /root/reference)")
    refenv.activate(ref)
    work = os.path.abspath(args.work)
    data_root = _dataset(work, args.res, args.train_samples, 2, args.fill)
    refenv.make_workdir(ref, work, {"stage1_ckpt": "None", "batch_size": args.batch, "num_steps": args.steps, "dataset": {"src_res": args.res, "data_root": data_root},
                                    "record": {"loss_freq": max(2, args.steps // 2), "eval_freq": args.eval_freq or 10 * args.steps}})
    os.chdir(work)
    stamps = []
    real_step = torch.optim.AdamW.step

    def step(self, *a, **k):
        stamps.append(time.perf_counter())
        return real_step(self, *a, **k)
    torch.optim.AdamW.step = step
    # the per-iteration metrics the reference's own loop hands to its Logger (train_stage2.py:78-82: l1, ssim, train_epe ...), unrounded
    import lib.train_recoder as TR
    pushed = []
    real_push = TR.Logger.push

    def push(self, metrics):
        pushed.append({k: float(v) for k, v in metrics.items()})
        return real_push(self, metrics)
    TR.Logger.push = push
    evals = []
    real_write = TR.Logger.write_dict

    def write_dict(self, results, write_step):
        evals.append({"step": int(write_step), **{k: float(v) for k, v in results.items()}})
        return real_write(self, results, write_step)
    TR.Logger.write_dict = write_dict
    t0 = time.perf_counter()
    runpy.run_path(refenv.script(ref, "train_stage2"), run_name="__main__")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    torch.optim.AdamW.step = real_step
    TR.Logger.push, TR.Logger.write_dict = real_push, real_write
    n = len(stamps)
    tail = stamps[n // 2:]
    exp = os.path.join(work, "experiments")
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(exp) for f in fs if f.endswith("_final.pth")]
    out = {"mode": "train", "reference": ref, "script": os.path.basename(refenv.script(ref, "train_stage2")), "res": args.res, "batch": args.batch,
           "optimizer_steps": n, "wall_s": round(wall, 2),
           "iters_per_s_second_half": round((len(tail) - 1) / (tail[-1] - tail[0]), 3) if len(tail) > 1 else None,
           "final_checkpoint_written": bool(ck), "metrics": pushed, "evals": evals, "accelerate": _accel_report(),
           "final_checkpoint": ck[-1] if ck else None}
    if ck:
        sd = torch.load(ck[-1], map_location="cpu")
        out["finite_weights"] = bool(all(torch.isfinite(v).all() for v in sd["network"].values() if v.is_floating_point()))
        out["total_steps"] = int(sd["total_steps"])
    print(json.dumps(out))
    return out


def ddp(args):
    """tools/launch_stage2.py (the data-parallel entry; world size from the torchrun environment, 1 without it) around the reference's
    Trainer with the REAL networks, data set class and losses; per-iteration hipEvent spans of the reference's own calls."""
    import numpy as np

    ref = refenv.reference_dir(args.reference)
    if ref is None:
        raise SystemExit("run_reference: no reference checkout (--reference, $GPSGS_REFERENCE, /root/reference)")
    work = os.path.abspath(args.work)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        data_root = _dataset(work, args.res, args.train_samples, 2, args.fill)
        refenv.make_workdir(ref, work, {"stage1_ckpt": "None", "dataset": {"src_res": args.res, "data_root": data_root}})
    import launch_stage2
    tfile = os.path.join(work, "timing_rank0.json")
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        launch_stage2.main(["--reference", ref, "--workdir", work, "--steps", str(args.steps), "--exp-root", os.path.join(work, "experiments"), "--timing", tfile,
                            "batch_size", str(args.batch), "record.loss_freq", str(max(2, args.steps // 2)), "record.eval_freq", str(args.eval_freq or 10 * args.steps)])
    if rank != 0:
        return None
    line = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
    t = json.load(open(tfile))
    half = len(t["iter_ms"]) // 2
    med = lambda v: round(float(np.median(v[len(v) // 2:])), 3)  # noqa: E731  (second half: after MIOpen / allocator / capacity warm-up)
    out = {"mode": "ddp", "reference": ref, "res": args.res, "render": "%dx%d" % (2 * args.res, 2 * args.res), "batch_per_gpu": args.batch, "world_size": line["world_size"],
           "steps": line["steps"], "backend": line["backend"], "exchange": line["exchange"],
           "iter_ms_median_second_half": med(t["iter_ms"]), "iters_per_s": round(1e3 / med(t["iter_ms"]), 3),
           "gpu_ms_per_iter": {k: med(v) for k, v in t.items() if k != "iter_ms"}, "accelerate": _accel_report(),
           "note": "iter_ms = host time between optimizer steps (includes the DataLoader and the reference's per-iteration .item() syncs); gpu_ms_per_iter = "
                   "hipEvent spans around the reference's own calls: network_forward = RtStereoHumanModel (RAFT-Stereo + regressor, AMP), pts2render = the "
                   "reference's per-sample mask gathers + render() -> HIP rasteriser forward, loss_* = lib/loss.py, backward = autograd incl. the HIP rasteriser backward"}
    print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("mode", choices=("interp", "real", "train", "ddp"))
    ap.add_argument("--reference", default=None)
    ap.add_argument("--work", default=os.environ.get("GPSGS_REF_WORK", "/tmp/gpsgs_ref_work"))
    ap.add_argument("--res", type=int, default=1024, help="source resolution (dataset.src_res); the render is 2x that (use_hr_img)")
    ap.add_argument("--fill", type=float, default=0.28, help="fraction of a source image the synthetic human covers (~P / (2 res^2))")
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--train-samples", type=int, default=4)
    ap.add_argument("--eval-freq", type=int, default=0)
    ap.add_argument("--write-images", action="store_true")
    ap.add_argument("--ratio", type=float, default=0.5, help="real: position of the novel camera between the two source cameras")
    ap.add_argument("--accelerate", default=None, help="value for GPSGS_ACCELERATE (gps-gaussian_amd/accelerate.py: e.g. 'all' or 'pack,loss'): the opt-in "
                                                       "import hook that lets the UNMODIFIED script reach the fused pack / loss / corr / upsample / unproject kernels")
    args = ap.parse_args()
    if args.accelerate is not None:
        os.environ["GPSGS_ACCELERATE"] = args.accelerate   # read by the drop-in shims when the reference imports them
    return {"interp": interp, "real": real, "train": train, "ddp": ddp}[args.mode](args)


if __name__ == "__main__":
    main()
