#!/usr/bin/env python
"""Data-parallel launcher around the reference's UNMODIFIED stage-2 trainer.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/launch_stage2.py --reference /path/to/GPS-Gaussian [--config config/stage2.yaml] [--steps K] [KEY VALUE ...]

The reference trains on ONE GPU (/root/reference/train_stage2.py:27-55 has no distributed code).  Its render path shards by stereo
pair with no data-path collective, so data parallelism is one process per GPU, each running the reference's own `Trainer` on its
shard of the dataset, plus ONE exchange step per iteration: a mean all-reduce of the 5,144,408 network gradients (20.6 MB, one
32 MiB bucket -- xGMI is point-to-point, ring collectives are per-link bound, so one large message) over RCCL (backend "nccl";
"gloo" on CPU) -- by this package's GradAllReducer (default), or by torch's stock DistributedDataParallel (--ddp).  Nothing in the
reference is edited; this file only
  * puts gps-gaussian_amd/dropin (the MI355X rasteriser / correlation sampler under the reference's import names) and then the
    reference on sys.path, plus stand-ins for yacs / cv2 / tensorboard when -- and only when -- the real packages are missing;
  * imports `train_stage2` WITHOUT executing its `__main__` block and repeats that block (train_stage2.py:183-207), with one repair the
    reference needs anyway: `Trainer` reads a module-global `cfg` (train_stage2.py:44,76,101,123) that only exists when the file runs
    as a script, so the launcher sets it on the imported module;
  * swaps the module's `DataLoader` for one that shards the training set with a `DistributedSampler` (per-rank batch size =
    cfg.batch_size: weak scaling; the validation loader stays whole and only rank 0 evaluates);
  * hooks the exchange step into the one place every iteration passes between backward and the optimizer:
    `GradScaler.unscale_(optimizer)` (train_stage2.py:84) first all-reduces the (still scaled) gradients -- parameters that received
    none (the reference constructs gru16 / gru32 but never runs them, core/update.py:105-106) are sent as zeros, which is what
    DDP(find_unused_parameters=True) would amount to -- so clip_grad_norm_, the inf check and the step see identical numbers on
    every rank and the GradScaler's skip decisions agree;
  * keeps logging, checkpoints and previews on rank 0 (the other ranks get a silent logger and no-op save / eval), seeds
    every rank differently for data order but identically for the model (torch.manual_seed(1314) before `Trainer(cfg)` as the
    reference does, rank offset afterwards).
"""
import argparse
import importlib
import json
import logging
import os
import sys
import types
from datetime import datetime
from pathlib import Path

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import refenv  # noqa: E402  (shims, reference discovery: shared with tools/run_reference.py)


class _SilentLogger:
    """What ranks > 0 hand to the reference's `self.logger`: same methods, no files."""

    class _W:
        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass

    def __init__(self):
        self.writer, self.total_steps = self._W(), 0

    def push(self, metrics):
        self.total_steps += 1

    def write_dict(self, results, write_step):
        pass

    def close(self):
        pass


def _install_timing(TS, trainer, torch):
    """hipEvent brackets around the calls of the reference's own training iteration (train_stage2.py:57-97), by wrapping the names its
    loop resolves at call time: `self.model(...)`, module-level `pts2render`, `l1_loss`, `ssim`, `self.scaler.scale(loss)` ->
    `.backward()`, `self.scaler.step`.  Returns a function producing {span: [ms per iteration ...], "iter_ms": [...]}."""
    import time
    spans, stamps = {}, []

    def bracket(name, fn):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            spans.setdefault(name, []).append((e0, e1))
            return out
        return timed

    trainer.model.forward = bracket("network_forward", trainer.model.forward)
    TS.pts2render = bracket("pts2render", TS.pts2render)
    TS.l1_loss = bracket("loss_l1", TS.l1_loss)
    TS.ssim = bracket("loss_ssim", TS.ssim)
    real_scale = trainer.scaler.scale

    def scale(loss):
        out = real_scale(loss)
        out.backward = bracket("backward", out.backward)
        return out
    trainer.scaler.scale = scale
    real_step = trainer.scaler.step

    def step(opt, *a, **k):
        stamps.append(time.perf_counter())
        return bracket("optimizer_step", real_step)(opt, *a, **k)
    trainer.scaler.step = step

    def result():
        torch.cuda.synchronize()
        out = {k: [round(a.elapsed_time(b), 3) for a, b in v] for k, v in spans.items()}
        out["iter_ms"] = [round((stamps[i] - stamps[i - 1]) * 1e3, 3) for i in range(1, len(stamps))]
        return out
    return result


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default=None, help="checkout of aipixel/GPS-Gaussian (unmodified); default: $GPSGS_REFERENCE, /root/reference")
    ap.add_argument("--config", default="config/stage2.yaml", help="relative to the working directory (--workdir, default: the reference)")
    ap.add_argument("--workdir", default=None, help="cwd for the reference's cwd-relative paths: a scratch directory made by tools/refenv.make_workdir "
                                                    "(links to the reference + its own config/stage2.yaml); default: the reference itself")
    ap.add_argument("--timing", default=None, help="write per-iteration hipEvent timings of the reference's own calls (model, pts2render, l1_loss + ssim, "
                                                   "backward, optimizer) as JSON to this file (rank 0)")
    ap.add_argument("--steps", type=int, default=None, help="override cfg.num_steps")
    ap.add_argument("--backend", default=None, help="nccl (= RCCL, default on GPUs) or gloo")
    ap.add_argument("--exp-root", default=None, help="where experiments/<name>/ goes (default: the reference's cwd-relative 'experiments')")
    ap.add_argument("--ddp", action="store_true", help="wrap the reference's model in torch's stock DistributedDataParallel(find_unused_parameters=True) instead of this "
                                                       "package's GradAllReducer: DDP's reducer hooks then issue the gradient all-reduces (25 MiB buckets) during the backward")
    ap.add_argument("--save-final", default=None, help="rank 0: torch.save the model's final state_dict here (tests compare the two exchange implementations)")
    ap.add_argument("--no-overlap", action="store_true", help="issue the gradient all-reduce after the whole backward (one 32 MiB bucket) instead of from "
                                                              "post-accumulate hooks while the backward of the earlier layers is still running (8 MiB buckets)")
    ap.add_argument("--pg-timeout-s", type=float, default=None, help="watchdog timeout of the collectives (default $GPSGS_PG_TIMEOUT_S, else 1800): must cover rank 0's "
                                                                     "validation pass, during which the other ranks wait at a barrier")
    ap.add_argument("--hook", default=None, help="python file exec'd after `train_stage2` is imported and before Trainer(cfg) is built, with TS (the "
                                                 "module), cfg, rank, world in scope: synthetic data sets, smoke tests")
    ap.add_argument("overrides", nargs="*", help="KEY VALUE pairs merged into the config (yacs merge_from_list syntax)")
    args = ap.parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import DROPIN_DIR
    from gps_gaussian_amd import dist as D

    ref = refenv.reference_dir(args.reference)
    if ref is None:
        raise SystemExit("launch_stage2: no reference checkout (looked at --reference, $GPSGS_REFERENCE, /root/reference)")
    rank, local_rank, world = D.env_rank()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if use_cuda else None
    D.init(backend=args.backend, device=dev, timeout_s=args.pg_timeout_s)  # no-op at world size 1
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus = D.set_cpu_affinity(local_rank, local_world, local_rank if use_cuda else None)  # this rank (and the DataLoader workers it forks) on its own CPU slice

    refenv.activate(ref)            # shims if needed; DROPIN_DIR (`import diff_gaussian_rasterization` / `import corr_sampler` -> the MI355X
    assert sys.path[0] == DROPIN_DIR  # kernels) ahead of the reference on sys.path
    os.chdir(args.workdir or ref)   # the reference uses cwd-relative paths ("config/stage2.yaml", file_backup's 'core', 'lib', ...)

    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING,
                        format="%(asctime)s %(levelname)-8s [rank " + str(rank) + " %(filename)s:%(lineno)d] %(message)s")
    TS = importlib.import_module("train_stage2")  # defines Trainer; its __main__ block does not run

    # ---- the reference's __main__ block (train_stage2.py:183-207) ------------------------------------------------------------
    cfg = TS.config()
    cfg.load(args.config)
    cfg = cfg.get_cfg()
    cfg.defrost()
    if args.overrides:
        cfg.merge_from_list(args.overrides)
    if args.steps is not None:
        cfg.num_steps = args.steps
    dt = datetime.today()
    cfg.exp_name = "%s_%s%s" % (cfg.name, str(dt.month).zfill(2), str(dt.day).zfill(2))
    exp = os.path.join(args.exp_root or "experiments", cfg.exp_name)
    cfg.record.ckpt_path, cfg.record.show_path = "%s/ckpt" % exp, "%s/show" % exp
    cfg.record.logs_path, cfg.record.file_path = "%s/logs" % exp, "%s/file" % exp
    cfg.freeze()
    if rank == 0:
        for path in (cfg.record.ckpt_path, cfg.record.show_path, cfg.record.logs_path, cfg.record.file_path):
            Path(path).mkdir(exist_ok=True, parents=True)
        TS.file_backup(cfg.record.file_path, cfg, train_script=os.path.basename(TS.__file__))  # train_stage2.py:204
    D.barrier(local_rank if use_cuda else None)
    TS.cfg = cfg  # Trainer's methods read this module global (train_stage2.py:44,76,101,123)

    # ---- shard the training set ---------------------------------------------------------------------------------------------------
    samplers = []
    RealLoader = TS.DataLoader

    from torch.utils.data.distributed import DistributedSampler

    class EpochSampler(DistributedSampler):
        """Every time the reference re-creates its iterator (train_stage2.py:157-160) a new epoch's shuffle starts."""
        _epoch_no = 0

        def __iter__(self):
            self.set_epoch(self._epoch_no)
            self._epoch_no += 1
            return super().__iter__()

    def sharded_loader(dataset, batch_size=1, shuffle=False, **kw):
        if world > 1 and shuffle:  # the training loader (the validation loader is built with shuffle=False and stays whole)
            s = EpochSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=1314, drop_last=True)
            samplers.append(s)
            return RealLoader(dataset, batch_size=batch_size, shuffle=False, sampler=s, **kw)
        return RealLoader(dataset, batch_size=batch_size, shuffle=shuffle, **kw)

    TS.DataLoader = sharded_loader
    if rank != 0:
        TS.Logger = lambda scheduler, rec: _SilentLogger()
        TS.tqdm = lambda it, *a, **k: it

    if args.hook:
        exec(compile(open(args.hook).read(), args.hook, "exec"), {"TS": TS, "cfg": cfg, "rank": rank, "world": world, "__name__": "launch_stage2_hook"})

    torch.manual_seed(1314)  # identical initial weights on every rank (train_stage2.py:206-207)
    np.random.seed(1314)
    trainer = TS.Trainer(cfg)
    torch.manual_seed(1314 + 1000 * rank)  # from here on: per-rank randomness
    np.random.seed(1314 + 1000 * rank)

    # ---- the one exchange step ------------------------------------------------------------------------------------------------------
    reducer = None
    if args.ddp:
        # torch's own DistributedDataParallel around the reference's model (what the north star names).  find_unused_parameters: the reference
        # constructs gru16 / gru32 but never runs them (core/update.py:105-106).  The Trainer keeps talking to `self.model` as before -- it calls
        # self.model.raft_stereo.freeze_bn() (train_stage2.py:54,95) and saves self.model.state_dict() (:178) -- so the wrapper forwards unknown
        # attributes to the wrapped module and keeps the reference's checkpoint keys (no "module." prefix).
        if not dist.is_initialized():
            raise SystemExit("launch_stage2 --ddp needs a process group (run under torch.distributed.run, or GPSGS_DIST_FORCE=1 at world size 1)")
        from torch.nn.parallel import DistributedDataParallel

        class _DDP(DistributedDataParallel):
            def __getattr__(self, name):
                try:
                    return super().__getattr__(name)
                except AttributeError:
                    return getattr(super().__getattr__("module"), name)

            def state_dict(self, *a, **k):
                return self.module.state_dict(*a, **k)

            def load_state_dict(self, *a, **k):
                return self.module.load_state_dict(*a, **k)

        # broadcast_buffers=False: DDP's forward would otherwise broadcast registered buffers from rank 0 on every call -- a collective the ranks that
        # do NOT evaluate never issue (the reference's model happens to have no buffers: GroupNorm / frozen BN)
        trainer.model = _DDP(trainer.model, device_ids=[local_rank] if use_cuda else None, find_unused_parameters=True, gradient_as_bucket_view=True,
                             broadcast_buffers=False)
    else:
        reducer = D.GradAllReducer(trainer.model.parameters(), overlap=not args.no_overlap)
        real_unscale = trainer.scaler.unscale_

        def unscale_after_allreduce(optimizer):
            reducer()  # mean over ranks of the (scaled) gradients; unused parameters travel as zeros
            return real_unscale(optimizer)

        trainer.scaler.unscale_ = unscale_after_allreduce
    timing = _install_timing(TS, trainer, torch) if (args.timing and rank == 0 and use_cuda) else None
    # validation (train_stage2.py:92-96 -> run_eval) stays on rank 0, like logging and checkpoints -- but every rank passes through it: the others
    # wait at an EXPLICIT barrier until rank 0 is done, instead of running ahead into the next iteration's all-reduce and sitting in a collective
    # for as long as the validation set takes (an RCCL watchdog timeout on a real validation set; the barrier is covered by --pg-timeout-s)
    real_eval = trainer.run_eval

    def eval_then_barrier(*a, **k):
        out = None
        if rank == 0:
            wrapped = trainer.model
            if args.ddp:
                trainer.model = wrapped.module  # rank-0-only validation runs on the PLAIN module: no DDP forward, so nothing collective
            try:
                out = real_eval(*a, **k)
            finally:
                trainer.model = wrapped
        D.barrier(local_rank if use_cuda else None)
        return out
    trainer.run_eval = eval_then_barrier
    if rank != 0:
        trainer.save_ckpt = lambda *a, **k: None
    trainer.train()
    D.barrier(local_rank if use_cuda else None)
    if timing is not None:
        with open(args.timing, "w") as f:
            json.dump(timing(), f)
    if rank == 0 and args.save_final:
        torch.save({k: v.detach().cpu() for k, v in trainer.model.state_dict().items()}, args.save_final)
    if rank == 0:
        n_par = sum(p.numel() for p in trainer.model.parameters() if p.requires_grad)
        print(json.dumps({"launcher": "launch_stage2", "world_size": world, "steps": int(trainer.total_steps),
                          "exchange": ("torch DistributedDataParallel(find_unused_parameters=True) over %d gradients" % n_par) if args.ddp else
                                      "mean all-reduce of %d gradients in %d bucket(s)" % (sum(p.numel() for p in reducer.params), len(reducer.buckets)),
                          "ddp": bool(args.ddp), "backend": dist.get_backend() if dist.is_initialized() else None,
                          "exchange_overlapped_with_backward": True if args.ddp else bool(reducer.overlap), "cpu_affinity": (len(cpus) if cpus else None)}))
    D.shutdown()


if __name__ == "__main__":
    main()
