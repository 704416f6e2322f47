#!/usr/bin/env python
"""SCALE-shaped dry run on a ONE-GPU box: runs bench.py at N = 1, 2, 4, 8 back to back exactly as the driver's scaling bench does
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`)
and prints, per N, the JSON record the driver would parse.  With fewer GPUs than N every rank shares device 0 and talks over gloo
(GPSGS_BENCH_SINGLE_DEVICE=1, GPSGS_BENCH_BACKEND=gloo): a FUNCTIONAL rehearsal of the multi-rank path -- env rendezvous, per-rank workload,
barrier + MAX-over-ranks timing, whole-job aggregate, the stage-2 leg's gradient all-reduce -- so that the first real 8-GPU lease is a
measurement and not a debug session.  The values it prints at N > 1 on one GPU are NOT scaling numbers (the ranks time-share one chip) and say so.

    python tools/scale_dry_run.py [--ns 1,2,4,8] [--steps 5] [--warmup 2] [--res 256 --gaussians 30000]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--gaussians", type=int, default=30000)
    ap.add_argument("--port", type=int, default=29571)
    a = ap.parse_args()
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    recs = []
    for i, n in enumerate(int(x) for x in a.ns.split(",")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        shared = n > have
        if shared:
            env.update(GPSGS_BENCH_SINGLE_DEVICE="1", GPSGS_BENCH_BACKEND="gloo")
        bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup), "--res", str(a.res), "--gaussians", str(a.gaussians),
                 "--no-full-pipeline", "--no-configs"]
        cmd = ([sys.executable] + bench) if n == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                                                        "--master-port", str(a.port + i)] + bench
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, timeout=1800)
        wall = time.perf_counter() - t0
        lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
        rec = {"n": n, "rc": r.returncode, "driver_run_s": round(wall, 1), "ranks_share_one_device": shared}
        if r.returncode == 0 and len(lines) == 1:
            d = json.loads(lines[0])
            rec["parsed"] = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
            rec["parsed"]["stage2_path"] = {k: (d.get("stage2_path") or {}).get(k) for k in ("iters_per_s", "n_gpus", "allreduce", "allreduce_alone_ms")}
            rec["consistency"] = {"n_gpus_claimed": d.get("n_gpus"), "steps_match": d.get("steps") == a.steps, "warmup_match": d.get("warmup") == a.warmup,
                                  "fits_in_driver_run": d.get("ms_per_step", 0) * a.steps * 1e-3 <= wall}
        else:
            rec["error"] = (r.stderr or r.stdout)[-600:]
        recs.append(rec)
        print(json.dumps(rec), flush=True)
    ok = all(r.get("rc") == 0 and "parsed" in r for r in recs)
    print(json.dumps({"scale_dry_run": "ok" if ok else "FAILED", "ns": [r["n"] for r in recs],
                      "note": "a functional rehearsal: with ranks sharing one device the per-N values are not scaling numbers; no N > 1 MI355X measurement exists"}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
