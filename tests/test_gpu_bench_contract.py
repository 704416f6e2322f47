"""GPU (-m gpu): bench.py prints ONE JSON line with the fields the driver's contract names (small workload, few steps)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line_has_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--res", "256", "--gaussians", "30000"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "views/s" and d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 0.02
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the roofline's duration is the EXCLUSIVE one (one view in flight): a kernel of the step cannot last longer than that step, and
    # achieved = algorithmic bytes (instances counted on upstream's 16x16 tiles) / that duration; the same with the
    # implementation's own 8x8-bin instance count, and the duration with several views in flight, are listed next to it
    step_one_view_ms = 1e3 / d["single_view_in_flight_views_per_s"]
    assert 0 < rf["avg_launch_us"] * 1e-3 <= step_one_view_ms * 1.05 and rf["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.05
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) <= 0.01 * rf["achieved"] + 0.1
    assert rf["instances"]["tile_16x16"] > 0 and rf["instances"]["bin_8x8"] > 0 and rf["frac_with_bin_8x8_instances"] > 0
    assert rf["instances"]["gradient_records_written"] <= rf["instances"]["bin_8x8"] <= rf["instances"]["gradient_record_slots"]
    hr = rf["views_in_flight_region"]
    assert hr["views_in_flight"] >= 1 and (hr["avg_launch_us"] is None or hr["avg_launch_us"] > 0)
    # `value` is the reference's plugin API (GaussianRasterizer + backward, one view at a time); the caller-owned-buffer sessions are an extra
    assert d["metric"] == "novel views/sec at 1024x1024 (~600k Gaussians), 1 GPU" and d["value"] == d["autograd_api_views_per_s"]
    assert d["session"]["views_in_flight_views_per_s"] > 0 and d["session"]["single_view_in_flight_views_per_s"] > 0
    for st_ in d["stages"].values():
        assert st_["algorithmic_bytes"] > 0 and st_["avg_us"] > 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1
    assert d["stage2_path"]["iters_per_s"] > 0
    # round 6: `value` is measured unpinned through the compiled host path; the pinned figure, the Python host path, the cold one-view step and the
    # neighbour kernels (bytes + HBM fractions) ride along; every stage names its bound
    assert d["plugin_api_unpinned_views_per_s"] == d["value"] and d["plugin_api_python_host_path_views_per_s"] > 0 and "compiled" in d["host_path"]
    assert d["single_view_cold"]["single_view_cold_views_per_s"] > 0 and d["single_view_cold"]["ms_cold"] >= 0.9 * d["single_view_cold"]["ms_warm"]
    for st_ in d["stages"].values():
        assert st_["bound"] in ("valu issue", "hbm", "latency (dependent round trips / occupancy)") and 0 < st_["hbm_frac"] < 1
    nb = d["neighbours"]
    for k in ("f1_pack_views", "f2_l1_ssim_loss", "f3_unproject", "f4_corr_volume_pyramid", "f4_convex_upsample"):
        assert nb[k]["forward"]["us"] > 0 and 0 < nb[k]["forward"]["hbm_frac"] < 1 and nb[k]["backward"]["algorithmic_bytes"] > 0, k


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The multi-rank bench path (env rendezvous, per-rank workload, barrier + MAX-over-ranks timing, whole-job aggregate, agreed failure
    handling of the stage-2 leg with its gradient all-reduce) exercised on the 1-GPU box: two ranks share device 0
    (GPSGS_BENCH_SINGLE_DEVICE=1) and talk over gloo instead of RCCL.  No 2/4/8-GPU number exists; this only proves the code path."""
    env = dict(os.environ, GPSGS_BENCH_SINGLE_DEVICE="1", GPSGS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--res", "256", "--gaussians", "30000"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02     # whole-job aggregate: both ranks' views / the slowest rank's time
    assert d["cpu_baseline"] is None                  # rank 0 at N = 1 only
    assert d["stage2_path"]["n_gpus"] == 2 and d["stage2_path"]["iters_per_s"] > 0
    # round 4: the exchange step is also timed on its own, rank 0 says what an N > 1 run does not measure, and every rank was given its own CPU slice
    assert d["stage2_path"]["allreduce_alone_ms"] > 0
    assert set(d["not_measured_at_this_world_size"]) >= {"configs", "full_pipeline", "cpu_baseline", "neighbours"}
    assert d["configs"] is None and d["full_pipeline"] is None and d["neighbours"] is None
    assert isinstance(d["cpu_affinity"], (dict, str))


def test_plain_command_with_gpus_2_starts_two_ranks():
    """The driver's command shape -- `python bench.py --gpus 2 ...`, no torchrun around it -- becomes two ranks by itself (re-exec under
    torch.distributed.run with a 127.0.0.1 rendezvous).  On this 1-GPU box both ranks share device 0 and talk over gloo."""
    env = dict(os.environ, GPSGS_BENCH_SINGLE_DEVICE="1", GPSGS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--res", "256", "--gaussians", "30000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stage2_path"]["n_gpus"] == 2 and d["stage2_path"]["allreduce"].startswith("gloo, world 2")


def test_gpus_2_on_a_one_gpu_node_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPSGS_BENCH_SINGLE_DEVICE")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "exposes 1 GPU" in r.stderr, r.stderr[-1000:]


def test_world_1_through_rccl():
    """GPSGS_DIST_FORCE=1: one rank, but the process group IS initialised with backend nccl (= RCCL) and the stage-2 exchange step (mean
    all-reduce of 5,144,408 gradients, one bucket) IS issued -- RCCL's communicator set-up and its all-reduce kernel run on the MI355X."""
    env = dict(os.environ, GPSGS_DIST_FORCE="1", GPSGS_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--res", "256", "--gaussians", "30000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["stage2_path"]["allreduce"].startswith("nccl, world 1 (forced"), d["stage2_path"]
    assert d["stage2_path"]["allreduce_alone_ms"] > 0 and d["not_measured_at_this_world_size"] == []   # the 20.6 MB RCCL all-reduce on its own


def test_scale_shaped_dry_run_prints_one_record_per_n():
    """tools/scale_dry_run.py: bench.py at N = 1 and 2 back to back with the driver's command shape, one parsed record per N (N = 2 on this 1-GPU box:
    both ranks on device 0 over gloo -- a rehearsal of the multi-rank path, not a measurement)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_dry_run.py"), "--ns", "1,2", "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    recs = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    per_n = [x for x in recs if "n" in x]
    assert [x["n"] for x in per_n] == [1, 2] and recs[-1]["scale_dry_run"] == "ok"
    for x in per_n:
        p = x["parsed"]
        assert x["rc"] == 0 and p["n_gpus"] == x["n"] and p["steps"] == 3 and p["value"] > 0 and p["scaling"] == "weak" and x["consistency"]["fits_in_driver_run"]
        assert p["stage2_path"]["n_gpus"] == x["n"]
