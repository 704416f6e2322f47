#!/bin/bash
# round 4, GPU call F: the round's profile evidence (config 2 trace + counters; the untrained-heads regime; configs 3 and 4 with the reference's scripts),
# the new parity tests, the bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/r04 gpurun_out/prof_r04_regime2 gpurun_out/prof_r04_config3 gpurun_out/prof_r04_config4b
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_f.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
stamp "new tests: full-size view from the real networks vs the oracle; unproject entry points; hook"
MIOPEN_FIND_MODE=FAST timeout 1500 python -m pytest tests/test_gpu_reference.py tests/test_gpu_unproject.py -x -q -m gpu -s -k "full_size or entry_points or import_hook" > gpurun_out/r04/tests_f1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error|tile_instances|worst_relative" gpurun_out/r04/tests_f1.log | tail -8 | cut -c1-700 | tee -a $OUT
stamp "config 2: trace + counters (tools/prof_r04.sh)"
bash tools/prof_r04.sh r04 > gpurun_out/r04/prof_r04.log 2>&1
stamp "exit $?"; tail -5 gpurun_out/r04/prof_r04.log | cut -c1-400 | tee -a $OUT
export MIOPEN_FIND_MODE=FAST
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"
cd /tmp
stamp "kernel trace: untrained-heads regime, one view at a time"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_regime2/trace -o t -- python $ROOT/tools/stage_times.py $REG > $ROOT/gpurun_out/prof_r04_regime2/trace.log 2>&1
stamp "kernel trace: config 3 (test_view_interp.py, as the reference runs it)"
timeout 900 python $ROOT/tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > /dev/null 2>&1   # warm: data set, MIOpen
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_config3/trace -o t -- python $ROOT/tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > $ROOT/gpurun_out/prof_r04_config3/run.log 2>&1
stamp "kernel trace: config 4 (train_stage2.Trainer, GPSGS_ACCELERATE=all)"
timeout 900 python $ROOT/tools/run_reference.py ddp --res 1024 --steps 6 --batch 4 --train-samples 4 --work /tmp/w4 --accelerate all > /dev/null 2>&1   # warm
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_config4b/trace -o t -- python $ROOT/tools/run_reference.py ddp --res 1024 --steps 8 --batch 4 --train-samples 4 --work /tmp/w4 --accelerate all > $ROOT/gpurun_out/prof_r04_config4b/run.log 2>&1
cd $ROOT
unset MIOPEN_FIND_MODE
stamp "bench"
timeout 1500 python bench.py > gpurun_out/r04/bench_f.json 2> gpurun_out/r04/bench_f.err
stamp "bench exit $?"; python - <<'PY' | tee -a $OUT
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_f.json') if l.startswith('{')][-1])
print("value", d["value"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api"]["views_per_s"], d["autograd_api"]["iqr_views_per_s"], "fwd", d["forward_only_views_per_s"], "deferred", d["deferred_check_views_per_s"], "graph", d["hip_graph_replay"])
print({k:v["avg_us"] for k,v in d["stages"].items()}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
for k,v in d["configs"].items():
    if "stages_one_view_in_flight" in v: print({a:b["avg_us"] for a,b in v["stages_one_view_in_flight"].items()}, v.get("R"), v.get("longest_bin_list"), {a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")})
fp=d["full_pipeline"]
for k in ("config4_stage2_accelerated","config4_stage2_as_the_reference_runs_it","config3_view_interp_accelerated","config3_view_interp_as_the_reference_runs_it"):
    v=fp.get(k,{}); print(k, {a:v.get(a) for a in ("stage2_iters_per_s","views_per_s_within_sample","views_per_s_whole_script","views_per_s_gpu_side","wall_s","skipped","error")})
print("stage2_path", d["stage2_path"])
PY
