#!/bin/bash
# round 4, GPU call U: final state (no-colour backward at 4 waves per SIMD) -- stage times, the whole GPU suite, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_u.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
stamp "stage times: config 2, config 2 at 2048, config 5, untrained-heads regime"
timeout 300 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
timeout 300 python tools/stage_times.py --families tiles --steps 30 --render-res 2048 2>&1 | tail -1 | tee -a $OUT
timeout 300 python tools/stage_times.py --families tiles --steps 20 --res 2048 --gaussians 2400000 2>&1 | tail -1 | tee -a $OUT
timeout 300 python tools/stage_times.py --families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10 2>&1 | tail -1 | tee -a $OUT
rm -f gpurun_out/parity_report.jsonl
stamp "whole GPU suite"
timeout 2700 python -m pytest tests -q -m gpu -s > gpurun_out/r04/tests_u.log 2>&1
stamp "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r04/tests_u.log | tail -12 | cut -c1-600 | tee -a $OUT
stamp "bench"
timeout 1500 python bench.py > gpurun_out/r04/bench_u.json 2> gpurun_out/r04/bench_u.err
stamp "bench exit $?"; python - <<'PY' | tee -a $OUT
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_u.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api"]["views_per_s"], d["autograd_api"]["iqr_views_per_s"], "fwd", d["forward_only_views_per_s"], "deferred", d["deferred_check_views_per_s"], "graph", d["hip_graph_replay"])
print({k:v["avg_us"] for k,v in d["stages"].items()}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"])
for k,v in d["configs"].items():
    print(k[:30], {a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")})
    if "stages_one_view_in_flight" in v: print({a:b["avg_us"] for a,b in v["stages_one_view_in_flight"].items()}, v.get("R"), v.get("longest_bin_list"))
fp=d["full_pipeline"]
for k in ("config4_stage2_accelerated","config4_stage2_as_the_reference_runs_it","config3_view_interp_accelerated","config3_view_interp_as_the_reference_runs_it"):
    v=fp.get(k,{}); print(k, {a:v.get(a) for a in ("stage2_iters_per_s","views_per_s_within_sample","views_per_s_whole_script","views_per_s_gpu_side","wall_s","skipped","error") if v.get(a) is not None})
print("leg wall", fp.get("leg_wall_s"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "stage2_path", d["stage2_path"]["ms_per_iter"])
PY
