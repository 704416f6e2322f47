#!/bin/bash
# round 4, GPU call B: multi-wave register sort + row-interval binning + flags-first gradient gather: per-stage times in the untrained-heads regime,
# the whole GPU suite (incl. the hook parity test and gradient parity on lists thousands deep), bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_b.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
stamp "stage times, untrained-heads regime (P 550k, 2048^2), fwd+bwd then fwd only"
timeout 600 python tools/stage_times.py --families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10 2>&1 | tail -2 | tee -a $OUT
timeout 600 python tools/stage_times.py --families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10 --fwd-only 2>&1 | tail -1 | tee -a $OUT
stamp "stage times, config 2 (regression check)"
timeout 600 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
stamp "sort paths + large splats"
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -s -k "sort_path or large_splats" > gpurun_out/r04/tests_b1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r04/tests_b1.log | tail -8 | tee -a $OUT
rm -f gpurun_out/parity_report.jsonl
stamp "whole GPU suite"
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r04/tests_b.log 2>&1
stamp "tests exit $?"; grep -E "passed|failed|FAILED|ERROR|worst_relative" gpurun_out/r04/tests_b.log | tail -12 | tee -a $OUT
stamp "bench"
timeout 1200 python bench.py > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
stamp "bench exit $?"; python - <<'PY' | tee -a $OUT
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_b.json') if l.startswith('{')][-1])
print("value", d["value"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print({k:v["avg_us"] for k,v in d["stages"].items()})
for k,v in d["configs"].items():
    print(k[:40], json.dumps({a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")}) if "fwd_bwd" in v else v)
    if "stages_one_view_in_flight" in v: print(json.dumps(v["stages_one_view_in_flight"]), v.get("R"), v.get("longest_bin_list"))
PY
