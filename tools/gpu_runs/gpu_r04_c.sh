#!/bin/bash
# round 4, GPU call C: f64-compare sort + aligned flag reads -- stage times in the untrained-heads regime, count-pass probes (what bounds k_preprocess with
# ~100-cell rects), sort / large-splat / hook tests, bench with the measured full-pipeline leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_c.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"
stamp "stage times, untrained-heads regime, fwd+bwd"
timeout 600 python tools/stage_times.py $REG 2>&1 | tail -1 | tee -a $OUT
for v in GSR_ABL_COUNT_NO_GLOBAL_ATOMIC GSR_ABL_COUNT_NO_LDS_ATOMIC GSR_ABL_COUNT_NO_LOOP; do
  stamp "probe $v (forward only; R = 0 by construction: only the preprocess time means anything)"
  timeout 300 python tools/stage_times.py $REG --fwd-only --lib gps-gaussian_amd/lib/abl/libgpsgs_hip_$v.so 2>&1 | tail -1 | tee -a $OUT
done
stamp "stage times, config 2"
timeout 600 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
stamp "sort paths + large splats + hook test"
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_reference.py -x -q -m gpu -s -k "sort_path or large_splats or import_hook" > gpurun_out/r04/tests_c1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error|worst_relative" gpurun_out/r04/tests_c1.log | tail -8 | cut -c1-900 | tee -a $OUT
stamp "bench (with the full-pipeline leg)"
timeout 1500 python bench.py > gpurun_out/r04/bench_c.json 2> gpurun_out/r04/bench_c.err
stamp "bench exit $?"; python - <<'PY' | tee -a $OUT
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_c.json') if l.startswith('{')][-1])
print("value", d["value"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api"], "fwd", d["forward_only_views_per_s"], "deferred", d["deferred_check_views_per_s"], "graph", d["hip_graph_replay"])
print({k:v["avg_us"] for k,v in d["stages"].items()})
for k,v in d["configs"].items():
    if "stages_one_view_in_flight" in v: print({a:b["avg_us"] for a,b in v["stages_one_view_in_flight"].items()}, v.get("R"), v.get("longest_bin_list"), {a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")})
print(json.dumps(d["full_pipeline"])[:3000])
PY
