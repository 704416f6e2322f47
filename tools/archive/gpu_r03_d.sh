#!/bin/bash
# round 3, GPU call D: the fix of the count / scatter threshold mismatch -- config 3 repro (both families, validated lists), full GPU suite, bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for fam in tiles valu; do
  echo "=== config 3, family=$fam, validated"
  GPSGS_COMPOSITE=$fam GPSGS_TRACE=1 timeout 900 python -X faulthandler tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > gpurun_out/r03/config3_d_$fam.json 2> gpurun_out/r03/config3_d_$fam.err
  echo "exit $?"; grep "gpsgs. header" gpurun_out/r03/config3_d_$fam.err | grep -v "ids out of range 0, out of order 0, non-finite records 0, bins whose scatter cursor missed its end 0" | tail -5
  grep -c "gpsgs. header" gpurun_out/r03/config3_d_$fam.err
done
echo "=== config 3 (timed, no tracing)"
timeout 900 python tools/run_reference.py interp --res 1024 --samples 3 --views 5 --write-images --work /tmp/w3 > gpurun_out/r03/config3.json 2> gpurun_out/r03/config3.err
echo "exit $?"; tail -2 gpurun_out/r03/config3.err; cat gpurun_out/r03/config3.json
cp /tmp/w3/interp_out/*novel00.jpg gpurun_out/r03/ 2>/dev/null
echo "=== full GPU suite"
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03/tests_d.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_d.log | tail -20
echo "=== bench"
timeout 900 python bench.py > gpurun_out/r03/bench_d.json 2> gpurun_out/r03/bench_d.err
echo "bench exit $?"; tail -3 gpurun_out/r03/bench_d.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_d.json') if l.startswith('{')][-1])
for k in ("value","ms_per_step","ms_per_step_iqr","single_view_in_flight_views_per_s","autograd_api_views_per_s","forward_only_views_per_s","configs","stage2_path"):
    print(k, json.dumps(d.get(k)))
print("roofline", json.dumps({k:d["roofline"][k] for k in ("avg_launch_us","launches_averaged","frac","shader_clock_mhz","valu_issue_frac")}))
print("stages", json.dumps({k:v["avg_us"] for k,v in d["stages"].items()}))
PY
