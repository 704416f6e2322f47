#!/bin/bash
# round 3, GPU call I: final verification -- full GPU suite (-x, as the driver runs it), bench line, config 4 with the exchange step issued through RCCL
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu -s > gpurun_out/r03/tests_i.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_i.log | tail -10
timeout 900 python bench.py > gpurun_out/r03/bench_i.json 2> gpurun_out/r03/bench_i.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_i.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print("s2", json.dumps(d["stage2_gradient_set"])[-230:]); print("stage2_path", d["stage2_path"]["ms_per_iter"])
print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","shader_clock_mhz","valu_issue_frac")})
print(json.dumps({k:{a:{b:(x["views_per_s"], x["blocks_ms_per_view"]) for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")} for k,v in d["configs"].items()}))
PY
echo "=== config 4 with RCCL forced at world 1"
GPSGS_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout 1500 python tools/run_reference.py ddp --res 1024 --steps 24 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r03/config4.json 2> gpurun_out/r03/config4.err
echo "exit $?"; tail -2 gpurun_out/r03/config4.err | cut -c1-300; cat gpurun_out/r03/config4.json
