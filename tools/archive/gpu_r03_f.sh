#!/bin/bash
# round 3, GPU call F: full GPU suite (parity report of the round) + bench line with the stage-2 gradient set
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r03/tests_f.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_f.log | tail -20
timeout 900 python bench.py > gpurun_out/r03/bench_f.json 2> gpurun_out/r03/bench_f.err
echo "bench exit $?"; tail -3 gpurun_out/r03/bench_f.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_f.json') if l.startswith('{')][-1])
for k in ("value","ms_per_step","ms_per_step_iqr","single_view_in_flight_views_per_s","autograd_api_views_per_s","stage2_gradient_set","forward_only_views_per_s","stage2_path","full_pipeline"):
    print(k, json.dumps(d.get(k))[:600])
print("roofline", json.dumps({k:d["roofline"][k] for k in ("avg_launch_us","launches_averaged","frac","traffic","shader_clock_mhz","valu_issue_frac")}))
print("stages", json.dumps({k:v["avg_us"] for k,v in d["stages"].items()}))
PY
timeout 300 python tools/wg_trace.py > gpurun_out/r03/wg_trace.json 2> gpurun_out/r03/wg_trace.err; tail -2 gpurun_out/r03/wg_trace.err; head -c 5000 gpurun_out/r03/wg_trace.json
