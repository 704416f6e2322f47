#!/bin/bash
# round 3, GPU call U: one-workgroup scan against the fused multi-workgroup form (stage times), full GPU suite (incl. the scan equality test and
# test_real_data.py as __main__), bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
OUT=gpurun_out/r03/call_u.log
: > $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('scan %.2f fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['scan'], s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"; }
for form in multi one multi one; do
  echo "== GPSGS_SCAN=$form" | tee -a $OUT; GPSGS_SCAN=$form timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | show | tee -a $OUT
done
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r03/tests_u.log 2>&1
echo "tests exit $?" | tee -a $OUT; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_u.log | tail -8 | tee -a $OUT
timeout 900 python bench.py > gpurun_out/r03/bench_u2.json 2> gpurun_out/r03/bench_u2.err
echo "bench exit $?"; python - <<'PY' | tee -a $OUT
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_u2.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print("s2", json.dumps(d["stage2_gradient_set"])[-230:]); print("stage2_path", d["stage2_path"]["ms_per_iter"], "graph", d["hip_graph_replay"])
print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","traffic","shader_clock_mhz","valu_issue_frac")})
print({k:v["avg_us"] for k,v in d["stages"].items()})
print(json.dumps({k:{a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")} for k,v in d["configs"].items()}))
PY
