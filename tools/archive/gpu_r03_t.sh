#!/bin/bash
# round 3, GPU call T: final state -- full GPU suite as the driver runs it, per-workgroup timelines without / with wave priority, the round's profiles
# regenerated from HEAD, bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu -s > gpurun_out/r03/tests_t.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_t.log | tail -5
timeout 300 python tools/wg_trace.py --no-priority --out gpurun_out/wg_trace_noprio.npz > gpurun_out/r03/wg_trace_noprio.json 2> gpurun_out/r03/wg_trace_noprio.err
timeout 300 python tools/wg_trace.py --out gpurun_out/wg_trace_prio.npz > gpurun_out/r03/wg_trace_prio.json 2> gpurun_out/r03/wg_trace_prio.err
python - <<'PY'
import json
for n in ("noprio","prio"):
    try:
        d=json.load(open("gpurun_out/r03/wg_trace_%s.json"%n))
        for k in ("fwd","bwd"):
            x=d[k]; print(n,k,"span",x["span_us"],"util",x["simd_utilisation (mean busy / span)"],"dry",x["simd_ran_dry_at_us"],"busy",x["busy_us_per_simd"])
    except Exception as e: print(n,"failed",e)
PY
bash tools/prof_r03.sh r03 > gpurun_out/r03/prof_t.log 2>&1; tail -2 gpurun_out/r03/prof_t.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/r03/bench_t.json 2> gpurun_out/r03/bench_t.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_t.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print("s2", json.dumps(d["stage2_gradient_set"])[-230:]); print("stage2_path", d["stage2_path"]["ms_per_iter"], "graph", d["hip_graph_replay"])
print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","traffic","shader_clock_mhz","valu_issue_frac")})
print({k:v["avg_us"] for k,v in d["stages"].items()})
print(json.dumps({k:{a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_only")} for k,v in d["configs"].items()}))
PY
