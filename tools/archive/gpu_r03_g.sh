#!/bin/bash
# round 3, GPU call G: smoke, full GPU suite, final bench line, sampler drop-in host cost
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu -s > gpurun_out/r03/tests_g.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_g.log | tail -10
timeout 900 python bench.py > gpurun_out/r03/bench_g.json 2> gpurun_out/r03/bench_g.err
echo "bench exit $?"; tail -3 gpurun_out/r03/bench_g.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_g.json') if l.startswith('{')][-1])
for k in ("metric","value","ms_per_step","ms_per_step_iqr","single_view_in_flight_views_per_s","autograd_api_views_per_s","stage2_gradient_set","forward_only_views_per_s","deferred_check_views_per_s","stage2_path","configs","hip_graph_replay","cpu_baseline","cpu_taichi_splat_port"):
    print(k, json.dumps(d.get(k))[:900])
print("roofline", json.dumps(d["roofline"])[:1500])
print("stages", json.dumps({k:v["avg_us"] for k,v in d["stages"].items()}))
PY
for dt in fp32 fp16; do timeout 300 python tools/bench_cs_dropin.py 2 $dt 2>/dev/null | tail -1; done
