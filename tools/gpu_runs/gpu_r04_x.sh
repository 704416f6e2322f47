#!/bin/bash
# round 4, GPU call X: bench with the stage-2 gradient set in the configs legs (no full-pipeline / CPU legs: a quick check)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 900 python bench.py --no-full-pipeline --no-cpu-baseline > gpurun_out/r04/bench_x.json 2> gpurun_out/r04/bench_x.err
echo "exit $?" | tee gpurun_out/r04/call_x.log
python - <<'PY' | tee -a gpurun_out/r04/call_x.log
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_x.json') if l.startswith('{')][-1])
print("value", d["value"], "api", d["autograd_api"]["views_per_s"], "s2", d["stage2_gradient_set"])
for k,v in d["configs"].items():
    print(k[:34], {a:{b:x["views_per_s"] for b,x in v[a].items()} for a in ("fwd_bwd","fwd_bwd_stage2_gradient_set","fwd_only")} if "fwd_bwd" in v else v)
PY
