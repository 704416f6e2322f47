#!/bin/bash
# round 3, GPU call J: sessions with explicit streams -- session tests + bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_bench_contract.py -x -q -m gpu -k "session or bench or contract or ranks or rccl or gpus" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r03/bench_j.json 2> gpurun_out/r03/bench_j.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_j.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print("s2", json.dumps(d["stage2_gradient_set"])[-230:]); print("stage2_path", d["stage2_path"]["ms_per_iter"], "graph", d["hip_graph_replay"])
print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","shader_clock_mhz","valu_issue_frac")})
print({k:v["avg_us"] for k,v in d["stages"].items()})
PY
