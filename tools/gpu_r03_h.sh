#!/bin/bash
# round 3, GPU call H: stall probe (config 2 at 2048^2, forward only), C++ host with row ranges, bench configs with three blocks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for n in 1 3; do timeout 300 python tools/stall_probe.py --inflight $n --steps 6000 2>/dev/null | tail -1; done
timeout 300 python tools/stall_probe.py --inflight 3 --steps 3000 --kind train 2>/dev/null | tail -1
timeout 600 python -m pytest tests/test_gpu_capi_host.py tests/test_gpu_pack.py -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_h.json 2> gpurun_out/r03/bench_h.err
echo "bench exit $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_h.json') if l.startswith('{')][-1])
print("value", d["value"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "s2", json.dumps(d["stage2_gradient_set"])[-200:])
print(json.dumps(d["configs"])[:2500])
PY
