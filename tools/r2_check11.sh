#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c11
(timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_pack.py tests/test_gpu_bench_contract.py -m gpu -q -x -k "deferred or pack or pts2render or bench or session" > gpurun_out/r2c11/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c11/pytest.log)
tail -8 gpurun_out/r2c11/pytest.log
timeout 600 python bench.py --steps 24 --warmup 5 --no-cpu-baseline > gpurun_out/r2c11/bench.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r2c11/bench.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'repeats_ms_per_step', 'single_view_in_flight_views_per_s', 'autograd_api_views_per_s')}); print(d['stage2_path'])
PY
