#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c7
(timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -k "session or boundary or overflow or graph" > gpurun_out/r2c7/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c7/pytest.log)
tail -5 gpurun_out/r2c7/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c7/bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r2c7/bench.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'repeats_ms_per_step', 'autograd_api_views_per_s', 'forward_only_views_per_s', 'deferred_check_views_per_s', 'hip_graph_replay')})
        print(d['roofline']); print({k: v['avg_us'] for k, v in d['stages'].items()}); print(d['stage2_path'])
PY
tail -3 gpurun_out/r2c7/bench.log | cut -c1-300
timeout 100 python tools/host_profile.py 2>&1 | grep -v amdgpu
