#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c15
(timeout 1500 python -m pytest tests/test_gpu_raster.py -m gpu -q -k "selftest or known_answers or parity or fuzz or clamp or config2" > gpurun_out/r2c15/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c15/pytest.log)
grep "passed\|failed\|^FAILED" gpurun_out/r2c15/pytest.log | tail -5
timeout 300 python tools/stage_times.py --families tiles,valu 2>&1 | grep family | cut -c1-330
