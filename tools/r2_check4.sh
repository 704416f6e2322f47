#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c4
(timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -k "selftest or known_answers or parity or fuzz or config2" > gpurun_out/r2c4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c4/pytest.log)
tail -8 gpurun_out/r2c4/pytest.log
timeout 300 python tools/stage_times.py > gpurun_out/r2c4/stage_times.log 2>&1
grep family gpurun_out/r2c4/stage_times.log
