#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c6
(timeout 1500 python -m pytest tests/test_gpu_raster.py -m gpu -q > gpurun_out/r2c6/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c6/pytest.log)
tail -8 gpurun_out/r2c6/pytest.log
timeout 300 python tools/stage_times.py --families valu > gpurun_out/r2c6/stage_times.log 2>&1
grep family gpurun_out/r2c6/stage_times.log
