#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c13
(timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -s -k "needle" > gpurun_out/r2c13/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c13/pytest.log)
grep "needles\|passed\|failed\|Error" gpurun_out/r2c13/pytest.log | tail
