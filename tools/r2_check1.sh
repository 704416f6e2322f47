#!/bin/bash
# first GPU call of round 2: matrix-core compositing kernels -- selftest, parity subset, stage times of both families
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c1
(timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -s -k "selftest or known_answers or parity or fuzz" > gpurun_out/r2c1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c1/pytest.log)
tail -25 gpurun_out/r2c1/pytest.log
timeout 300 python tools/stage_times.py > gpurun_out/r2c1/stage_times.log 2>&1
cat gpurun_out/r2c1/stage_times.log | tail -5
