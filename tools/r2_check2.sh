#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c2
(timeout 1500 python -m pytest tests/test_gpu_raster.py -m gpu -x -q > gpurun_out/r2c2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c2/pytest.log)
tail -15 gpurun_out/r2c2/pytest.log
bash tools/prof2.sh r02a
cat gpurun_out/prof_r02a/pmc_summary.md | grep "composite\|kernel"
