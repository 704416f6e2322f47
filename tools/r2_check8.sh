#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c8
rm -f gpurun_out/parity_report.jsonl
(timeout 2400 python -m pytest tests/test_gpu_raster.py -m gpu -q > gpurun_out/r2c8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c8/pytest.log)
tail -12 gpurun_out/r2c8/pytest.log
wc -l gpurun_out/parity_report.jsonl
