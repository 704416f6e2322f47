#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c9
rm -f gpurun_out/parity_report.jsonl
(timeout 2700 python -m pytest tests -m gpu -q -x > gpurun_out/r2c9/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c9/pytest.log)
tail -15 gpurun_out/r2c9/pytest.log
