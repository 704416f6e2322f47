#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c12
rm -f gpurun_out/parity_report.jsonl
(timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r2c12/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c12/pytest.log)
tail -8 gpurun_out/r2c12/pytest.log
grep needles gpurun_out/parity_report.jsonl | cut -c1-400
timeout 200 python tools/stage_times.py --families valu 2>&1 | grep family
