#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c14
rm -f gpurun_out/parity_report.jsonl
(timeout 1500 python -m pytest tests/test_gpu_raster.py -m gpu -q -s -k "selftest or known_answers or parity or fuzz or clamp" > gpurun_out/r2c14/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c14/pytest.log)
grep "selftest:\|passed\|failed\|^FAILED" gpurun_out/r2c14/pytest.log | tail -12
timeout 300 python tools/stage_times.py --families valu,tiles 2>&1 | grep family
for fam in valu tiles; do GPSGS_COMPOSITE=$fam timeout 300 python tools/pipeline2.py 6 2>&1 | grep views_in | tail -2; done
