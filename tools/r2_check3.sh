#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c3
timeout 600 python tools/diag_grad.py > gpurun_out/r2c3/diag_grad.log 2>&1
tail -3 gpurun_out/r2c3/diag_grad.log | cut -c1-3000
for pad in 0 20000 40000 80000; do
  echo "== LDS pad $pad" >> gpurun_out/r2c3/occ.log
  GPSGS_DEBUG_LDS_PAD=$pad timeout 200 python tools/stage_times.py --steps 15 2>&1 | grep family | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['family'], d['stages_us']['composite_fwd'], d['stages_us']['composite_bwd'])" >> gpurun_out/r2c3/occ.log
done
cat gpurun_out/r2c3/occ.log
