#!/bin/bash
# round 3, GPU call P: compositing kernels with more waves resident per SIMD (B operands of the exponent MFMAs from a device table, the backward's
# dL/dpixel rows in LDS): variants, a dense scene (config 5), parity
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
L=gps-gaussian_amd/lib
OUT=gpurun_out/r03/call_p.log
: > $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"; }
for v in $VARIANTS; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | show | tee -a $OUT
done
for v in $DENSE; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v config 5 (2048^2, 2.4 M)" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles --res 2048 --gaussians 2400000 --steps 20 2>&1 | tail -1 | show | tee -a $OUT
done
for v in $PARITY; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== parity on $v" | tee -a $OUT
  timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_pack.py tests/test_gpu_capi_host.py -q -m gpu 2>&1 | grep -E "^E  .*Error|passed|failed|FAILED" | cut -c1-200 | head -20 | tee -a $OUT
done
