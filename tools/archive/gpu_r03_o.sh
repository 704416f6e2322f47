#!/bin/bash
# round 3, GPU call O: how the compositing kernels scale with the waves resident per SIMD (GPSGS_DEBUG_LDS_PAD), wave priority by remaining work
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
L=gps-gaussian_amd/lib
OUT=gpurun_out/r03/call_o.log
: > $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"; }
cp $L/variants/g0.so $L/libgpsgs_hip.so
for pad in 0 8000 11500 18000 38000 6000 3000; do
  echo "== g0 pad $pad" | tee -a $OUT; GPSGS_DEBUG_LDS_PAD=$pad timeout 300 python tools/stage_times.py --families tiles --steps 60 2>&1 | tail -1 | show | tee -a $OUT
done
for v in $VARIANTS; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | show | tee -a $OUT
done
for v in $PARITY; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== parity on $v" | tee -a $OUT
  timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_pack.py tests/test_gpu_capi_host.py -q -m gpu 2>&1 | grep -E "^E  .*Error|passed|failed|FAILED" | cut -c1-200 | head -20 | tee -a $OUT
done
