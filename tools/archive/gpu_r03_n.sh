#!/bin/bash
# round 3, GPU call N: sensitivity probes of the compositing kernels (extra VALU / SALU / transcendental work per pair, results unchanged) and the
# parity failure of call M looked at per variant
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
L=gps-gaussian_amd/lib
for v in $VARIANTS; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v"; timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"
done
for v in $PARITY; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== parity on $v"
  timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_pack.py tests/test_gpu_capi_host.py -q -m gpu 2>&1 | grep -E "^E  |passed|failed|FAILED" | head -40
done
