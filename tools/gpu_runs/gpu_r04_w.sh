#!/bin/bash
# round 4, GPU call W: k_sort_multi<1> at 3 waves per SIMD (168 VGPRs, product) against 4 (128 VGPRs, 17 spilled dwords), large-splat regime
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/call_w.log
: > $OUT
V="gps-gaussian_amd/lib/abl/libgpsgs_hip_GSR_SORT_MULTI_WAVES=4.so"
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10 --fwd-only"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stages_us']['sort'], d['sum_us'])"; }
for rep in 1 2; do
  echo "== product (3 waves)" | tee -a $OUT; timeout 300 python tools/stage_times.py $REG 2>&1 | tail -1 | show | tee -a $OUT
  echo "== 4 waves" | tee -a $OUT; timeout 300 python tools/stage_times.py $REG --lib "$V" 2>&1 | tail -1 | show | tee -a $OUT
done
