#!/bin/bash
# round 4, GPU call V: the forward compositing kernel at 5 waves per SIMD (92 VGPRs, product) against 6 (capped at 80, 6-7 spilled dwords)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/call_v.log
: > $OUT
V="gps-gaussian_amd/lib/abl/libgpsgs_hip_GSR_FWD_WAVES=6.so"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stages_us']['composite_fwd'], d['sum_us'], d['views_per_s'])"; }
for rep in 1 2; do
for w in "config2:--steps 50" "config2@2048:--steps 30 --render-res 2048" "config5:--steps 20 --res 2048 --gaussians 2400000" "regime:--res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"; do
  name=${w%%:*}; args=${w#*:}
  echo "== $name product (5 waves)" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles $args 2>&1 | tail -1 | show | tee -a $OUT
  echo "== $name 6 waves" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles $args --lib "$V" 2>&1 | tail -1 | show | tee -a $OUT
done
done
