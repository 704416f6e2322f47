#!/bin/bash
# round 4, GPU call S: per-workgroup timeline of the compositing kernels in the large-splat regime and at config 2 rendered at 2048^2 (is the time balance or throughput?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/call_s.log
: > $OUT
echo "== regime, wave priority on" | tee -a $OUT
timeout 600 python tools/wg_trace.py --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --out gpurun_out/r04/wg_regime.npz 2>&1 | grep -v amdgpu.ids | tee -a $OUT
echo "== regime, wave priority off" | tee -a $OUT
timeout 600 python tools/wg_trace.py --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --no-priority --out gpurun_out/r04/wg_regime_noprio.npz 2>&1 | grep -v amdgpu.ids | tee -a $OUT
echo "== config 2 rendered at 2048^2" | tee -a $OUT
timeout 600 python tools/wg_trace.py --render-res 2048 --out gpurun_out/r04/wg_c2hr.npz 2>&1 | grep -v amdgpu.ids | tee -a $OUT
