#!/bin/bash
# round 4, GPU call J: threshold of the flags-first record gather at the REAL stage-2 render size with trained-like scales (config 2 at 2048^2) and at config 2 / 5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_j.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
for thr in 16 8 4 2 0; do
  stamp "GPSGS_DEBUG_FLAGS_FIRST=$thr : config 2 at 2048, config 2, config 5"
  GPSGS_DEBUG_FLAGS_FIRST=$thr timeout 300 python tools/stage_times.py --families tiles --steps 30 --render-res 2048 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2@2048', d['stages_us']['preprocess_bwd'], d['sum_us'])" | tee -a $OUT
  GPSGS_DEBUG_FLAGS_FIRST=$thr timeout 300 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['stages_us']['preprocess_bwd'], d['sum_us'])" | tee -a $OUT
  GPSGS_DEBUG_FLAGS_FIRST=$thr timeout 300 python tools/stage_times.py --families tiles --steps 20 --res 2048 --gaussians 2400000 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', d['stages_us']['preprocess_bwd'], d['sum_us'])" | tee -a $OUT
done
