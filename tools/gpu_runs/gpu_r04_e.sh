#!/bin/bash
# round 4, GPU call E: compare-exchanges as v_min_f64 / v_max_f64, one-wave sort class on 16,384 workgroups: stage times (both regimes), sort tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_e.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"
stamp "stage times, untrained-heads regime, fwd+bwd"
timeout 600 python tools/stage_times.py $REG 2>&1 | tail -1 | tee -a $OUT
for g in 8192 32768; do
  stamp "one-wave sort grid $g (forward only)"
  GPSGS_DEBUG_SORT_GRID=$g timeout 300 python tools/stage_times.py $REG --fwd-only 2>&1 | tail -1 | tee -a $OUT
done
stamp "stage times, config 2"
timeout 600 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
timeout 600 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
stamp "sort paths + large splats + stress"
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "sort_path or large_splats or scan or overflow or stream" > gpurun_out/r04/tests_e1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/r04/tests_e1.log | tail -5 | cut -c1-400 | tee -a $OUT
