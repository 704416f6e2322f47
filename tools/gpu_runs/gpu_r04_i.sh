#!/bin/bash
# round 4, GPU call I: k_preprocess_bwd reads its flag runs cooperatively through LDS; launcher-with-eval and hooked-interp tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_i.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"
stamp "stage times, untrained-heads regime, fwd+bwd (twice)"
timeout 600 python tools/stage_times.py $REG 2>&1 | tail -1 | tee -a $OUT
timeout 600 python tools/stage_times.py $REG 2>&1 | tail -1 | tee -a $OUT
stamp "stage times, config 2; config 2 at 2048"
timeout 600 python tools/stage_times.py --families tiles --steps 50 2>&1 | tail -1 | tee -a $OUT
timeout 600 python tools/stage_times.py --families tiles --steps 30 --render-res 2048 2>&1 | tail -1 | tee -a $OUT
stamp "gradient tests (large splats, fuzz, inputs, stale records) + launcher with eval + hooked interp"
MIOPEN_FIND_MODE=FAST timeout 2000 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_reference.py -x -q -m gpu -s -k "large_splats or fuzz or stale or clamp or config2 or config5 or launcher_world_1 or import_hook_writes" > gpurun_out/r04/tests_i1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error|^\{" gpurun_out/r04/tests_i1.log | tail -8 | cut -c1-500 | tee -a $OUT
