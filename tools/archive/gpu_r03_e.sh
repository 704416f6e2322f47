#!/bin/bash
# round 3, GPU call E: profiles (kernel stats, PMC, traffic) + the full pipeline at BASELINE sizes (configs 3 and 4)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
bash tools/prof_r03.sh r03 > gpurun_out/r03/prof.log 2>&1; tail -5 gpurun_out/r03/prof.log | cut -c1-600
echo "=== config 3 (timed)"
timeout 900 python tools/run_reference.py interp --res 1024 --samples 3 --views 5 --work /tmp/w3 > gpurun_out/r03/config3.json 2> gpurun_out/r03/config3.err
echo "exit $?"; tail -2 gpurun_out/r03/config3.err | cut -c1-300; cat gpurun_out/r03/config3.json
echo "=== config 4 (launcher, world 1, batch 4, real networks)"
timeout 1500 python tools/run_reference.py ddp --res 1024 --steps 24 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r03/config4.json 2> gpurun_out/r03/config4.err
echo "exit $?"; tail -2 gpurun_out/r03/config4.err | cut -c1-300; cat gpurun_out/r03/config4.json
echo "=== config 4 stand-alone script (train_stage2.py as __main__, batch 4)"
timeout 900 python tools/run_reference.py train --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r03/config4_script.json 2> gpurun_out/r03/config4_script.err
echo "exit $?"; cat gpurun_out/r03/config4_script.json
timeout 600 python -m pytest tests/test_gpu_raster.py -q -m gpu -k "large_splats" 2>&1 | tail -3
du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null
