#!/bin/bash
# round 4, GPU call M: host time of the drop-in autograd path after trimming the backward's Python (one split instead of six slices, raw stream handle)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/call_m.log
: > $OUT
lscpu | grep -E "Model name|max MHz" | head -2 | tee -a $OUT
timeout 600 python tools/host_cprofile.py 2>&1 | head -12 | tee -a $OUT
timeout 300 python tools/host_profile.py 2>&1 | tail -1 | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py -x -q -m gpu -k "graph or loss_scale or arena or batch or stale or reproduc or stream" 2>&1 | tail -2 | tee -a $OUT
