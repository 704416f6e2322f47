#!/bin/bash
# round 4, GPU call P: the four-wave register sort and the LDS workgroup sort at full size (1e8 / 2.7e8 instances), lists validated on the device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -s -k "full_size_lists" 2>&1 | tail -8 | cut -c1-400 | tee gpurun_out/r04/call_p.log
