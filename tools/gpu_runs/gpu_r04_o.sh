#!/bin/bash
# round 4, GPU call O: the row-interval binning lists a superset of the per-cell test (new test)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -s -k "row_interval" 2>&1 | tail -6 | cut -c1-400 | tee gpurun_out/r04/call_o.log
