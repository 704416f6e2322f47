#!/bin/bash
# round 4, GPU call L: where the host time of the drop-in autograd path goes (the plugin-API number read 2,891 views/s on one box and 3,590 - 3,668 on four others)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/call_l.log
: > $OUT
lscpu | grep -E "Model name|MHz|^CPU\(s\)" | tee -a $OUT
timeout 600 python tools/host_cprofile.py 2>&1 | tee -a $OUT
timeout 600 python tools/host_cprofile.py --check deferred 2>&1 | head -3 | tee -a $OUT
timeout 300 python tools/host_profile.py 2>&1 | tail -1 | tee -a $OUT
