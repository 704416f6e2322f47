#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c5
timeout 300 python tools/stage_times.py > gpurun_out/r2c5/stage_times.log 2>&1
grep family gpurun_out/r2c5/stage_times.log
bash tools/prof2.sh r02b > /dev/null 2>&1
grep "composite\|kernel" gpurun_out/prof_r02b/pmc_summary.md
