#!/bin/bash
# round 3, GPU call B: locate the config-3 memory fault, full GPU suite with the hardened / new tests, bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
echo "=== repro 1: faulthandler only"
timeout 600 python -X faulthandler tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > gpurun_out/r03/config3_b1.json 2> gpurun_out/r03/config3_b1.err
echo "exit $?"; grep -v Warning gpurun_out/r03/config3_b1.err | tail -25; cat gpurun_out/r03/config3_b1.json
echo "=== repro 2: GPSGS_TRACE + launch blocking"
GPSGS_TRACE=1 HIP_LAUNCH_BLOCKING=1 timeout 900 python -X faulthandler tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > gpurun_out/r03/config3_b2.json 2> gpurun_out/r03/config3_b2.err
echo "exit $?"; grep -v Warning gpurun_out/r03/config3_b2.err | tail -30
echo "=== full GPU suite"
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r03/tests_b.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r03/tests_b.log | tail -30
echo "=== bench"
timeout 600 python bench.py > gpurun_out/r03/bench_b.json 2> gpurun_out/r03/bench_b.err
echo "bench exit $?"; tail -2 gpurun_out/r03/bench_b.err; head -c 3000 gpurun_out/r03/bench_b.json
du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null
