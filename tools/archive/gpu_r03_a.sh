#!/bin/bash
# round 3, GPU call A: the reference's own code on the HIP drop-in (tests + configs 3 / 4 at full size), the new bench.py launch tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
ls oracle/_ref/GPS-Gaussian | head -3
timeout 2400 python -m pytest tests/test_gpu_reference.py tests/test_gpu_bench_contract.py -x -q -m gpu -s > gpurun_out/r03/tests_a.log 2>&1
echo "tests exit $?"; tail -15 gpurun_out/r03/tests_a.log
timeout 900 python tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > gpurun_out/r03/config3.json 2> gpurun_out/r03/config3.err
echo "config3 exit $?"; tail -3 gpurun_out/r03/config3.err; cat gpurun_out/r03/config3.json
timeout 1200 python tools/run_reference.py ddp --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r03/config4.json 2> gpurun_out/r03/config4.err
echo "config4 exit $?"; tail -3 gpurun_out/r03/config4.err; cat gpurun_out/r03/config4.json
