#!/bin/bash
# round 4, GPU call Q: bench contract tests (N > 1 fields over gloo on one GPU, RCCL at world 1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -8 | cut -c1-600 | tee gpurun_out/r04/call_q.log
