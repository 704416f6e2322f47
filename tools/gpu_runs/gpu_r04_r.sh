#!/bin/bash
# round 4, GPU call R: the whole GPU suite at HEAD (all new tests together), smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_r.log
: > $OUT
rm -f gpurun_out/parity_report.jsonl
timeout 2700 python -m pytest tests -q -m gpu -s > gpurun_out/r04/tests_r.log 2>&1
echo "tests exit $?" | tee -a $OUT; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r04/tests_r.log | tail -12 | cut -c1-600 | tee -a $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT
