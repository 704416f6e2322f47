#!/bin/bash
# round 4, GPU call G: where does the hooked test_view_interp.py lose host time (config 3 inside a sample: 12.9 vs 16.2 views/s as the reference runs it, although
# the GPU side is faster)?  One feature at a time.  Plus: the full-size oracle test, config-3 trace with the larger two-wave sort grid.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/r04 gpurun_out/prof_r04_config3b
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
OUT=$ROOT/gpurun_out/r04/call_g.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({'within_sample': d['script_run'].get('views_per_s_within_sample'), 'whole': d['script_run']['views_per_s_end_to_end'], 'gpu_side': d.get('views_per_s_gpu_side'), 'gpu_ms': d.get('gpu_ms_per_view')})"; }
timeout 600 python tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > /dev/null 2>&1   # warm: data set, MIOpen
for acc in "" pack corr upsample unproject "corr,upsample,unproject" all; do
  stamp "interp --accelerate '$acc'"
  if [ -z "$acc" ]; then timeout 600 python tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 2>/dev/null | show | tee -a $OUT
  else timeout 600 python tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 --accelerate "$acc" 2>/dev/null | show | tee -a $OUT; fi
done
stamp "full-size view vs the oracle"
timeout 1500 python -m pytest tests/test_gpu_reference.py -x -q -m gpu -s -k "full_size" > gpurun_out/r04/tests_g1.log 2>&1
stamp "exit $?"; grep -E "passed|failed|FAILED|Error|tile_instances" gpurun_out/r04/tests_g1.log | tail -5 | cut -c1-700 | tee -a $OUT
stamp "kernel trace: config 3 again (two-wave sort class on 8,192 workgroups)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_config3b/trace -o t -- python $ROOT/tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > $ROOT/gpurun_out/prof_r04_config3b/run.log 2>&1
cd $ROOT
python - <<'PY' | tee -a $OUT
import csv, glob
for fn in glob.glob('gpurun_out/prof_r04_config3b/trace/*kernel_stats.csv'):
    for r in csv.DictReader(open(fn)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        if n.startswith("k_"): print("%-34s %5s calls %9.1f us avg" % (n[:34], r["Calls"], float(r["AverageNs"])/1e3))
PY
