#!/bin/bash
# round 4, GPU call A: the opt-in import hook (GPSGS_ACCELERATE) under the reference's unmodified train_stage2.py -- parity test at 256^2, then BASELINE
# config 4 at full size (batch 4, 1024^2 -> 2048^2) plain vs accelerated on the same box, MIOpen find-mode timing, rocprofv3 kernel stats of the accelerated run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/r04 gpurun_out/prof_r04_config4
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_a.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
stamp "hook parity test (256^2)"
timeout 900 python -m pytest tests/test_gpu_reference.py -x -q -m gpu -s -k "import_hook" > gpurun_out/r04/hook_test.log 2>&1
stamp "exit $?"; grep -E "worst_relative|passed|failed|Error|assert" gpurun_out/r04/hook_test.log | tail -12 | tee -a $OUT
# config 4, MIOpen FAST find mode (no exhaustive search on a fresh box), plain then accelerated
export MIOPEN_FIND_MODE=FAST
stamp "config 4 plain, MIOPEN_FIND_MODE=FAST"
timeout 1200 python tools/run_reference.py ddp --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r04/config4_plain_fast.json 2> gpurun_out/r04/config4_plain_fast.err
stamp "exit $?"; tail -c 1500 gpurun_out/r04/config4_plain_fast.json | tee -a $OUT
stamp "config 4 accelerated, MIOPEN_FIND_MODE=FAST"
timeout 1200 python tools/run_reference.py ddp --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 --accelerate all > gpurun_out/r04/config4_accel_fast.json 2> gpurun_out/r04/config4_accel_fast.err
stamp "exit $?"; tail -c 1800 gpurun_out/r04/config4_accel_fast.json | tee -a $OUT
stamp "rocprofv3 kernel stats of the accelerated run"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_config4 -o t -- python $ROOT/tools/run_reference.py ddp --res 1024 --steps 8 --batch 4 --train-samples 4 --work /tmp/w4 --accelerate all > $ROOT/gpurun_out/prof_r04_config4/run.log 2>&1
cd $ROOT
stamp "exit $?"
python - <<'PY' | tee -a $OUT
import csv, glob
for fn in glob.glob('gpurun_out/prof_r04_config4/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(fn)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(fn, "kernels", len(rows), "total ms", round(tot/1e6,1))
    for r in rows[:40]:
        print("%-70s %6s calls %9.1f us avg %6.2f %%" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
unset MIOPEN_FIND_MODE
stamp "config 4 plain, default find mode (fresh user db): how long does the find take, and is the iteration faster?"
timeout 1500 python tools/run_reference.py ddp --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 > gpurun_out/r04/config4_plain_default.json 2> gpurun_out/r04/config4_plain_default.err
stamp "exit $?"; tail -c 1500 gpurun_out/r04/config4_plain_default.json | tee -a $OUT
stamp "config 4 accelerated, default find mode (db warm)"
timeout 1200 python tools/run_reference.py ddp --res 1024 --steps 16 --batch 4 --train-samples 4 --work /tmp/w4 --accelerate all > gpurun_out/r04/config4_accel_default.json 2> gpurun_out/r04/config4_accel_default.err
stamp "exit $?"; tail -c 1800 gpurun_out/r04/config4_accel_default.json | tee -a $OUT
