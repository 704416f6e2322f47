#!/bin/bash
# round 4, GPU call D: banded table for the one workgroup that straddles the two source views (the straggler behind k_preprocess / k_scatter at R = 3e7),
# sort grids sized to one resident generation; kernel trace + SQ counters of the untrained-heads regime; the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/r04 gpurun_out/prof_r04_regime
export TMPDIR=/tmp
OUT=gpurun_out/r04/call_d.log
: > $OUT
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT; }
REG="--families tiles --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --steps 10"
stamp "stage times, untrained-heads regime, fwd+bwd"
timeout 600 python tools/stage_times.py $REG 2>&1 | tail -1 | tee -a $OUT
for g in 1024 2048 4096 6144; do
  stamp "one-wave sort grid $g (forward only)"
  GPSGS_DEBUG_SORT_GRID=$g timeout 300 python tools/stage_times.py $REG --fwd-only 2>&1 | tail -1 | tee -a $OUT
done
stamp "kernel trace of the regime (fwd+bwd)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r04_regime/trace -o t -- python $ROOT/tools/stage_times.py $REG > $ROOT/gpurun_out/prof_r04_regime/trace.log 2>&1
stamp "SQ counters of the regime"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "k_sort|k_preprocess|k_scatter|k_composite" -f csv -d $ROOT/gpurun_out/prof_r04_regime/pmc_sq -o p -- python $ROOT/tools/stage_times.py $REG > $ROOT/gpurun_out/prof_r04_regime/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-include-regex "k_sort|k_preprocess|k_scatter" -f csv -d $ROOT/gpurun_out/prof_r04_regime/pmc_sq2 -o p -- python $ROOT/tools/stage_times.py $REG --fwd-only > $ROOT/gpurun_out/prof_r04_regime/pmc_sq2.log 2>&1
cd $ROOT
python - <<'PY' | tee -a $OUT
import csv, glob, collections
for fn in glob.glob('gpurun_out/prof_r04_regime/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        if n.startswith("k_"): print("%-40s %5s calls %9.1f us avg (min %.1f max %.1f)" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
for sub in ("pmc_sq","pmc_sq2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob('gpurun_out/prof_r04_regime/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(k, {c: "%.4g" % (sum(x)/len(x)) for c,x in sorted(v.items())})
PY
stamp "whole GPU suite"
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r04/tests_d.log 2>&1
stamp "tests exit $?"; grep -E "passed|failed|FAILED|ERROR" gpurun_out/r04/tests_d.log | tail -12 | cut -c1-600 | tee -a $OUT
