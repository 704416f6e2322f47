#!/bin/bash
# GPU box: rocprofv3 passes over a short rasteriser run (tools/stage_times.py), kernel trace first, then SQ counter sets in
# their own passes (never combined with trace domains).  tools/prof2.sh <tag> [stage_times args]
TAG=${1:-r02}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/stage_times.py --steps 10 $*"
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan'
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc ${PMC2:-SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD} --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1
if [ -n "$PROF_TRAFFIC" ]; then
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
fi
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
cd $ROOT
python tools/pmc_summary.py $OUT $OUT/pmc_summary.md > /dev/null 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
tail -3 $OUT/pmc_sq2.log
