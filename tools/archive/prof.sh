#!/bin/bash
# GPU-box profiling recipe (run through gpurun):  tools/prof.sh <tag>
#  pass 0: kernel trace + stats (timings)        -> gpurun_out/prof_<tag>/trace
#  pass 1: SQ counters, pass 2: FETCH_SIZE, pass 3: WRITE_SIZE (separate --pmc passes, as the guide prescribes; never
#  combined with sys/hip/hsa traces)             -> gpurun_out/prof_<tag>/pmc_{sq,fetch,write}
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan'
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
