#!/bin/bash
# GPU box: one-view-at-a-time stage times + HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes, no trace domains) of the
# rasteriser kernels over tools/stage_times.py.  tools/prof_traffic.sh <tag> [stage_times args]
TAG=${1:-traffic}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/stage_times.py --families tiles --steps 10 $*"
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan'
$CMD > $OUT/stage_times.json 2> $OUT/stage_times.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT $OUT/pmc_summary.md > /dev/null 2>&1
cat $OUT/stage_times.json; cat $OUT/pmc_summary.md
