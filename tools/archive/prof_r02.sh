#!/bin/bash
# GPU box (through gpurun): the round's profile evidence, all from the SAME command the driver runs (bench.py), in separate passes:
#   pass 0  rocprofv3 --kernel-trace --stats (default = 6 views in flight, and --inflight 1)   -> per-kernel durations
#   pass 1  --pmc SQ_* (instructions, wave cycles)               pass 2  --pmc FETCH_SIZE     pass 3  --pmc WRITE_SIZE
# (counter passes never combined with trace domains).  tools/make_profiles.py then writes profiles/<tag>_kernel_stats.md,
# profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json (with the workload it was measured on).
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline"
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan'
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
# the same bench, headline legs only, with ONE view in flight: every launch of the run has the chip to itself -> the exclusive durations
# the roofline uses (the secondary legs -- stage-2 batch on several streams, HIP graph child -- would mix overlapped launches in)
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace1 -o t -- $CMD --inflight 1 --headline-only > $OUT/trace1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/make_profiles.py $OUT $TAG
