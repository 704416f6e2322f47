#!/bin/bash
# GPU box (through gpurun): the round's profile evidence, from the command the driver runs (bench.py), in separate passes:
#   pass 0  rocprofv3 --kernel-trace --stats: the default command (6 views in flight) and --inflight 1 --headline-only (every launch exclusive)
#   pass 1  --pmc SQ_* (instructions, wave cycles)   pass 2  --pmc FETCH_SIZE   pass 3  --pmc WRITE_SIZE
# The counter passes run the ONE-VIEW-IN-FLIGHT command (FETCH / WRITE of the default command are taken while six
# views share the L2s; the roofline's duration is the exclusive one, so its traffic is now measured in the same mode) and are never
# combined with trace domains.  tools/make_profiles.py then writes <tag>_kernel_stats.md, <tag>_kernel_stats_one_view.md,
# <tag>_pmc_summary.md and pmc_traffic.json (with the workload it was measured on) into gpurun_out/prof_<tag>/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 12 --warmup 3 --repeats 5 --no-cpu-baseline --no-configs"
ONE="$CMD --inflight 1 --headline-only"
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan'
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace1 -o t -- $ONE > $OUT/trace1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq -o p -- $ONE > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $ONE > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $ONE > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/make_profiles.py $OUT $TAG
