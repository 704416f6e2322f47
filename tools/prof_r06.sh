#!/bin/bash
# GPU box (through gpurun): the round's profile evidence, in separate passes (counter passes are never combined with trace domains):
#   trace   rocprofv3 --kernel-trace --stats: the default bench command (several views in flight in its session region) and
#           --inflight 1 --headline-only (every launch has the chip to itself: the exclusive durations)
#   pmc     --pmc SQ_* (instructions, wave / wait cycles), FETCH_SIZE, WRITE_SIZE, TCP->TCC requests on the ONE-VIEW-IN-FLIGHT command
#   regime  FETCH_SIZE / WRITE_SIZE / SQ instruction counts / kernel trace of the untrained-heads regime (bench.py's config3_regime leg: P = 550,000, 2048^2, scales at their clamp)
# tools/make_profiles.py then writes <tag>_kernel_stats.md, <tag>_kernel_stats_one_view.md, <tag>_pmc_summary.md, pmc_traffic.json,
# <tag>_regime_pmc_summary.md and pmc_traffic_regime.json (each with the workload it was measured on) into gpurun_out/prof_<tag>/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 12 --warmup 3 --repeats 5 --no-cpu-baseline --no-configs --no-full-pipeline"
ONE="$CMD --inflight 1 --headline-only"
REG="python $ROOT/tools/stage_times.py --families tiles --steps 10 --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --seed-offset 77"
KRE='k_composite|k_preprocess|k_scatter|k_sort|k_scan|k_zero16'
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace1 -o t -- $ONE > $OUT/trace1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq -o p -- $ONE > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_sq2 -o p -- $ONE > $OUT/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_fetch -o p -- $ONE > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_write -o p -- $ONE > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_tcc -o p -- $ONE > $OUT/pmc_tcc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/regime_trace -o t -- $REG > $OUT/regime_trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/regime_pmc_fetch -o p -- $REG > $OUT/regime_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/regime_pmc_write -o p -- $REG > $OUT/regime_pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-include-regex "$KRE" -f csv -d $OUT/regime_pmc_sq -o p -- $REG > $OUT/regime_pmc_sq.log 2>&1
cd $ROOT
python tools/make_profiles.py $OUT $TAG
