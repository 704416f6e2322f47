ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ONE="python $ROOT/bench.py --steps 12 --warmup 3 --repeats 3 --no-cpu-baseline --no-configs --no-full-pipeline --inflight 1 --headline-only"
KRE='k_preprocess|k_scatter|k_sort|k_scan'
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace1 -o t -- $ONE > $OUT/trace1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_a -o p -- $ONE > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_b -o p -- $ONE > $OUT/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_c -o p -- $ONE > $OUT/pmc_c.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --kernel-include-regex "$KRE" -f csv -d $OUT/pmc_d -o p -- $ONE > $OUT/pmc_d.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT $OUT/pmc.md > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05g"
for fn in glob.glob(out + "/trace1/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(fn)))
    with open(out + "/kernel_stats.txt", "w") as f:
        for r in rows[:16]:
            f.write("%-60s calls %s avg_us %.2f total%% %s\n" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))
PY
cat $OUT/kernel_stats.txt; cat $OUT/pmc.md; tail -2 $OUT/pmc_d.log
rm -rf $OUT/trace1 $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/pmc_d
