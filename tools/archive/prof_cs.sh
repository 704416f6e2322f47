#!/bin/bash
# GPU box (through gpurun): rocprofv3 rows + HBM traffic of the drop-in correlation sampler (row a14) -> gpurun_out/prof_cs/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_cs
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/bench_cs_dropin.py 2"
timeout 300 $CMD > $OUT/plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_cs_" -f csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_cs_" -f csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-include-regex "k_cs_" -f csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
d = "$OUT"
print(open(d + "/plain.log").read()[-900:])
for fn in glob.glob(d + "/trace/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(fn)))[:12]:
        print("%-70s calls %5s avg %8.2f us  min %8.2f  max %8.2f  %s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", max(len(x) for x in v.values()))
PY
