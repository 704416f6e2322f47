#!/bin/bash
# GPU box: exclusive per-kernel durations (rocprofv3 --kernel-trace --stats) of a stage_times.py command; prints the top of the stats table.
# usage: tools/prof_quick.sh <tag> [stage_times.py arguments...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $ROOT/tools/stage_times.py --families tiles --steps 40 "$@" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s  avg %9.2f us  min %8.2f  max %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
