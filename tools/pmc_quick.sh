#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (counters only, no trace domains) over a stage_times.py command; mean per launch of each counter for kernels matching a regex.
# usage: tools/pmc_quick.sh <tag> <kernel-regex> "<counters>" [stage_times.py arguments...]
TAG=$1; KRE=$2; CTRS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc $CTRS --kernel-include-regex "$KRE" -f csv -d $OUT -o p -- python $ROOT/tools/stage_times.py --families tiles --steps 12 "$@" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %14.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
