ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
rm -f $ROOT/gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest.log
tail -4 $OUT/pytest.log
export TMPDIR=/tmp
cd /tmp
ONE="python $ROOT/bench.py --steps 12 --warmup 3 --repeats 3 --no-cpu-baseline --no-configs --no-full-pipeline --inflight 1 --headline-only"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace1 -o t -- $ONE > $OUT/trace1.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05h"
for fn in glob.glob(out + "/trace1/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(fn)))
    with open(out + "/kernel_stats.txt", "w") as f:
        for r in rows[:10]:
            f.write("%-60s calls %s avg_us %.2f\n" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat $OUT/kernel_stats.txt; tail -1 $OUT/trace1.log | cut -c1-600
rm -rf $OUT/trace1
