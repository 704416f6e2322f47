O=gpurun_out/r05n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace1 -o t -- python $R/bench.py --steps 12 --warmup 3 --repeats 5 --no-cpu-baseline --no-configs --no-full-pipeline --inflight 1 --headline-only > $R/$O/trace1.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05n/trace1/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -2 $O/trace1.log | cut -c1-400
rm -rf $O/trace1
