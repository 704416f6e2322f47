#!/bin/bash
# round 3, GPU call K2: large-list sort on up to 1,024 workgroups -- sort-path tests, config 3 timed, rocprofv3 kernel stats of config 3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
ROOT=$PWD
mkdir -p gpurun_out/prof_r03_config3 gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "sort_path or large_splats or skipped_large or huge" 2>&1 | tail -3
timeout 900 python tools/run_reference.py interp --res 1024 --samples 3 --views 5 --work /tmp/w3 > gpurun_out/r03/config3.json 2> gpurun_out/r03/config3.err
echo "exit $?"; cat gpurun_out/r03/config3.json
cd /tmp
rm -rf $ROOT/gpurun_out/prof_r03_config3/*
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/prof_r03_config3 -o t -- python $ROOT/tools/run_reference.py interp --res 1024 --samples 2 --views 5 --work /tmp/w3 > $ROOT/gpurun_out/prof_r03_config3/run.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob
rows=[]
for fn in glob.glob('gpurun_out/prof_r03_config3/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(fn)))
for r in rows:
    n=r["Name"]
    if any(x in n for x in ("k_composite","k_preprocess","k_scatter","k_sort","k_scan","k_cs_")):
        print(n.replace("(anonymous namespace)::","").split("(")[0][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg")
PY
