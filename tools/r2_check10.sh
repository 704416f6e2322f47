#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2c10
for F in 1 3 4 6; do
timeout 600 python bench.py --steps 24 --warmup 5 --inflight $F --no-cpu-baseline > gpurun_out/r2c10/bench_F$F.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r2c10/bench_F$F.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print($F, {k: d[k] for k in ('value', 'ms_per_step', 'repeats_ms_per_step', 'single_view_in_flight_views_per_s', 'autograd_api_views_per_s')}, d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['frac_one_view_in_flight'])
PY
done
tail -2 gpurun_out/r2c10/bench_F4.log | cut -c1-400
