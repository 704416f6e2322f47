#!/bin/bash
# round 3, GPU call S: the backward without the wave-uniform skip of entries no pixel blends (a branch per entry on a VALU -> SALU round trip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
L=gps-gaussian_amd/lib
OUT=gpurun_out/r03/call_s.log
: > $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"; }
for v in $VARIANTS; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v" | tee -a $OUT; timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | show | tee -a $OUT
done
for v in $BENCH; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== bench $v" | tee -a $OUT
  timeout 600 python bench.py --no-configs --repeats 9 > gpurun_out/r03/bench_s_$v.json 2> gpurun_out/r03/bench_s_$v.err
  python - $v <<'PY' | tee -a $OUT
import json,sys
d=json.loads([l for l in open('gpurun_out/r03/bench_s_%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"], "s2", d["stage2_gradient_set"]["views_per_s"], d["stage2_gradient_set"]["single_view_in_flight_views_per_s"])
print({k:v["avg_us"] for k,v in d["stages"].items()}, d["roofline"]["avg_launch_us"], d["roofline"]["shader_clock_mhz"], "stage2_path", d["stage2_path"]["ms_per_iter"])
PY
done
cp $L/variants/fin.so $L/libgpsgs_hip.so
