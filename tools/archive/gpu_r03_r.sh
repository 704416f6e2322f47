#!/bin/bash
# round 3, GPU call R: GSR_FLAG_WAVE_PRIORITY end to end (kernel times with the flag on / off, the bitwise test, raster parity), and the forward's
# colour sum by parts against the plain form with several views in flight (bench `value`)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
L=gps-gaussian_amd/lib
OUT=gpurun_out/r03/call_r.log
: > $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_us']; print('fwd %.2f bwd %.2f step %.1f views/s %.0f' % (s['composite_fwd'], s['composite_bwd'], d['sum_us'], d['views_per_s']))"; }
cp $L/variants/xp.so $L/libgpsgs_hip.so
for wp in 1 0 1 0; do
  echo "== xp GPSGS_WAVE_PRIORITY=$wp" | tee -a $OUT; GPSGS_WAVE_PRIORITY=$wp timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | show | tee -a $OUT
done
echo "== parity xp" | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_pack.py tests/test_gpu_capi_host.py -q -m gpu 2>&1 | grep -E "^E  .*Error|passed|failed|FAILED" | cut -c1-200 | head -20 | tee -a $OUT
for v in $BENCH; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== bench $v" | tee -a $OUT
  timeout 600 python bench.py --no-configs --repeats 9 > gpurun_out/r03/bench_r_$v.json 2> gpurun_out/r03/bench_r_$v.err
  python - $v <<'PY' | tee -a $OUT
import json,sys
d=json.loads([l for l in open('gpurun_out/r03/bench_r_%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"], "s2", d["stage2_gradient_set"]["views_per_s"], d["stage2_gradient_set"]["single_view_in_flight_views_per_s"])
print({k:v["avg_us"] for k,v in d["stages"].items()}, d["roofline"]["avg_launch_us"], d["roofline"]["shader_clock_mhz"], "stage2_path", d["stage2_path"]["ms_per_iter"])
PY
done
cp $L/variants/xp.so $L/libgpsgs_hip.so
