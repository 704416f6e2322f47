#!/bin/bash
# round 3, GPU call M: A/B of compositing-kernel variants on ONE box (clocks differ between boxes): per-kernel hipEvent times of the step for
# each prebuilt library under gps-gaussian_amd/lib/variants/, then the raster parity tests + a short bench on the candidate (last variant named)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
VARIANTS=${VARIANTS:-"base e1 e5 base e1 e5"}
L=gps-gaussian_amd/lib
for v in $VARIANTS; do
  cp $L/variants/$v.so $L/libgpsgs_hip.so
  echo "== $v"; timeout 300 python tools/stage_times.py --families tiles --steps 100 2>&1 | tail -1 | cut -c1-420
done
CAND=${CAND:-e1}
cp $L/variants/$CAND.so $L/libgpsgs_hip.so
echo "== parity on $CAND"
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_raster_inputs.py tests/test_gpu_pack.py tests/test_gpu_capi_host.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --no-configs --repeats 7 > gpurun_out/r03/bench_m.json 2> gpurun_out/r03/bench_m.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_m.json') if l.startswith('{')][-1])
print("value", d["value"], d["ms_per_step_iqr"], "single", d["single_view_in_flight_views_per_s"], "api", d["autograd_api_views_per_s"], "fwd", d["forward_only_views_per_s"])
print("s2", json.dumps(d["stage2_gradient_set"])[-160:]); print("roofline", {k:d["roofline"][k] for k in ("avg_launch_us","frac","shader_clock_mhz")})
print({k:v["avg_us"] for k,v in d["stages"].items()})
PY
