mkdir -p gpurun_out/r05k
O=gpurun_out/r05k
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_appearance.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" | tee $O/pytest_exit.txt
tail -3 $O/pytest.log
for lib in default gps-gaussian_amd/lib/abl/libgpsgs_hip_g1024.so gps-gaussian_amd/lib/abl/libgpsgs_hip_g256.so; do
  L=""; [ "$lib" != default ] && L="--lib $lib"
  for rep in 1 2; do
  python tools/stage_times.py --families tiles --steps 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['stages_us'])" | tee -a $O/stages.txt
  done
  python tools/stage_times.py --families tiles --steps 10 --res 1024 --render-res 2048 --gaussians 550000 --attributes untrained --seed-offset 77 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('regime $lib', d['stages_us'])" | tee -a $O/stages.txt
done
