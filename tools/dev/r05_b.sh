mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
python tools/stage_times.py --families tiles --steps 40 > $O/base.json 2>$O/base.err
python tools/stage_times.py --families tiles --steps 40 --gaussians 524288 > $O/base_p512k.json 2>>$O/base.err
for v in rows nofence sort32 r5a; do
  python tools/stage_times.py --families tiles --steps 40 --lib gps-gaussian_amd/lib/abl/libgpsgs_hip_$v.so > $O/$v.json 2>$O/$v.err
done
GPSGS_LIB=$PWD/gps-gaussian_amd/lib/abl/libgpsgs_hip_r5a.so timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_appearance.py -m gpu -q -x -p no:cacheprovider -k "not config5 and not tens_of_millions and not full_size_lists" 2>&1 | tail -30 > $O/pytest_r5a.log
tail -3 $O/pytest_r5a.log
cat $O/*.json
