mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
python tools/dev/diag_c5.py 2048 2400000 > $O/diag.log 2>&1; tail -4 $O/diag.log
python tools/stage_times.py --families tiles --steps 40 > $O/new.json 2>$O/new.err
for v in sort_nonet sort_nofix sc_noclear; do
  python tools/stage_times.py --families tiles --steps 40 --lib gps-gaussian_amd/lib/abl/libgpsgs_hip_$v.so > $O/$v.json 2>$O/$v.err
done
cat $O/*.json
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_appearance.py tests/test_gpu_raster_inputs.py tests/test_gpu_capi_host.py tests/test_gpu_pack.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
tail -5 $O/pytest.log
