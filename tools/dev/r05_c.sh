mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_appearance.py tests/test_gpu_raster_inputs.py tests/test_gpu_capi_host.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
tail -3 $O/pytest.log
python tools/stage_times.py --families tiles --steps 40 > $O/new.json 2>$O/new.err
python tools/stage_times.py --families tiles --steps 40 --attributes untrained --render-res 2048 --gaussians 550000 > $O/new_regime.json 2>>$O/new.err
cat $O/new.json $O/new_regime.json
