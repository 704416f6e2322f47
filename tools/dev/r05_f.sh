mkdir -p gpurun_out/r05f
O=gpurun_out/r05f
python tools/stage_times.py --families tiles --steps 40 > $O/new.json 2>$O/new.err
cat $O/new.json
timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -p no:cacheprovider -k "parity or config2 or sort_path or fuzz or clamp or overflow or large_splats_gradients or stale" 2>&1 | tail -5 > $O/pytest.log
tail -3 $O/pytest.log
