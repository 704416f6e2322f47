mkdir -p gpurun_out/r05d
python tools/dev/diag_c5.py 2048 2400000 > gpurun_out/r05d/diag.log 2>&1
tail -25 gpurun_out/r05d/diag.log
