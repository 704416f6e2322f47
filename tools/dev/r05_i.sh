mkdir -p gpurun_out/r05i
O=gpurun_out/r05i
for pad in 0 4096 65536 1048576 9600000 3145728; do
  GPSGS_DEBUG_TAIL_PAD=$pad python tools/stage_times.py --families tiles --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($pad, d['stages_us'])" | tee -a $O/pads.txt
done
