mkdir -p gpurun_out/r05l
O=gpurun_out/r05l
for pad in 0 33000 50000 20000; do
  GPSGS_DEBUG_PRE_LDS=$pad python tools/stage_times.py --families tiles --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pre $pad', d['stages_us'])" | tee -a $O/stages.txt
done
for pad in 27000 32768 40000 54000; do
  GPSGS_DEBUG_PBWD_LDS=$pad python tools/stage_times.py --families tiles --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pbwd $pad', d['stages_us'])" | tee -a $O/stages.txt
done
