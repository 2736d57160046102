cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s25; mkdir -p $O
for s in 8 7; do
  CV_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --streams $s > $O/trace_s$s.json 2> $O/trace_s$s.txt
done
for s in 7 8 7 8; do
  echo -n "streams $s 240 steps: " >> $O/s7_240.txt
  python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --streams $s 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])" >> $O/s7_240.txt
done
cat $O/s7_240.txt; grep "^step" $O/trace_s8.txt; grep "^step" $O/trace_s7.txt
