# explicit in-order lanes (threads share streams) against one stream per thread
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s28; mkdir -p $O
python -m pytest tests/test_scene_call_gpu.py tests/test_decode_gpu.py -q -m gpu 2>&1 | tail -2
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
for cfg in "--streams 7" "--streams 8 --lanes 4" "--streams 8 --lanes 4 --stagger-us 200" "--streams 12 --lanes 4" "--streams 10 --lanes 5" "--streams 6 --lanes 3" "--streams 8 --lanes 4 --stagger-us 800" "--streams 4 --lanes 4" "--streams 5 --lanes 5"; do
  echo -n "$cfg : 240 steps " >> $O/lanes.txt
  python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 $cfg 2>/dev/null | val >> $O/lanes.txt
  echo -n " | 20 steps " >> $O/lanes.txt
  for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 $cfg 2>/dev/null | val >> $O/lanes.txt; done
  echo >> $O/lanes.txt
done
CV_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --streams 8 --lanes 4 > $O/trace_l4.json 2> $O/trace_l4.txt
cat $O/lanes.txt; grep "^step" $O/trace_l4.txt
