# the driver's 20-step command against the number of scene threads and the start stagger (fresh process each)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s24; mkdir -p $O
run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
for cfg in "--streams 8" "--streams 10" "--streams 10 --stagger-us 200" "--streams 10 --stagger-us 0" "--streams 7" "--streams 12" "--streams 16" "--streams 20" "--streams 20 --stagger-us 100" "--streams 5"; do
  echo -n "$cfg : " >> $O/streams_20steps.txt
  for i in 1 2 3 4; do run $cfg >> $O/streams_20steps.txt; done
  echo >> $O/streams_20steps.txt
done
cat $O/streams_20steps.txt
