# hardware queues x scene threads, 240 steps and the 20-step command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s27; mkdir -p $O
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
for q in 2 3 4 5 6; do for s in 6 7 8 10 12; do
  echo -n "hwq $q streams $s : 240 steps " >> $O/hwq_grid.txt
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --streams $s 2>/dev/null | val >> $O/hwq_grid.txt
  echo -n " | 20 steps " >> $O/hwq_grid.txt
  for i in 1 2; do GPU_MAX_HW_QUEUES=$q python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --streams $s 2>/dev/null | val >> $O/hwq_grid.txt; done
  echo >> $O/hwq_grid.txt
done; done
cat $O/hwq_grid.txt
