# the 20-step command against the number of hardware queues (GPU_MAX_HW_QUEUES, default 4) and scene threads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s26; mkdir -p $O
run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
for q in 4 8 16; do for s in 8 7 10; do
  echo -n "hwq $q streams $s : " >> $O/hwq_20steps.txt
  for i in 1 2 3 4; do GPU_MAX_HW_QUEUES=$q run --streams $s >> $O/hwq_20steps.txt; done
  echo >> $O/hwq_20steps.txt
done; done
GPU_MAX_HW_QUEUES=8 CV_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/trace_q8.json 2> $O/trace_q8.txt
for q in 4 8; do echo -n "hwq $q 240 steps: " >> $O/hwq_20steps.txt; GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'])" >> $O/hwq_20steps.txt; done
cat $O/hwq_20steps.txt; grep "^step" $O/trace_q8.txt
