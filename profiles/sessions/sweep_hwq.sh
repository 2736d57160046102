# hardware queues (GPU_MAX_HW_QUEUES, ROCm default 4) x scenes in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_hwq.txt
: > $O
for q in 2 4 8 16; do
  for s in 6 8; do
    echo "GPU_MAX_HW_QUEUES=$q streams=$s" >> $O
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 240 --cpu-scenes 0 --streams $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))" >> $O
  done
done
cat $O
