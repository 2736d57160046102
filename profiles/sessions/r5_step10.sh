cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s9
mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" >> $O/tail_priority.txt 2>&1
for rep in 1 2 3; do for tp in 0 -3 -6; do
  v=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --tail-priority=$tp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['step_host_ms'])")
  echo "tail_priority=$tp rep=$rep 20 steps: $v" >> $O/tail_priority.txt
done; done
tail -10 $O/tail_priority.txt
