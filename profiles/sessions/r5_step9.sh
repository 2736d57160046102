# round 5, step 9: the last N steps of the driver's 20-step command on high-priority streams (bench.py --tail-priority)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s9
mkdir -p $O
: > $O/tail_priority.txt
for rep in 1 2 3; do for tp in 0 3 6 7 12; do
  v=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --tail-priority $tp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['step_host_ms'])")
  echo "tail_priority=$tp rep=$rep 20 steps: $v" >> $O/tail_priority.txt
done; done
for tp in 0 7; do
  v=$(python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --tail-priority $tp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "tail_priority=$tp 240 steps: $v" >> $O/tail_priority.txt
done
cat $O/tail_priority.txt
