cd $GRAFT_REPO_ROOT
for s in 6 8 6 8 6 8; do
  echo "streams=$s: 20 steps $(python3 bench.py --gpus 1 --steps 20 --warmup 5 --streams $s --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")  240 steps $(python3 bench.py --gpus 1 --steps 240 --warmup 5 --streams $s --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")"
done
