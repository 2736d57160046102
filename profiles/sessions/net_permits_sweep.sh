# needs the --net-permits patch of bench.py described in profiles/r3/net_permits_sweep.txt (not kept)
cd $GRAFT_REPO_ROOT
for cfg in "0" "5" "4" "6" "0" "5" "4"; do
  echo "net-permits=$cfg: 20 steps $(python3 bench.py --gpus 1 --steps 20 --warmup 5 --net-permits $cfg --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")  240 steps $(python3 bench.py --gpus 1 --steps 240 --warmup 5 --net-permits $cfg --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")"
done
