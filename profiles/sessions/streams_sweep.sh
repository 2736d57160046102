# scenes in flight (bench.py --streams) at the driver's 20 steps and at 240 steps, final tree
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --gpus 1 --steps $2 --warmup 5 --streams $1 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for rep in 1 2 3; do for s in 8 10 7 12; do echo "streams=$s: 20 steps $(run $s 20)"; done; done
for s in 8 10 12; do echo "streams=$s: 240 steps $(run $s 240)"; done
