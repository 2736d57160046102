# what the HIP events of the timed region cost: every step / every 4th / (almost) none, with and without the pair around the vote kernel
cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'frac', round(d['roofline']['frac'],3))"; }
for rep in 1 2; do
for cfg in "--event-every 1 --kernel-events 1" "--event-every 1 --kernel-events 0" "--event-every 4 --kernel-events 1" "--event-every 4 --kernel-events 0" "--event-every 1000 --kernel-events 1"; do
  echo "$cfg: 20 steps $(run --gpus 1 --steps 20 --warmup 5 $cfg) | 240 steps $(run --steps 240 --warmup 5 $cfg)"
done; done
