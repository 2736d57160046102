# split target / mask groups of the network's convolutions with EIGHT scenes in flight (the round-2 tuning was done at six)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for cfg in "CV_SPLIT_TARGET=512" "CV_SPLIT_TARGET=384" "CV_SPLIT_TARGET=256" "CV_SPLIT_TARGET=128" "CV_SPLIT_TARGET=512 CV_NET_MASK_GROUPS=3" "CV_SPLIT_TARGET=256 CV_SPLIT_TRAFFIC_MB=12"; do
  echo "$cfg: $(env $cfg bash -c "$(declare -f run); run") $(env $cfg bash -c "$(declare -f run); run")"
done
