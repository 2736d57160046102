# the round-3 conv experiments once more in the final throughput setting (8 scenes in flight, split target 256), scenes/s at 240 steps
cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for cfg in "CV_NONE=0" "CV_NB_WIDE=4" "CV_NB_COARSE=4" "CV_HL_NS1=7" "CV_HL_NS2=0" "CV_GROUP_CHAIN=1" "CV_HL_FUSE_FINISH=0"; do
  echo "$cfg: $(env $cfg bash -c "$(declare -f run); run") $(env $cfg bash -c "$(declare -f run); run")"
done
