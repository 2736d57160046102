#!/bin/bash
# vote tile shapes on the round-3 kernel (no packed fp32 instructions: every shape is exact under scenes in flight now):
# vote op ms one scene in flight, scenes/s one / six in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-vote_tile_sweep}; mkdir -p $O
for cfg in "32 16" "16 8" "16 16" "32 8" "16 4" "8 8"; do
  set -- $cfg
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="-DHV_TX=$1 -DHV_TW=$2" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "HV_TX=$1 HV_TW=$2: one in flight $(python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'vote ms', round(d['stage_ms_median']['vote'],4))")  six in flight $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))") $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")" | tee -a $O/vote_tile_sweep.txt
done
touch canonicalvoting_amd/csrc/hv_vote.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
