#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ai; mkdir -p $O
touch canonicalvoting_amd/csrc/hv_vote.hip
CV_HV_DEFS="-DHV_TX=16 -DHV_TW=8" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
for lists in 1 2; do
  export CV_HV_LISTS=$lists
  echo "HV_TX=16 HV_TW=8 CV_HV_LISTS=$lists: one in flight $(python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'vote ms', round(d['stage_ms_median']['vote'],4))")  six in flight $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))") $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")" | tee -a $O/tile16_lists.txt
done
unset CV_HV_LISTS
touch canonicalvoting_amd/csrc/hv_vote.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
