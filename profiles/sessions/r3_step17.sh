#!/bin/bash
# 8-wave workgroups with two unit slots (CV_HL_NW8 bit mask per column width) and mask-group counts under the nontemporal
# partial tiles: net one in flight, scenes/s six in flight (240 steps, twice)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for cfg in "CV_HL_NW8=0 CV_NET_MASK_GROUPS=3" "CV_HL_NW8=4 CV_NET_MASK_GROUPS=3" "CV_HL_NW8=6 CV_NET_MASK_GROUPS=3" "CV_HL_NW8=0 CV_NET_MASK_GROUPS=2" "CV_HL_NW8=0 CV_NET_MASK_GROUPS=4"; do
  echo "$cfg: net $(env $cfg bash -c "$(declare -f one); one") | six in flight $(env $cfg bash -c "$(declare -f six); six") $(env $cfg bash -c "$(declare -f six); six")" | tee -a $O/nw8_groups.txt
done
