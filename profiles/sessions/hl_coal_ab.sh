#!/bin/bash
# A/B of the line-coalesced gather of conv_hl (HL_COAL bit mask per column width, compile time) and of the 8-wave
# workgroups (CV_HL_NW8 bit mask, run time): net stage one scene in flight, scenes/s with the default streams
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 120 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for coal in 7 3 0; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DHL_COAL=$coal" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  for nw8 in 0 4 7; do
    if [ $coal != 7 ] && [ $nw8 != 0 ]; then continue; fi
    echo "HL_COAL=$coal CV_HL_NW8=$nw8: one in flight $(CV_HL_NW8=$nw8 one) | $(CV_HL_NW8=$nw8 one) ; six in flight $(CV_HL_NW8=$nw8 six)" | tee -a $O/hl_coal_ab.txt
  done
done
