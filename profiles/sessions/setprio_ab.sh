#!/bin/bash
# s_setprio 1 around the MFMA cluster of a conv_hl unit (HL_SETPRIO, compile time): net one in flight, scenes/s six in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for v in 0 1; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DHL_SETPRIO=$v" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "HL_SETPRIO=$v: net $(one) $(one) | six in flight $(six) $(six)" | tee -a $O/setprio_ab.txt
done
touch canonicalvoting_amd/csrc/sparse_conv.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
