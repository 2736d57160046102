#!/bin/bash
# A/B of the nontemporal policy on the partial tiles (CV_NT_PARTIAL, compile time: bit 0 stores, bit 1 finish loads) and
# of the XCD-aware tile numbering (CV_XCD_TILES, run time) on top: net stage one scene in flight, scenes/s default streams
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 120 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for nt in 0 1 3; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DCV_NT_PARTIAL=$nt" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  for xcd in 0 1; do
    echo "CV_NT_PARTIAL=$nt CV_XCD_TILES=$xcd: one in flight $(CV_XCD_TILES=$xcd one) | $(CV_XCD_TILES=$xcd one) ; six in flight $(CV_XCD_TILES=$xcd six)" | tee -a $O/nt_partial_ab.txt
  done
done
