#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-vote_pad}; mkdir -p $O
for cfg in "-DHV_TX=16 -DHV_TW=8 -DHV_LDS_PAD=14000" "-DHV_TX=16 -DHV_TW=8 -DHV_LDS_PAD=50000"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$cfg" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  rm -f gpurun_out/vote_ref.pt
  echo "== $cfg" | tee -a $O/vote_pad_probe.txt
  python profiles/vote_race_probe3.py 2>&1 | grep -E "interference" | head -4 | tee -a $O/vote_pad_probe.txt
  python profiles/vote_time.py 2>&1 | grep "event ms" | tee -a $O/vote_pad_probe.txt
done
