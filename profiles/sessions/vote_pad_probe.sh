#!/bin/bash
# scenes-in-flight finding (profiles/r3/vote_concurrency_findings.txt): does the 16 x 32 / 8-wave tile kernel need a CONVOLUTION
# workgroup on its own CU to go wrong?  LDS padding controls what fits next to it: 68 KB (two tile workgroups + convs),
# ~100 KB (one tile workgroup + convs), ~140 KB (one tile workgroup, no room for a conv workgroup's 26 KB)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-vote_pad}; mkdir -p $O
for cfg in "-DHV_TX=16 -DHV_TW=8" "-DHV_TX=16 -DHV_TW=8 -DHV_LDS_PAD=32768" "-DHV_TX=16 -DHV_TW=8 -DHV_LDS_PAD=73728" "-DHV_TX=16 -DHV_TW=4 -DHV_LDS_PAD=92160" "-DHV_TX=32 -DHV_TW=16"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$cfg" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  rm -f gpurun_out/vote_ref.pt
  echo "== $cfg" | tee -a $O/vote_pad_probe.txt
  python profiles/vote_race_probe3.py 2>&1 | grep -E "interference" | head -4 | tee -a $O/vote_pad_probe.txt
done
