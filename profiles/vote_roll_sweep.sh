#!/bin/bash
# rolling vote kernel: tile width x waves per workgroup (compile time) x target workgroup count (run time); vote op ms
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for cfg in "16 16" "32 8" "16 8" "32 4"; do
  set -- $cfg
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="-DHV_RTX=$1 -DHV_RTW=$2" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  for wgs in 384 768 1536; do export CV_HV_ROLL=1
    echo "RTX=$1 RTW=$2 target=$wgs: $(CV_HV_ROLL_WGS=$wgs python profiles/vote_time.py 2>&1 | grep 'event ms')" | tee -a $O/vote_roll_sweep.txt
  done
done
echo "old kernel: $(CV_HV_ROLL=0 python profiles/vote_time.py 2>&1 | grep 'event ms')" | tee -a $O/vote_roll_sweep.txt
