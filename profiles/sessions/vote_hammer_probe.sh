#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-vote_hammer}; mkdir -p $O
(cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -shared -fPIC lds_hammer.hip -o liblds_hammer.so 2>/dev/null)
touch canonicalvoting_amd/csrc/hv_vote.hip
CV_HV_DEFS="-DHV_TX=16 -DHV_TW=8" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
echo "tile kernel 16 x 32 cells / 8 waves (68 KB of LDS)" | tee -a $O/vote_hammer_probe.txt
timeout 300 python profiles/vote_hammer_probe.py 2>&1 | grep "co-resident" | tee -a $O/vote_hammer_probe.txt
