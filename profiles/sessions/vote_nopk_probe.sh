#!/bin/bash
# after the build lost its packed fp32 instructions: the tile shape that went wrong next to the convolutions (16 x 32 / 8 waves,
# two workgroups per CU), under the convolutions of six streams and under the synthetic 16-bit MFMA load
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-vote_nopk}; mkdir -p $O
(cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC lds_hammer.hip -o liblds_hammer.so 2>/dev/null)
for cfg in "-DHV_TX=16 -DHV_TW=8" "-DHV_TX=16 -DHV_TW=4"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$cfg" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  rm -f gpurun_out/vote_ref.pt
  echo "== no packed fp32, tile kernel $cfg" | tee -a $O/vote_nopk_probe.txt
  python profiles/vote_race_probe3.py 2>&1 | grep -E "interference" | tee -a $O/vote_nopk_probe.txt
  timeout 300 python profiles/vote_hammer_probe.py 0 3 5 2>&1 | grep "co-resident" | tee -a $O/vote_nopk_probe.txt
done
touch canonicalvoting_amd/csrc/hv_vote.hip
CV_HV_DEFS="-DHV_TX=16 -DHV_TW=8" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
python -m pytest tests/test_concurrency_gpu.py tests/test_vote_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/vote_nopk_probe.txt
for i in 1 2; do echo "16x32/8: $(python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms_median'])")  six: $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")" | tee -a $O/vote_nopk_probe.txt; done
touch canonicalvoting_amd/csrc/hv_vote.hip
python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
for i in 1 2; do echo "32x32/16: $(python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms_median'])")  six: $(python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")" | tee -a $O/vote_nopk_probe.txt; done
