#!/bin/bash
# decode greedy variants measured inside the bench (one scene in flight; the isolated decode_time.py runs at idle clocks)
O=gpurun_out/r3h; mkdir -p $O
for defs in "-DDEC_BLOCKED=1 -DDEC_CACHED=1 -DDEC_BBOX=1" "-DDEC_BLOCKED=0 -DDEC_CACHED=0 -DDEC_BBOX=0" "-DDEC_BLOCKED=0 -DDEC_CACHED=1 -DDEC_BBOX=0"; do
  CV_DEC_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "== $defs" >> $O/decode_variants_bench.txt
  python bench.py --streams 1 --stage vote_decode --steps 60 --warmup 5 --cpu-scenes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms'], d['stage_ms_median'])" >> $O/decode_variants_bench.txt
done
cat $O/decode_variants_bench.txt
python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
for l in 1 0; do echo "== CV_HV_LISTS=$l" >> $O/vote_bench.txt; CV_HV_LISTS=$l python bench.py --streams 1 --stage vote_decode --steps 60 --warmup 5 --cpu-scenes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms_median'], d['roofline']['frac'])" >> $O/vote_bench.txt; done
for l in 2 1; do echo "== 300k CV_HV_LISTS=$l" >> $O/vote_bench.txt; CV_HV_LISTS=$l python bench.py --streams 1 --stage vote_decode --steps 30 --warmup 5 --cpu-scenes 0 --large --points 300000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms_median'], d['roofline']['frac'])" >> $O/vote_bench.txt; done
cat $O/vote_bench.txt
