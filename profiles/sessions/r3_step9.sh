#!/bin/bash
# vote work queue: PART_VOTES sweep (compile-time), 80k streaming and 300k lists
O=gpurun_out/r3j; mkdir -p $O; export TMPDIR=/tmp
for pv in 16384 32768 65536 131072; do
  CV_HV_DEFS="-DHV_PART_VOTES=$pv" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "== PART_VOTES $pv" >> $O/sweep.txt
  python profiles/vote_time.py 2>&1 | grep "event ms" >> $O/sweep.txt
  python profiles/vote_time.py --large 2>&1 | grep "event ms" >> $O/sweep.txt
  python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms_median'], d['roofline']['frac'])" >> $O/sweep.txt
done
cat $O/sweep.txt
