#!/bin/bash
# compile-time sweeps, REBUILT with --force each time (the staleness check does not see the environment):
# decode greedy-walk variants and the vote queue's PART_VOTES, measured inside the bench with one scene in flight
O=gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,4) for k,v in d['stage_ms_median'].items()}, round(d['roofline']['frac'],4))"; }
for defs in "-DDEC_BLOCKED=0 -DDEC_CACHED=0 -DDEC_BBOX=0" "-DDEC_BLOCKED=1 -DDEC_CACHED=1 -DDEC_BBOX=1" "-DDEC_BLOCKED=0 -DDEC_CACHED=1 -DDEC_BBOX=0" "-DDEC_BLOCKED=1 -DDEC_CACHED=0 -DDEC_BBOX=1"; do
  CV_DEC_DEFS="$defs" python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1
  echo "== $defs" >> $O/decode_variants.txt
  python bench.py --streams 1 --steps 60 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | line >> $O/decode_variants.txt
done
cat $O/decode_variants.txt
for pv in 4096 8192 32768 1000000; do
  CV_HV_DEFS="-DHV_PART_VOTES=$pv" python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1
  echo "== PART_VOTES $pv" >> $O/sweep.txt
  python bench.py --streams 1 --steps 60 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | line >> $O/sweep.txt
  python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 --stage vote_decode 2>/dev/null | line >> $O/sweep.txt
done
cat $O/sweep.txt
python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1
