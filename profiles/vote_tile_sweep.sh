# vote tile shape / waves per workgroup (compile-time HV_TX, HV_TW) on the headline workload's predictions
cd $GRAFT_REPO_ROOT
O=gpurun_out/vote_tile_sweep.txt
: > $O
for defs in "-DHV_PART_RECORDS=4096 -DHV_MAX_PARTS=8" "-DHV_PART_RECORDS=2048 -DHV_MAX_PARTS=8" "-DHV_PART_RECORDS=2048 -DHV_MAX_PARTS=16" "-DHV_PART_RECORDS=8192 -DHV_MAX_PARTS=8"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "== $defs" >> $O
  timeout 300 python -m pytest tests/test_vote_gpu.py -x -q 2>&1 | tail -1 >> $O
  python bench.py --streams 1 --steps 60 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('net-pred vote_ms', round(d['stage_ms']['vote'],4), 'frac', round(d['roofline']['frac'],4))" >> $O
  python bench.py --streams 1 --steps 60 --cpu-scenes 0 --teacher-forced 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('teacher vote_ms', round(d['stage_ms']['vote'],4), 'frac', round(d['roofline']['frac'],4))" >> $O
  python bench.py --steps 240 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('six in flight scenes/s', round(d['value'],1))" >> $O
done
python bench.py --streams 1 --steps 20 --cpu-scenes 0 --algo 24 2>&1 | grep -i "ticks" | tail -2 >> $O
cat $O
