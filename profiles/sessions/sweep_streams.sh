# scenes in flight sweep of the default bench (and the vote tests after the fill merge)
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_streams.txt
: > $O
timeout 900 python -m pytest tests/test_vote_gpu.py tests/test_decode_gpu.py -x -q 2>&1 | tail -2 >> $O
for s in 4 5 6 7 8 10; do
  echo "streams=$s" >> $O
  python bench.py --steps 240 --cpu-scenes 0 --streams $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms'], round(d['roofline']['frac'],4))" >> $O
done
cat $O
