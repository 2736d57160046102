# stem conv (5x5x5, 3 -> 32): map offsets staged per chunk (CV_STEM_JC); 125 = the whole slice resident (the previous kernel)
cd $GRAFT_REPO_ROOT
O=gpurun_out/stem_sweep.txt
: > $O
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -k "minkunet or conv_matches or modules" 2>&1 | tail -1 >> $O
for jc in 125 64 32 16 8; do
  echo "CV_STEM_JC=$jc" >> $O
  CV_STEM_JC=$jc python profiles/layer_times.py 2>&1 | grep -E "^ +0 " >> $O
done
python bench.py --streams 1 --cpu-scenes 0 --steps 120 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one in flight', round(d['value'],1), d['stage_ms'])" >> $O
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('six in flight', round(d['value'],1))" >> $O
cat $O
