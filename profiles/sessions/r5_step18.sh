# round 5, step 18: mask groups summed inside the conv_hd launch (option "gfuse"): bit identity, then scene rates A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s18
mkdir -p $O
timeout 600 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "summed_inside or conv_hd" 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
grep -q passed $O/pytest.txt || exit 1
for g in 0 1 0 1; do
  CV_GFUSE=$g timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gfuse $g 240 steps:', round(d['value'],1), 'parity', d.get('parity'))" >> $O/rates.txt
  CV_GFUSE=$g timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gfuse $g one in flight:', round(d['value'],1), d.get('stage_ms_isolated'))" >> $O/rates.txt
done
cat $O/rates.txt
