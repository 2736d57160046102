# round 5, step 17: conv_hh (conv_hd with half-chunk ring stages, two workgroups per CU): bit identity, per-layer time, scene rates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s17
mkdir -p $O
timeout 600 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "conv_hd or zskip" 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python profiles/hh_micro.py 50 > $O/hh_micro.txt 2>&1
cat $O/hh_micro.txt
for shape in 2 3 2 3; do
  CV_HD_SHAPE=$shape timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shape $shape 240 steps:', round(d['value'],1), 'net', d.get('stage_ms',{}))" >> $O/rates.txt
  CV_HD_SHAPE=$shape timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shape $shape one in flight:', round(d['value'],1))" >> $O/rates.txt
done
cat $O/rates.txt
