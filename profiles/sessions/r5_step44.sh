# round 5, step 44: the occupancy bitmap filled by the level build's insert pass (no bitmap_set launch): exactness, rates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s44
mkdir -p $O
timeout 2400 python -m pytest tests/test_sparse_gpu.py tests/test_scene_call_gpu.py tests/test_production_size_gpu.py tests/test_windows_gpu.py tests/test_concurrency_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
cat $O/pytest.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('240 steps:', round(d['value'],1))" >> $O/rates.txt
  timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight:', round(d['value'],1))" >> $O/rates.txt
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 steps:', round(d['value'],1))" >> $O/rates.txt
done
cat $O/rates.txt
