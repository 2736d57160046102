# round 5, step 59: split target 768 as the one-scene-at-a-time default: tests (summation order moves), rates, training
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s59
mkdir -p $O
timeout 2400 python -m pytest tests/test_sparse_gpu.py tests/test_scene_call_gpu.py tests/test_production_size_gpu.py tests/test_train_gpu.py tests/test_concurrency_gpu.py -m gpu -q 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight', round(d['value'],1), 'net', round(d['stage_ms']['net'],3))" >> $O/rates.txt
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('240 steps', round(d['value'],1), '| net one in flight', round(d['stage_ms_isolated']['net'],3))" >> $O/rates.txt
done
timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train step', round(d['ms_per_step'],2))" >> $O/rates.txt
timeout 300 python bench.py --streams 1 --steps 40 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --large --points 300000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('300k one in flight', round(d['value'],1))" >> $O/rates.txt
cat $O/rates.txt
