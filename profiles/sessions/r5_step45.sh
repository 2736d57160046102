# round 5, step 45: the range flag and the sort's bound words initialised by the bounds' final launch (two fill launches fewer): exactness, rates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s45
mkdir -p $O
timeout 1800 python -m pytest tests/test_scene_call_gpu.py tests/test_concurrency_gpu.py tests/test_vote_gpu.py tests/test_decode_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 1 --cpu-reps 1 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('240 steps:', round(d['value'],1), d['parity'])" >> $O/rates.txt
timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight:', round(d['value'],1))" >> $O/rates.txt
cat $O/rates.txt
