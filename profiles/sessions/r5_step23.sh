# round 5, step 23: gfuse on its own kernel instance (the default conv_hd keeps its registers): tests, rates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s23
mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_cabi.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
cat $O/pytest.txt
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label 240 steps:', round(d['value'],1))" >> $O/rates.txt
  env "$@" timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label one in flight:', round(d['value'],1))" >> $O/rates.txt
}
run "finish launches" CV_GFUSE=0
run "gfuse" CV_GFUSE=1
run "finish launches" CV_GFUSE=0
run "gfuse" CV_GFUSE=1
cat $O/rates.txt
