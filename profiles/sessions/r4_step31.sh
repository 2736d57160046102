cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s31; mkdir -p $O
python -m pytest tests/test_sparse_gpu.py tests/test_scene_call_gpu.py tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -3
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 > /tmp/p1.log 2>&1; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $GRAFT_REPO_ROOT/$O/layer_times.txt)
head -12 $O/layer_times.txt
python3 bench.py --streams 1 --steps 80 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight %.1f' % d['value'])"
