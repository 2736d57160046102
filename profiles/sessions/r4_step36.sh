cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4s36; mkdir -p $O
python -m pytest tests/test_sparse_gpu.py tests/test_scene_call_gpu.py -q -m gpu -x 2>&1 | tail -2
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 > /tmp/p1.log 2>&1; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times.txt)
grep -n "build_kernel_maps\|conv_stem\|bitmap_set" $O/layer_times.txt | head
for i in 1 2 3; do python3 bench.py --streams 1 --steps 80 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight %.1f' % d['value'])"; done
