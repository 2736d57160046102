# round 5, step 38: where the main stream of a training step idles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s38
mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_gpu.py -m gpu -q -x 2>&1 | tail -1
(cd /tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 3 --cpu-scenes 0 > /tmp/pt.log 2>&1; t=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/train_gaps.py "$t" > $O/train_gaps.txt)
cat $O/train_gaps.txt
