# round 5, step 39: joint_loss without host waits (sums over all rows times the mask instead of boolean indexing)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s39
mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2))" >> $O/train.txt
done
cat $O/train.txt
(cd /tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 3 --cpu-scenes 0 > /tmp/pt.log 2>&1; t=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/train_gaps.py "$t" > $O/train_gaps.txt)
cat $O/train_gaps.txt
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q 2>&1 | tail -2
