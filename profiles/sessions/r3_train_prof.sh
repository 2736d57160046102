# round 3: per-kernel times of the training step (3 x 80k rows, fp32-level products), 10 timed steps, one stream
# (CV_BACKWARD_OVERLAP=0: kernel durations without neighbours) and the default overlap
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3train; mkdir -p $O
for ov in 0 2; do
  (cd /tmp && CV_BACKWARD_OVERLAP=$ov rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt$ov -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 > $O/bench_train_prof_$ov.log 2>&1
   f=$(find /tmp/pt$ov -name '*kernel_stats.csv' | head -1); cp "$f" $O/train_kernel_stats_overlap$ov.csv)
  grep '"metric"' $O/bench_train_prof_$ov.log | tail -1 > $O/bench_train_under_rocprof_overlap$ov.json
done
head -22 $O/train_kernel_stats_overlap0.csv | cut -c1-150
