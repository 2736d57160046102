# round 3: per-kernel times of the training step (3 x 80k rows, fp32-level products), 10 timed steps
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3train; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof -o train -- python3 bench.py --mode train --steps 10 --warmup 3 > $O/bench_train_prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/train_kernel_stats.csv; head -25 $O/train_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
