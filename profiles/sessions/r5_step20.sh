# round 5, step 20: kernel trace of one scene with the in-launch group sum (gfuse) against the finish launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s20
mkdir -p $O
for g in 0 1; do
  (cd /tmp && rm -rf /tmp/p$g && CV_GFUSE=$g rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$g -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p$g.log 2>&1; t=$(find /tmp/p$g -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times_gfuse$g.txt)
done
sed -n 105,140p $O/layer_times_gfuse1.txt
