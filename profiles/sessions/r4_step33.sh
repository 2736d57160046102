cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4s33; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 --large --points 300000 > /tmp/p5.log 2>&1; f=$(find /tmp/p5 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_300k.csv; t=$(find /tmp/p5 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times_300k.txt)
grep -n "dec_\|hv_\|head_joint" $O/layer_times_300k.txt | head -30
tail -30 $O/layer_times_300k.txt
