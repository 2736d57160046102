# round 5, step 14: rocprofv3 kernel stats + per-launch listing of one scene with the neighbour windows ON (conv_win v5), and its SQ counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s14
mkdir -p $O
(cd /tmp && rm -rf /tmp/p1 && CV_WIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/win_v5_full_path_kernel_stats.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/win_v5_layer_times.txt)
grep "conv_win\|build_windows\|launches" $O/win_v5_layer_times.txt | head -24
bash profiles/win_pmc.sh 80000 2 > $O/win_v5_pmc_ts2.txt 2>&1
bash profiles/win_pmc.sh 80000 1 > $O/win_v5_pmc_ts1.txt 2>&1
cat $O/win_v5_pmc_ts2.txt | cut -c1-250
