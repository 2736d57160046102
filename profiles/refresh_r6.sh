# regenerates the round-6 artefacts in one gpurun call (copy gpurun_out/r6final/* to profiles/r6/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6final
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# the driver's exact command, three fresh processes, then the 240-step line with the CPU baseline and parity objects
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>> $O/bench_driver_cmd.err; done
python bench.py --steps 240 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --streams 1 --steps 40 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --large --points 300000 2>/dev/null | tail -1 > $O/bench_streams1_300k.json
python bench.py --mode train --steps 10 --warmup 2 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_train.json
CV_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode train --steps 6 --warmup 2 --measure-traffic 0 2>$O/bench_train_rccl1.err | tail -1 > $O/bench_train_rccl_one_rank.json
CV_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-reps 1 --train-steps 0 --measure-traffic 0 2>$O/bench_eval_rccl1.err | tail -1 > $O/bench_eval_rccl_one_rank.json
python bench.py --points 8000 --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_8k.json
# kernel stats + per-dispatch trace, one scene in flight and the default streams
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times.txt)
(cd /tmp && rm -rf /tmp/p3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p3.log 2>&1; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_default_streams.csv)
# the training step per kernel (weight gradients on their side stream) and where its main stream idles
(cd /tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 3 --cpu-scenes 0 > /tmp/pt.log 2>&1; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv; t=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/train_gaps.py "$t" > $O/train_gaps.txt)
bash profiles/vote_pmc_sq.sh r6final > /dev/null 2>&1
bash profiles/vote_pmc.sh > $O/vote_pmc.log 2>&1; cp gpurun_out/vote_pmc/* $O/ 2>/dev/null
bash profiles/net_traffic_pmc.sh > $O/net_traffic_pmc.txt 2>&1
MICRO_HL=1 bash profiles/conv_pmc.sh > $O/conv_pmc_hd.txt 2>&1
# counters and trace of the seven-in-flight regime
bash profiles/in_flight_counters.sh > $O/in_flight_counters.log 2>&1; cp gpurun_out/r6/in_flight_counters.txt gpurun_out/r6/in_flight_counters.json gpurun_out/r6/if_trace_bench.json $O/ 2>/dev/null
ls -la $O
