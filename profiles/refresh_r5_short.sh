# the bench / listing artefacts of refresh_r5.sh once more on the final tree (no PMC passes, no RCCL legs): copy gpurun_out/r5final/* to profiles/r5/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5final
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>> $O/bench_driver_cmd.err; done
python bench.py --steps 240 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --streams 1 --steps 40 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --large --points 300000 2>/dev/null | tail -1 > $O/bench_streams1_300k.json
python bench.py --mode train --steps 10 --warmup 2 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --points 8000 --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_8k.json
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times.txt)
(cd /tmp && rm -rf /tmp/p3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > /tmp/p3.log 2>&1; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_default_streams.csv)
ls -la $O
