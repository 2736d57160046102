# regenerates the round-3 artefacts of profiles/r3/ in one gpurun call (copy gpurun_out/r3final/* to profiles/r3/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# the driver's exact command, twice (fresh process each), then the 240-step line with the CPU baseline and parity objects
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_2.json 2>> $O/bench_driver_cmd.err
python bench.py --steps 240 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --streams 1 --steps 40 --cpu-scenes 0 --train-steps 0 --large --points 300000 2>/dev/null | tail -1 > $O/bench_streams1_300k.json
python bench.py --mode separate --large --points 300000 --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_separate_300k.json
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --points 8000 --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_8k.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --cpu-scenes 0 --train-steps 0 > /tmp/p3.log 2>&1; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_default_streams.csv)
bash profiles/trace_one.sh r3final --train-steps 0 > /dev/null 2>&1
bash profiles/vote_pmc_sq.sh r3final > /dev/null 2>&1
bash profiles/vote_pmc.sh > $O/vote_pmc.log 2>&1; cp gpurun_out/vote_pmc/* $O/ 2>/dev/null
MICRO_HL=1 bash profiles/conv_pmc.sh > $O/conv_pmc_hl.txt 2>&1
ls -la $O
