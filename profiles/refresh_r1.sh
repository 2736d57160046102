set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r1n
mkdir -p $O
python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --teacher-forced --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_teacher_forced_streams1.json
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --mode train --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 > $O/bench_train_bf16.json
python bench.py --cpu-scenes 0 --dtype bf16 2>/dev/null | tail -1 > $O/bench_bf16.json
python bench.py --streams 1 --cpu-scenes 0 --dtype bf16 2>/dev/null | tail -1 > $O/bench_bf16_streams1.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 --dtype bf16 > /tmp/pt.log 2>&1; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_bf16_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 > /tmp/pf.log 2>&1; f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_f32_kernel_stats.csv)
python profiles/layer_times.py 2>&1 | tail -66 > $O/layer_times.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --cpu-scenes 0 > /tmp/p3.log 2>&1; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_streams6.csv)
ls -la $O
