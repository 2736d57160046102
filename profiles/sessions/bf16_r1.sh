# bf16 compute mode: tests, then eval / train bench lines in both modes (run on the GPU box from the repo root)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bf16
mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_train_f32.json
python bench.py --mode train --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 > $O/bench_train_bf16.json
python bench.py --streams 1 --cpu-scenes 0 --steps 60 --dtype bf16 2>/dev/null | tail -1 > $O/bench_eval_bf16_streams1.json
python bench.py --cpu-scenes 0 --dtype bf16 2>/dev/null | tail -1 > $O/bench_eval_bf16.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 --dtype bf16 > /tmp/pt.log 2>&1; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_bf16_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 > /tmp/pf.log 2>&1; f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_f32_kernel_stats.csv)
cat $O/bench_train_f32.json $O/bench_train_bf16.json $O/bench_eval_bf16_streams1.json $O/bench_eval_bf16.json | cut -c1-400
