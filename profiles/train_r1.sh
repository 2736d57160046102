# training path: tests, then bench lines (f32 / bf16) and rocprofv3 kernel summaries (run on the GPU box from the repo root)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/train
mkdir -p $O
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_bf16_gpu.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for dt in f32 bf16; do
  python bench.py --mode train --steps 10 --warmup 2 --dtype $dt 2>/dev/null | tail -1 > $O/bench_train_${dt}.json
done
CV_WGRAD_NA=1 python bench.py --mode train --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 > $O/bench_train_bf16_na1.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 --dtype bf16 > /tmp/pt.log 2>&1; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_bf16_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 > /tmp/pf.log 2>&1; f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_f32_kernel_stats.csv)
for f in $O/bench_train_*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f; done
