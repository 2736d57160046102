# round 5, step 3: conv_win v3 (in-wave software pipeline): tests, bench with windows on, per-launch trace
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5s3}
mkdir -p $O
timeout 900 python -m pytest tests/test_windows_gpu.py -x -q > $O/pytest_windows.log 2>&1; tail -5 $O/pytest_windows.log
CV_WIN=1 timeout 300 python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 2>$O/bench_s1_win1.err | tail -1 > $O/bench_s1_win1.json
CV_WIN=1 timeout 300 python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>$O/bench_240_win1.err | tail -1 > $O/bench_240_win1.json
CV_WIN=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_20_win1.json
python - <<P
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d['value'],1), d.get('stage_ms_isolated'), (d.get('parity') or {}).get('net_max_abs_err'))
    except Exception as e: print(f, 'ERR', e, open(f).read()[:300])
P
(cd /tmp && rm -rf /tmp/p1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/layer_trace.py "$t" > $O/layer_times.txt)
grep "conv_win\|build_windows" $O/layer_times.txt | head -20
if [ "$2" = "full" ]; then timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py tests/test_scene_call_gpu.py tests/test_concurrency_gpu.py -x -q > $O/pytest_net.log 2>&1; tail -4 $O/pytest_net.log; fi
