# conv_finish change: parity tests, layer times, eval bench (one / six scenes in flight), train bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/finish
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_train_gpu.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python profiles/layer_times.py 2>&1 | tail -66 > $O/layer_times.txt; tail -1 $O/layer_times.txt
python bench.py --streams 1 --cpu-scenes 0 --steps 120 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --mode train --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 > $O/bench_train_bf16.json
for f in bench_streams1 bench bench_train bench_train_bf16; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d.get('stage_ms'))"; done
