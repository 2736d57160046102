cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "split_target or fused_and_modular or scenes_in_flight" 2>&1 | tail -3
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver cmd', round(d['value'],1), 'net iso', round(d['stage_ms_isolated']['net'],3), d['config']['conv_split_target'], d.get('parity'))"; done
for st in 512 256; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --split-target $st 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split-target $st: 20 steps', round(d['value'],1))"; done
python3 bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('240 steps', round(d['value'],1))"
