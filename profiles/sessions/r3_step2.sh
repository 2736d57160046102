#!/bin/bash
# round 3 step 2: coordinate plan in fewer launches - parity tests, one-scene trace, bench
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/pytest_sparse.log 2>&1; tail -3 $O/pytest_sparse.log
bash profiles/trace_one.sh r3c > $O/trace.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 > $O/b20.json 2> $O/err.txt
python3 bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 > $O/b240.json 2>> $O/err.txt
python3 bench.py --gpus 1 --steps 60 --warmup 5 --cpu-scenes 0 --streams 1 > $O/s1.json 2>> $O/err.txt
tail -c 400 $O/err.txt
