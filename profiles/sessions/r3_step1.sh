#!/bin/bash
# round 3 step 1: the driver's exact command on a fresh box, first GPU process of the lease, then repeats / longer runs
O=gpurun_out/r3b; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_1.json 2> $O/err.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 > $O/driver_cmd_2.json 2>> $O/err.txt
python3 bench.py --gpus 1 --steps 240 --warmup 12 --cpu-scenes 0 > $O/b240.json 2>> $O/err.txt
for s in 1 3 6; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --streams $s > $O/s${s}_20.json 2>> $O/err.txt; done
for s in 3 6; do python3 bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 --streams $s > $O/s${s}_240.json 2>> $O/err.txt; done
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
tail -c 600 $O/err.txt
