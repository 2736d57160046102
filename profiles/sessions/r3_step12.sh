#!/bin/bash
# new defaults (32x32/16-wave vote tiles, queue launch for large grids): full GPU suite, driver command, 300k, separate
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/err.txt
python bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/b240.json 2>> $O/err.txt
python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 > $O/s1_300k.json 2>> $O/err.txt
CV_HV_LISTS=1 python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 > $O/s1_300k_streaming.json 2>> $O/err.txt
python bench.py --mode separate --large --points 300000 --steps 6 --warmup 2 > $O/separate_300k.json 2>> $O/err.txt
tail -c 400 $O/err.txt
for f in driver_cmd b240 s1_300k s1_300k_streaming; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], round(r['roofline']['frac'],3), r['roofline']['isolated_frac'], r.get('train_step_ms') and r['train_step_ms']['value'])"; done
