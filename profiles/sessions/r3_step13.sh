#!/bin/bash
# tree at d214f9f: full GPU suite, the driver's command twice (fresh process each), 240-step line
O=gpurun_out/r3n; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/err.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd2.json 2>> $O/err.txt
python bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/b240.json 2>> $O/err.txt
tail -c 400 $O/err.txt
for f in driver_cmd driver_cmd2 b240; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], round(r['roofline']['frac'],3), r['roofline']['isolated_frac'], r.get('train_step_ms') and r['train_step_ms']['value'])"; done
