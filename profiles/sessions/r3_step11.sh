#!/bin/bash
# full GPU suite + the new bench modes on the restored streaming launch (80k) / queue launch (300k)
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/err.txt
python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 > $O/s1_300k.json 2>> $O/err.txt
python bench.py --mode separate --large --points 300000 --steps 6 --warmup 2 > $O/separate_300k.json 2>> $O/err.txt
python bench.py --mode separate --steps 10 --warmup 2 > $O/separate_80k.json 2>> $O/err.txt
tail -c 600 $O/err.txt
for f in driver_cmd s1_300k; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], round(r['roofline']['frac'],3), r['roofline']['isolated_frac'], r.get('train_step_ms'))"; done
cat $O/separate_300k.json | cut -c1-1500; cat $O/separate_80k.json | cut -c1-1500
