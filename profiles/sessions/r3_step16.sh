#!/bin/bash
# vote tile kernel with the one-load prologue and the polynomial arc atan2: exactness tests, tick profile, op time, bench
O=gpurun_out/r3t; mkdir -p $O
python -m pytest tests/test_vote_gpu.py tests/test_production_size_gpu.py tests/test_concurrency_gpu.py tests/test_cabi.py -m gpu -x -q -s > $O/pytest.log 2>&1; grep -E "3 x 20k|passed|failed|same ReLU" $O/pytest.log
python profiles/vote_time.py --ticks > $O/vote_time.txt 2>&1; grep -v amdgpu.ids $O/vote_time.txt
python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/s1.json 2> $O/err.txt
python bench.py --gpus 1 --steps 120 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/b120.json 2>> $O/err.txt
for f in s1 b120; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], r['roofline']['isolated_frac'])"; done
