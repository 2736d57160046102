#!/bin/bash
# new vote default (16 x 32 / 8 waves, streaming below 128 tiles): full GPU suite, 80k / 300k stage times, rocprof of the tile kernel
O=gpurun_out/r3aj; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/s1.json 2> $O/err.txt
python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 > $O/s1_300k.json 2>> $O/err.txt
CV_HV_LISTS=1 python bench.py --streams 1 --steps 30 --warmup 5 --cpu-scenes 0 --train-steps 0 --large --points 300000 > $O/s1_300k_streaming.json 2>> $O/err.txt
for f in s1 s1_300k s1_300k_streaming; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], round(r['roofline']['frac'],3))"; done
bash profiles/decode_prof.sh r3aj > /dev/null 2>&1; grep hv_fwd $O/decode_kernels.txt
