#!/bin/bash
# rolling two-plane vote kernel: exactness tests, op time, bench one / six in flight, old kernel for comparison
O=gpurun_out/r3x; mkdir -p $O
python -m pytest tests/test_vote_gpu.py tests/test_concurrency_gpu.py tests/test_cabi.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -m pytest tests/test_production_size_gpu.py tests/test_decode_gpu.py -m gpu -x -q -k "not training" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
python profiles/vote_time.py 2>&1 | grep -v amdgpu.ids
CV_HV_ROLL=0 python profiles/vote_time.py 2>&1 | grep -v amdgpu.ids
python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/s1.json 2> $O/err.txt
python bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/b240.json 2>> $O/err.txt
for f in s1 b240; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], r['roofline']['isolated_frac'], r.get('parity'))"; done
