#!/bin/bash
# conv_hl with the line-coalesced gather (HL_COAL=1): parity of the network tests, net time one scene in flight, headline
O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q -k "not training" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/s1.json 2> $O/err.txt
python bench.py --gpus 1 --steps 120 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/b120.json 2>> $O/err.txt
tail -c 300 $O/err.txt
for f in s1 b120; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['stage_ms_isolated'] or r['stage_ms_median'], r.get('parity'))"; done
