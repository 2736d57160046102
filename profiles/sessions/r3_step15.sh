#!/bin/bash
# gradient test with the fp64 yardstick; per-dispatch trace of one scene in flight (default tree); vote tick profile
O=gpurun_out/r3r; mkdir -p $O
python -m pytest tests/test_production_size_gpu.py -m gpu -x -q -k "training" -s > $O/pytest.log 2>&1; grep -E "3 x 20k|passed|failed" $O/pytest.log
bash profiles/trace_one.sh r3r/trace --train-steps 0 > $O/trace.log 2>&1
tail -3 $O/trace.log
for a in 2 24 21 22 23 25; do python profiles/vote_time.py --algo $a --teacher 2>&1 | tail -12; done > $O/vote_ablate.txt 2>&1
cat $O/vote_ablate.txt
