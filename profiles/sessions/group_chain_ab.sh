#!/bin/bash
# chained mask-group launches (CV_GROUP_CHAIN=1) against one launch + finish: parity, net one in flight, scenes/s six in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
CV_GROUP_CHAIN=1 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q -k "not training" > $O/pytest_chain.log 2>&1; tail -2 $O/pytest_chain.log
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms_median']['net'],3), d.get('parity'))"; }
six() { timeout 300 python bench.py --steps 120 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for chain in 0 1; do
  for groups in 3 4; do
    echo "CV_GROUP_CHAIN=$chain CV_NET_MASK_GROUPS=$groups: one in flight $(CV_GROUP_CHAIN=$chain CV_NET_MASK_GROUPS=$groups one) | $(CV_GROUP_CHAIN=$chain CV_NET_MASK_GROUPS=$groups one) ; six in flight $(CV_GROUP_CHAIN=$chain CV_NET_MASK_GROUPS=$groups six)" | tee -a $O/group_chain_ab.txt
  done
done
