#!/bin/bash
# conv_hl after the occupancy attribute fix (96-column kernel back at 128 VGPRs = four workgroups per CU) and the one-slot
# variant (CV_HL_NS1 bit mask per column width: no prefetch, 68 / 78 / 96 VGPRs = seven / six / five workgroups per CU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
CV_HL_NS1=7 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q -k "not training and not gradients" > $O/pytest_ns1.log 2>&1; tail -2 $O/pytest_ns1.log
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for ns1 in 0 7 4 3 1; do
  echo "CV_HL_NS1=$ns1: net $(CV_HL_NS1=$ns1 one) $(CV_HL_NS1=$ns1 one) | six in flight $(CV_HL_NS1=$ns1 six) $(CV_HL_NS1=$ns1 six)" | tee -a $O/hl_ns1_ab.txt
done
