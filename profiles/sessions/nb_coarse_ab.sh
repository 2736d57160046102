#!/bin/bash
# 128-column conv_hl workgroups on the coarse levels (CV_NB_COARSE=4 below CV_NB_COARSE_ROWS rows): parity, net one in flight, six in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
CV_NB_COARSE=4 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q -k "not training" > $O/pytest_nb4.log 2>&1; tail -2 $O/pytest_nb4.log
one() { timeout 300 python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_median']['net'],3))"; }
six() { timeout 300 python bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for cfg in "CV_NB_COARSE=0" "CV_NB_COARSE=4 CV_NB_COARSE_ROWS=4096" "CV_NB_COARSE=4 CV_NB_COARSE_ROWS=16384" "CV_NB_COARSE=4 CV_NB_COARSE_ROWS=40000" "CV_NB_COARSE=4 CV_NB_COARSE_ROWS=16384 CV_SPLIT_TARGET=768"; do
  echo "$cfg: net $(env $cfg bash -c "$(declare -f one); one") $(env $cfg bash -c "$(declare -f one); one") | six in flight $(env $cfg bash -c "$(declare -f six); six") $(env $cfg bash -c "$(declare -f six); six")" | tee -a $O/nb_coarse_ab.txt
done
