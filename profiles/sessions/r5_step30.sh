# round 5, step 30: the input gradients on the hl-format kernels too (CV_TRAIN_BWD_HL)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s30
mkdir -p $O
for cfg in "0 0" "1 0" "1 1" "0 0" "1 1"; do
  set -- $cfg
  CV_TRAIN_FWD_HL=$1 CV_TRAIN_BWD_HL=$2 timeout 600 python bench.py --mode train --steps 12 --warmup 3 --cpu-scenes 0 2>$O/err_$1$2.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FWD_HL=$1 BWD_HL=$2 train step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2))" >> $O/train_hl.txt
done
cat $O/train_hl.txt; tail -3 $O/err_11.txt
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
