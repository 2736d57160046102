# round 5, step 27: the training forward on the hl-format kernels (CV_TRAIN_FWD_HL, hl twins written by the BatchNorm apply pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s27
mkdir -p $O
for hl in 0 1 0 1; do
  CV_TRAIN_FWD_HL=$hl timeout 600 python bench.py --mode train --steps 12 --warmup 3 --cpu-scenes 0 2>$O/err_$hl.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CV_TRAIN_FWD_HL=$hl train step', round(d['ms_per_step'],2), 'ms', 'fallbacks', d.get('train_range_fallbacks'))" >> $O/train_hl.txt
done
cat $O/train_hl.txt; tail -5 $O/err_1.txt
true

