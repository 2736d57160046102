# round 5, step 36: the training forward on the spatially sorted twin of the coordinate set (CV_TRAIN_SORTED=1) with the hl kernels underneath
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s36
mkdir -p $O
for srt in 0 1 0 1; do
  CV_TRAIN_SORTED=$srt timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CV_TRAIN_SORTED=$srt train step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2))" >> $O/sorted.txt
done
cat $O/sorted.txt
