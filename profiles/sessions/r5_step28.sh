# round 5, step 28: host side of the training step (enqueue time per step, and the step at 3 x 2000 points where the kernels are short)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s28
mkdir -p $O
for hl in 0 1; do
  for pts in 80000 2000; do
    CV_TRAIN_FWD_HL=$hl timeout 600 python bench.py --mode train --steps 12 --warmup 3 --cpu-scenes 0 --points $pts 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CV_TRAIN_FWD_HL=$hl points $pts: step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2), 'ms')" >> $O/train_host.txt
  done
done
cat $O/train_host.txt
