# round 5, step 50: rows from which a level runs its 3x3x3 convolutions mask-sorted (CV_MASKED_MIN_ROWS; 16384 = ts1 + ts2 so far)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s50
mkdir -p $O
for mr in 16384 8192 4096 2048 16384 8192 4096; do
  CV_MASKED_MIN_ROWS=$mr timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('masked min rows $mr: 240 steps', round(d['value'],1), '| net one in flight', round(d['stage_ms_isolated']['net'],3), 'ms')" >> $O/masked_min_rows.txt
done
for mr in 16384 8192; do
  CV_MASKED_MIN_ROWS=$mr timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('masked min rows $mr: one in flight', round(d['value'],1))" >> $O/masked_min_rows.txt
  CV_MASKED_MIN_ROWS=$mr python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('masked min rows $mr: 20 steps', round(d['value'],1))" >> $O/masked_min_rows.txt
done
cat $O/masked_min_rows.txt
