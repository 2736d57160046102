# round 5, step 34: training step after set_materialize_grads(False); launches per step by kernel family
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s34
mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2))" >> $O/train.txt
done
cat $O/train.txt
HL=1 bash profiles/sessions/r5_step26.sh | head -40 > $O/kernels.txt
grep -i "fill\|copy\|total" $O/kernels.txt
