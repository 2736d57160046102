# round 5, step 33: mask groups of the generic (training) path with the hl kernels underneath: 4 (default so far) against 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s33
mkdir -p $O
for g in 4 3 4 3 2; do
  CV_MASK_GROUPS=$g timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CV_MASK_GROUPS=$g train step', round(d['ms_per_step'],2), 'ms')" >> $O/mask_groups.txt
done
cat $O/mask_groups.txt
