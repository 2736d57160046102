# round 5, step 25: weight-gradient tile heights (CV_WGRAD_NA: input-channel blocks per wave -> accumulator registers) with the
# weight gradients on their side stream: does a slimmer wgrad wave share CUs with the input gradient?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s25
mkdir -p $O
for na in 4 2 1 3 4; do
  CV_WGRAD_NA=$na timeout 600 python bench.py --mode train --steps 12 --warmup 3 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CV_WGRAD_NA=$na train step', round(d['ms_per_step'],2), 'ms')" >> $O/wgrad_na.txt
done
cat $O/wgrad_na.txt
