# round 5, step 60: conv_hd reach with the ts4 level mask-sorted (hd_mask bits, hd_min_rows) on the in-flight defaults
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s60
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1))" >> $O/hd_reach.txt
  done
}
run "defaults (96-column launches from 16384 rows on conv_hd)" CV_NOP=1
run "64- and 96-column launches from 8192 rows (CV_HD=6 CV_HD_MIN_ROWS=8192)" CV_HD=6 CV_HD_MIN_ROWS=8192
run "96-column launches from 8192 rows (CV_HD=4 CV_HD_MIN_ROWS=8192)" CV_HD=4 CV_HD_MIN_ROWS=8192
run "32-, 64-, 96-column launches from 16384 rows (CV_HD=7)" CV_HD=7
cat $O/hd_reach.txt
