# round 5, step 54: the ts16 level (below 1024 rows, 256 columns): 32 / 64 / 96 / 128-column workgroups
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s54
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1), '| net one in flight', round(d['stage_ms_isolated']['net'],3))" >> $O/nb_ts16.txt
  done
}
run "ts16: 64 columns" CV_NB_COARSE=2 CV_NB_COARSE_ROWS=1024
run "ts16: 96 columns" CV_NB_COARSE=3 CV_NB_COARSE_ROWS=1024
run "ts16: 128 columns" CV_NB_COARSE=4 CV_NB_COARSE_ROWS=1024
run "ts16 and ts8 (below 4096 rows): 128 columns" CV_NB_COARSE=4 CV_NB_COARSE_ROWS=4096
cat $O/nb_ts16.txt
