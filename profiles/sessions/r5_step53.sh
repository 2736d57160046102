# round 5, step 53: column blocks per workgroup of the coarse levels (chosen one scene at a time in rounds 1-3) with seven scenes in flight
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s53
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1), '| net one in flight', round(d['stage_ms_isolated']['net'],3))" >> $O/nb.txt
  done
}
run "defaults (64-column workgroups at 128 / 256 columns, 32 below 1024 rows)" CV_NOP=1
run "128-column workgroups (CV_NB_WIDE=4)" CV_NB_WIDE=4
run "64 columns on the ts16 level too (CV_NB_COARSE=2 below 1024 rows)" CV_NB_COARSE=2 CV_NB_COARSE_ROWS=1024
run "128 columns on every coarse level (CV_NB_COARSE=4)" CV_NB_COARSE=4
run "96 columns on every coarse level (CV_NB_COARSE=3)" CV_NB_COARSE=3
cat $O/nb.txt
