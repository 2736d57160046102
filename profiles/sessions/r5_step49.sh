# round 5, step 49: knobs that were tuned one scene at a time (or inside the noise), re-run with seven in flight on today's defaults
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s49
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1))" >> $O/knobs.txt
  done
}
run "defaults" CV_NOP=1
run "masked min rows 8192 (ts4 mask-sorted too)" CV_MASKED_MIN_ROWS=8192
run "partial-tile traffic cap 12 MB" CV_SPLIT_TRAFFIC_MB=12
run "partial-tile traffic cap 48 MB" CV_SPLIT_TRAFFIC_MB=48

cat $O/knobs.txt
for st in 192 320; do for i in 1 2; do timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --split-target $st 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split target $st: 240 steps', round(d['value'],1))" >> $O/knobs.txt; done; done
cat $O/knobs.txt
