# round 5, step 56: more knobs that were chosen one scene at a time: conv_hl unit slots, the training path's mask threshold
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s56
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1), '| net one in flight', round(d['stage_ms_isolated']['net'],3))" >> $O/knobs3.txt
  done
}
run "defaults (conv_hl: two unit slots)" CV_NOP=1
run "conv_hl: three unit slots (CV_HL_NS2=0)" CV_HL_NS2=0
run "conv_hl: one unit slot (CV_HL_NS1=7)" CV_HL_NS1=7
run "conv_hl: two slots for 32 / 64 columns, three for 96 (CV_HL_NS2=3)" CV_HL_NS2=3
for mr in 16384 8192 4096; do
  CV_AUTO_MASK_MIN_ROWS=$mr timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('training, mask-sorted from $mr rows: step', round(d['ms_per_step'],2), 'ms')" >> $O/knobs3.txt
done
cat $O/knobs3.txt
