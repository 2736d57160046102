# round 5, step 61: old one-scene-at-a-time experiments once more with seven scenes in flight
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s61
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1), 'parity', d.get('parity'))" >> $O/old_experiments.txt
  done
}
run "defaults" CV_NOP=1
run "XCD-aware tile numbering (CV_XCD_TILES=1)" CV_XCD_TILES=1
run "256-row conv_hl workgroups, 32 / 64 columns (CV_HL_NW8=3)" CV_HL_NW8=3
run "vote work lists forced (CV_HV_LISTS=2)" CV_HV_LISTS=2
run "mask groups as a chain of launches (CV_GROUP_CHAIN=1)" CV_GROUP_CHAIN=1
cat $O/old_experiments.txt
