# round 5, step 52: scenes in flight and mask groups once more on the new in-flight defaults
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s52
mkdir -p $O
run() {  # label, args...
  label=$1; shift
  for i in 1 2; do
    env $ENVV timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: 240 steps', round(d['value'],1))" >> $O/knobs2.txt
  done
}
ENVV="CV_NOP=1" run "defaults (7 scenes in flight)"
ENVV="CV_NOP=1" run "6 in flight" --streams 6
ENVV="CV_NOP=1" run "8 in flight" --streams 8
ENVV="CV_NOP=1" run "10 in flight" --streams 10
ENVV="CV_NET_MASK_GROUPS=4" run "4 mask groups"
ENVV="CV_NOP=1" run "vote part records 16384" --vote-part-records 16384
ENVV="CV_NOP=1" run "masked min rows 4096" --masked-min-rows 4096
cat $O/knobs2.txt
