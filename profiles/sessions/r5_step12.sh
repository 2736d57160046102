# round 5, step 12: neighbour windows on the ts2 level only (CV_WIN_LEVELS=2) against both fine levels (3) and off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s12
mkdir -p $O
: > $O/win_levels.txt
for cfg in "0 31" "1 2" "1 3" "0 31" "1 2"; do
  set -- $cfg
  v1=$(CV_WIN=$1 CV_WIN_LEVELS=$2 python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms_isolated']['net'])")
  v2=$(CV_WIN=$1 CV_WIN_LEVELS=$2 python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "win=$1 levels=$2: 240 steps (7 in flight) $v1 | one in flight $v2" >> $O/win_levels.txt
done
cat $O/win_levels.txt
