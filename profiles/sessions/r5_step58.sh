# round 5, step 58: the one-scene-at-a-time defaults once more on today's kernels: split target, mask groups
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s58
mkdir -p $O
for st in 512 384 768 1024 512; do
  timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --split-target $st 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight, split target $st:', round(d['value'],1), 'net', round(d['stage_ms']['net'],3))" >> $O/one_in_flight.txt
done
for g in 3 4; do
  CV_NET_MASK_GROUPS=$g timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight, $g mask groups:', round(d['value'],1), 'net', round(d['stage_ms']['net'],3))" >> $O/one_in_flight.txt
done
cat $O/one_in_flight.txt
