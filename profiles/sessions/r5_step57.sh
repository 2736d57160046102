# round 5, step 57: start stagger of the scene threads on the driver's command, with the in-flight sizing of this round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s57
mkdir -p $O
for sg in 400 0 200 800 400 0 200 800; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --stagger-us $sg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger $sg us: 20 steps', round(d['value'],1))" >> $O/stagger.txt
done
for st in 6 8; do for i in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --streams $st 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$st scenes in flight: 20 steps', round(d['value'],1))" >> $O/stagger.txt
done; done
cat $O/stagger.txt
