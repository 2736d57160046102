cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s29; mkdir -p $O
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
nproc >> $O/spin.txt
for sp in 0 20000 0 20000; do
  echo -n "CV_SPIN_WAIT_US=$sp one in flight: " >> $O/spin.txt
  CV_SPIN_WAIT_US=$sp python3 bench.py --streams 1 --steps 80 --cpu-scenes 0 --train-steps 0 2>/dev/null | val >> $O/spin.txt
  echo -n " | 20-step command: " >> $O/spin.txt
  for i in 1 2; do CV_SPIN_WAIT_US=$sp python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | val >> $O/spin.txt; done
  echo >> $O/spin.txt
done
cat $O/spin.txt
