# level-0 3x3x3 map + mask orders on a side stream (CV_PLAN_SIDE 0 never / 1 when the scene starts alone / 2 always)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s34; mkdir -p $O
python -m pytest tests/test_scene_call_gpu.py tests/test_concurrency_gpu.py -q -m gpu -x 2>&1 | tail -2
CV_PLAN_SIDE=2 python -m pytest tests/test_scene_call_gpu.py -q -m gpu -x 2>&1 | tail -2
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], end=' ')"; }
for m in 0 1 2 0 1 2; do
  echo -n "CV_PLAN_SIDE=$m : one in flight " >> $O/plan_side.txt
  CV_PLAN_SIDE=$m python3 bench.py --streams 1 --steps 80 --cpu-scenes 0 --train-steps 0 2>/dev/null | val >> $O/plan_side.txt
  echo -n " | 240 steps " >> $O/plan_side.txt
  CV_PLAN_SIDE=$m python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | val >> $O/plan_side.txt
  echo -n " | 20 steps " >> $O/plan_side.txt
  for i in 1 2; do CV_PLAN_SIDE=$m python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | val >> $O/plan_side.txt; done
  echo >> $O/plan_side.txt
done
cat $O/plan_side.txt
