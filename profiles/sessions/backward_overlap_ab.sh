# training step (3 x 80k rows): weight gradient on a side stream next to the input gradient (CV_BACKWARD_OVERLAP) vs one stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3train; mkdir -p $O
python3 -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for ov in 0 1 2; do
  echo "CV_BACKWARD_OVERLAP=$ov: $(CV_BACKWARD_OVERLAP=$ov python3 bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step', round(d['value'],1), d['unit'])")"
done; done | tee $O/backward_overlap_ab.txt
