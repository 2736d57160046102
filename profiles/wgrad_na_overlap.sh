# weight-gradient tile height (input-channel blocks per wave, CV_WGRAD_NA) under the side-stream overlap: smaller tiles hold
# fewer registers and could share CUs with the input gradient
cd $GRAFT_REPO_ROOT
for na in 4 2 1; do for ov in 2 0; do
  echo "CV_WGRAD_NA=$na CV_BACKWARD_OVERLAP=$ov: $(CV_WGRAD_NA=$na CV_BACKWARD_OVERLAP=$ov python3 bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step')")"
done; done
