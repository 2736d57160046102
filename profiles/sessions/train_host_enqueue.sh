# host side of the training step: at 3 x 2000 points the kernels are short and the step time is the host's (Python autograd +
# ~1500 launches); at 3 x 80k the step is the GPU's if it is well above that
cd $GRAFT_REPO_ROOT
for pts in 2000 80000; do for ov in 0 2; do CV_BACKWARD_OVERLAP=$ov python3 bench.py --mode train --points $pts --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('points $pts overlap', d['backward_overlap'], round(d['ms_per_step'],2), 'ms/step, host enqueue', round(d['host_enqueue_ms_per_step'],2), 'ms/step')"; done; done
