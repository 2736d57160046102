cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_vote_gpu.py -x -q -m gpu -k "kernel_events" 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > gpurun_out/kev_driver.json
python3 bench.py --streams 1 --steps 40 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > gpurun_out/kev_streams1.json
python3 -c "
import json
for f in ('kev_driver','kev_streams1'):
    r=json.loads(open('gpurun_out/%s.json'%f).read()); ro=r['roofline']
    print(f, round(r['value'],1), ro['kernel'][:14], 'frac', round(ro['frac'],3), 'avg_ms', round(ro['avg_ms'],3), 'op', round(ro['op_avg_ms'],3), round(ro['op_frac'],3), 'iso', ro['isolated_avg_ms'], ro['isolated_frac'], ro['isolated_op_avg_ms'])
"
