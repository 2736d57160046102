cd $GRAFT_REPO_ROOT
O=gpurun_out/r3last; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_2.json 2>/dev/null
python bench.py --steps 240 2>/dev/null | tail -1 > $O/bench.json
for f in bench_driver_cmd bench_driver_cmd_2 bench; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['steps'], r['config']['scenes_in_flight_per_gpu'], r['stage_ms_isolated'], round(r['roofline']['isolated_frac'],3), r.get('train_step_ms') and round(r['train_step_ms']['value'],1))"; done
