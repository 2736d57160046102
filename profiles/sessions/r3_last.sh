cd $GRAFT_REPO_ROOT
O=gpurun_out/r3last2; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do echo "default (8 in flight, 400 us stagger) 20 steps: $(python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")"; done | tee $O/driver_cmd_8runs.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_2.json 2>/dev/null
python bench.py --steps 240 2>/dev/null | tail -1 > $O/bench.json
for f in bench_driver_cmd bench_driver_cmd_2 bench; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['steps'], r['config']['scenes_in_flight_per_gpu'], r['stage_ms_isolated'], round(r['roofline']['isolated_frac'],3), r.get('train_step_ms') and round(r['train_step_ms']['value'],1), r['parity']['net_within_1e-4'])"; done
