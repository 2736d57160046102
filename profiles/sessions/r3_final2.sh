# last validation of round 3 on the final tree: GPU suite, smoke, the driver's command twice, the default 240-step line, training line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final2; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_2.json 2>> $O/bench_driver_cmd.err
python3 bench.py --steps 240 2>/dev/null | tail -1 > $O/bench.json
python3 bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_train.json
for f in bench_driver_cmd bench_driver_cmd_2 bench; do python3 -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['steps'], r['config']['scenes_in_flight_per_gpu'], r['config']['conv_split_target'], r['stage_ms_isolated'], round(r['roofline']['isolated_frac'],3), r.get('train_step_ms') and round(r['train_step_ms']['value'],1), r['parity'] and r['parity']['net_within_1e-4'])"; done
python3 -c "
import json
r=json.loads(open('$O/bench_train.json').read()); print('train', round(r['ms_per_step'],2), r['backward_overlap'])"
