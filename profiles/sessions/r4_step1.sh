# round 4, first GPU session: the new tests of this round first, then the whole GPU suite, then the driver's command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s1; mkdir -p $O
python -m pytest tests/test_layer_grads_gpu.py tests/test_vote_gpu.py tests/test_concurrency_gpu.py -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_new.log
python -m pytest "tests/test_train_gpu.py::test_rccl_one_rank_ddp_equals_the_plain_step_bit_for_bit" -m gpu -x -q -s 2>&1 | tail -30 > $O/pytest_rccl.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 > $O/bench_driver_cmd_2.json 2>/dev/null
python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_240.json
python bench.py --steps 60 --streams 1 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_streams1.json
tail -5 $O/pytest_new.log $O/pytest_rccl.log $O/pytest_gpu.log
for f in bench_driver_cmd bench_driver_cmd_2 bench_240 bench_streams1; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(r['value'],1), r['steps'], r['config']['scenes_in_flight_per_gpu'], r['stage_ms_isolated'], round(r['roofline']['isolated_frac'],3), r.get('train_step_ms') and round(r['train_step_ms']['value'],1))"; done
