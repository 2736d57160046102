# round 5, step 11: the whole GPU suite after the ABI / bench changes, the default bench line (CPU thread sweep, in-run PMC traffic)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s11
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.err
python - <<P
import json
d=json.load(open('$O/bench_driver_cmd.json'))
print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'][:160])
print(json.dumps(d['cpu_baseline'])[:900])
print(d['collective'], d['train_step_ms'])
P
