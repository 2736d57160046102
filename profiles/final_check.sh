cd $GRAFT_REPO_ROOT
O=gpurun_out/r3fin2; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -2 $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null; python -c "
import json
r=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1]); print('driver cmd', round(r['value'],1), r['stage_ms_isolated'], r['roofline']['isolated_frac'])"
