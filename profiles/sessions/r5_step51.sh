# round 5, step 51: bench defaults with the two in-flight policies (vote part records 12288, masked min rows 8192 from four scenes in flight)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s51
mkdir -p $O
for i in 1 2 3; do
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 1 --cpu-reps 1 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('240 steps', round(d['value'],1), '| in-region frac', round(r['frac'],3), 'isolated frac', round(r['isolated_frac'],3), '| net isolated', round(d['stage_ms_isolated']['net'],3), d['config'].get('masked_min_rows'), d['config'].get('vote_part_records'), 'parity exact', all(v for k,v in d['parity'].items() if k.endswith('exact')), d['parity']['net_max_abs_err'])" >> $O/defaults.txt
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 steps', round(d['value'],1))" >> $O/defaults.txt
done
timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one in flight', round(d['value'],1), d['config'].get('masked_min_rows'), d['config'].get('vote_part_records'))" >> $O/defaults.txt
cat $O/defaults.txt
timeout 900 python -m pytest tests/test_bench_rccl_gpu.py tests/test_scene_call_gpu.py -m gpu -q -x 2>&1 | tail -1
