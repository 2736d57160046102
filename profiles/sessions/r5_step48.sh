# round 5, step 48: the vote's records per part as a run-time knob (cv_hv_set_part_records): values with seven scenes in flight, repeats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s48
mkdir -p $O
timeout 900 python -m pytest tests/test_vote_gpu.py tests/test_scene_call_gpu.py -m gpu -q -x 2>&1 | tail -1
for pr in 4096 8192 12288 16384 4096 8192 12288; do
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --vote-part-records $pr 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('part records $pr: 240 steps', round(d['value'],1), '| in-region frac', round(r['frac'],3), 'isolated frac', round(r['isolated_frac'],3))" >> $O/vote_part_records.txt
done
for pr in 4096 8192 4096 8192; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 --vote-part-records $pr 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('part records $pr: 20 steps', round(d['value'],1))" >> $O/vote_part_records.txt
done
cat $O/vote_part_records.txt
