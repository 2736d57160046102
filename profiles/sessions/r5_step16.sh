# round 5, step 16: the driver's exact command twice, with the wall time of the whole process (side legs: CPU sweep, training side field, in-run PMC traffic)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s15
mkdir -p $O
for i in 9 10; do
  t0=$SECONDS
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $O/t.err | tail -1 > $O/run_$i.json
  dt=$((SECONDS - t0))
  python -c "import sys,json; d=json.load(open('$O/run_$i.json')); print('run $i (the exact command, $dt s of wall for the process):', round(d['value'],1), 'scenes/s, traffic', round(d['roofline']['traffic']), d['roofline']['traffic_source'][:20], '| cpu_baseline', round(d['cpu_baseline']['value'],3), 'cores', d['cpu_baseline']['cores'], '| train', round(d['train_step_ms']['value'],1), 'ms')" >> $O/driver_cmd_10runs.txt
done
tail -3 $O/driver_cmd_10runs.txt
