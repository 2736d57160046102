# round 5, step 15: the window tests on the final tree, then the driver's command ten times (fresh processes) for its spread
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s15
mkdir -p $O
timeout 900 python -m pytest tests/test_windows_gpu.py tests/test_sparse_gpu.py::test_zskip_leaves_every_bit_in_place -x -q > $O/pytest_windows.log 2>&1; tail -3 $O/pytest_windows.log
: > $O/driver_cmd_10runs.txt
for i in 1 2 3 4 5 6 7 8; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i (side legs off):', round(d['value'],1), 'scenes/s, ms_per_step', round(d['ms_per_step'],3), 'step_host_ms', d['step_host_ms'])" >> $O/driver_cmd_10runs.txt
done
for i in 9 10; do
  /usr/bin/time -f "wall %e s" python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $O/t.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i (the exact command):', round(d['value'],1), 'scenes/s, traffic', d['roofline']['traffic'], 'cpu_baseline', d['cpu_baseline']['value'], 'cores', d['cpu_baseline']['cores'])" >> $O/driver_cmd_10runs.txt
  tail -1 $O/t.err >> $O/driver_cmd_10runs.txt
done
cat $O/driver_cmd_10runs.txt
