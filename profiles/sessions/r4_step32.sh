# stability of the final tree: ten fresh processes of the driver's command, one long run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s32; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f parity %s allocs %s' % (d['value'], all(v is True for k,v in d['parity'].items() if isinstance(v,bool)), d.get('device_allocs_in_timed_region')))" >> $O/driver_cmd_10runs.txt
done
python3 bench.py --steps 4000 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('4000 steps: %.1f scenes/s, %.3f ms/step, step_host_ms %s, allocs %s' % (d['value'], d['ms_per_step'], d.get('step_host_ms'), d.get('device_allocs_in_timed_region')))" >> $O/driver_cmd_10runs.txt
cat $O/driver_cmd_10runs.txt
