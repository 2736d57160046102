# the driver's command N times (default 24) with the per-step host timeline; the timeline of any run below 470 scenes/s is kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/outliers
for i in $(seq 1 ${1:-24}); do
  CV_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>gpurun_out/outliers/trace_$i.txt | tail -1 > gpurun_out/outliers/line_$i.json
  v=$(python3 -c "import json; d=json.loads(open('gpurun_out/outliers/line_$i.json').read()); print(round(d['value'],1), d['step_host_ms'], d.get('device_allocs_in_timed_region'))")
  echo "run $i: $v"
  python3 -c "
import json,os,sys
d=json.loads(open('gpurun_out/outliers/line_$i.json').read())
if d['value'] >= 470:
    os.remove('gpurun_out/outliers/trace_$i.txt'); os.remove('gpurun_out/outliers/line_$i.json')
"
done
