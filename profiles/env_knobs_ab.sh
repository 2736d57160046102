# runtime knobs of the HIP runtime on the scene rate (240 steps, default seven in flight; 20-step driver command; one in flight)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O; : > $O/env_knobs.txt
run() { tag="$1"; shift; env "$@" python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > /tmp/a.json
  env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > /tmp/b.json
  env "$@" python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > /tmp/c.json
  python -c "
import json
a,b,c=[json.load(open('/tmp/%s.json'%k))['value'] for k in 'abc']
print('%-40s 240 steps %.1f | 20 steps %.1f | one in flight %.1f scenes/s' % ('$tag', a, b, c))" >> $O/env_knobs.txt; tail -1 $O/env_knobs.txt; }
run "default" CV_NOP=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "GPU_MAX_HW_QUEUES=5" GPU_MAX_HW_QUEUES=5
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "default again" CV_NOP=1
