cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s8; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']; s=d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']), 'in-region net %.2f vote %.2f decode %.2f' % (s['net'], s['vote'], s['decode']), 'host median %.2f' % (d['step_host_ms']['median']))"; }
{
for rep in 1 2; do
echo "py: $(run --scene-call py --steps 240)"
echo "c: $(run --scene-call c --steps 240)"
echo "c serialized enqueue: $(CV_SCENE_SERIALIZE=1 run --scene-call c --steps 240)"
done
for st in 4 6 10 12; do echo "c streams $st: $(run --scene-call c --steps 240 --streams $st)"; done
echo "c stagger 800: $(run --scene-call c --steps 240 --stagger-us 800)"
echo "c HIP_FORCE... GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 run --scene-call c --steps 240)"
echo "py GPU_MAX_HW_QUEUES=8: $(GPU_MAX_HW_QUEUES=8 run --scene-call py --steps 240)"
} 2>&1 | tee $O/scene_call_probe.txt
