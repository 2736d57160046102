cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s10; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']; s=d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']), 'in-region net %.2f vote %.2f decode %.2f' % (s['net'], s['vote'], s['decode']), 'host median %.2f' % (d['step_host_ms']['median']))"; }
{
for q in 4 8 16; do for sc in py c; do
echo "GPU_MAX_HW_QUEUES=$q $sc 240: $(GPU_MAX_HW_QUEUES=$q run --scene-call $sc --steps 240)"
echo "GPU_MAX_HW_QUEUES=$q $sc 240: $(GPU_MAX_HW_QUEUES=$q run --scene-call $sc --steps 240)"
echo "GPU_MAX_HW_QUEUES=$q $sc 20: $(GPU_MAX_HW_QUEUES=$q run --scene-call $sc --gpus 1 --steps 20 --warmup 5)"
echo "GPU_MAX_HW_QUEUES=$q $sc 20: $(GPU_MAX_HW_QUEUES=$q run --scene-call $sc --gpus 1 --steps 20 --warmup 5)"
done; done
} 2>&1 | tee $O/hwq_scene_call.txt
