cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s9; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']; s=d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']), 'in-region net %.2f vote %.2f decode %.2f' % (s['net'], s['vote'], s['decode']), 'host median %.2f' % (d['step_host_ms']['median']))"; }
{
for rep in 1 2; do
echo "py: $(run --scene-call py --steps 240)"
echo "c: $(run --scene-call c --steps 240)"
echo "c grids from torch: $(CV_SCENE_GRIDS=torch run --scene-call c --steps 240)"
done
} 2>&1 | tee $O/scene_call_probe2.txt
tail -3 $O/err.txt
