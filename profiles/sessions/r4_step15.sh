cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s15; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 --scene-call py "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
for rep in 1 2; do
echo "full: $(run --steps 240)"
echo "no plan: $(run --steps 240 --ablate noplan)"
echo "no plan, no vote/decode: $(run --steps 240 --ablate noplan,novote)"
echo "no plan, no vote/decode, no finish: $(run --steps 240 --ablate noplan,novote,finish)"
done
} 2>&1 | tee $O/plan_ablation.txt
