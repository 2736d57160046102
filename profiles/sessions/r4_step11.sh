cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s11; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
for rep in 1 2 3 4; do
echo "hl 240: $(CV_HD=0 run --steps 240)"
echo "hd shape1 240: $(CV_HD=4 CV_HD_SHAPE=1 run --steps 240)"
echo "hd shape2 240: $(CV_HD=4 CV_HD_SHAPE=2 run --steps 240)"
echo "hl 20: $(CV_HD=0 run --gpus 1 --steps 20 --warmup 5)"
echo "hd shape1 20: $(CV_HD=4 CV_HD_SHAPE=1 run --gpus 1 --steps 20 --warmup 5)"
echo "hd shape2 20: $(CV_HD=4 CV_HD_SHAPE=2 run --gpus 1 --steps 20 --warmup 5)"
done
} 2>&1 | tee $O/hd_ab_repeated.txt
