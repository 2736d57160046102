# conv_hd with the fragment reads of a unit issued at once and the next unit's requests going out while they travel (HD_EARLY)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s6; mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "conv_hd" 2>&1 | tail -3
run8() { timeout 400 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
for rep in 1 2; do
echo "CV_HD=0: $(CV_HD=0 run8)"
for sh in 0 1 2; do echo "CV_HD=4 CV_HD_SHAPE=$sh: $(CV_HD=4 CV_HD_SHAPE=$sh run8)"; done
done
} 2>&1 | tee $O/hd_early.txt
