# conv_hd shapes: 8 waves x 3 stages (one workgroup per CU), 4 x 2 (two per CU), 8 x 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s5; mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "conv_hd" 2>&1 | tail -3
run8() { timeout 400 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
echo "CV_HD=0: $(CV_HD=0 run8)"
for sh in 0 1 2; do for hd in 7 4; do echo "CV_HD=$hd CV_HD_SHAPE=$sh: $(CV_HD=$hd CV_HD_SHAPE=$sh run8)"; done; done
echo "CV_HD=7 shape 1 rows>=4096: $(CV_HD=7 CV_HD_SHAPE=1 CV_HD_MIN_ROWS=4096 run8)"
echo "CV_HD=7 shape 1 all rows: $(CV_HD=7 CV_HD_SHAPE=1 CV_HD_MIN_ROWS=1 run8)"
echo "CV_HD=0: $(CV_HD=0 run8)"
} 2>&1 | tee $O/hd_shapes.txt
