# round 4, third GPU session: conv_hd (LDS-DMA operand rings, 256-row workgroups) against conv_hl
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s3; mkdir -p $O
timeout 900 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "conv_hd" 2>&1 | tail -15 | tee $O/pytest_hd.log
run8() { timeout 400 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f vote %.3f decode %.3f' % (i['net'], i['vote'], i['decode']))"; }
{
for hd in 0 7 4 3 0 7; do echo "CV_HD=$hd: $(CV_HD=$hd run8)"; done
echo "CV_HD=7 all rows (CV_HD_MIN_ROWS=1): $(CV_HD=7 CV_HD_MIN_ROWS=1 run8)"
echo "CV_HD=7 rows >= 4096: $(CV_HD=7 CV_HD_MIN_ROWS=4096 run8)"
} 2>&1 | tee $O/hd_ab.txt
for hd in 0 7; do echo "== layer times CV_HD=$hd"; CV_HD=$hd timeout 600 python profiles/layer_times.py 2>&1 | tail -70; done > $O/layer_times_hd.txt 2>&1
tail -3 $O/layer_times_hd.txt
