cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s20; mkdir -p $O
python -m pytest "tests/test_train_gpu.py::test_sorted_training_forward_equals_the_caller_order_forward" -m gpu -x -q 2>&1 | tail -3
MICRO_HL=1 bash profiles/conv_pmc.sh > $O/conv_pmc_hd.txt 2>&1
CV_HD=0 MICRO_HL=1 bash profiles/conv_pmc.sh > $O/conv_pmc_hl.txt 2>&1
tail -8 $O/conv_pmc_hd.txt
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
echo "default: $(run --steps 240) $(run --steps 240)"
echo "CV_MASKED_MIN_ROWS=8192: $(CV_MASKED_MIN_ROWS=8192 run --steps 240) $(CV_MASKED_MIN_ROWS=8192 run --steps 240)"
echo "CV_MASKED_MIN_ROWS=8192 CV_HD_MIN_ROWS=8192 CV_HD=6: $(CV_MASKED_MIN_ROWS=8192 CV_HD_MIN_ROWS=8192 CV_HD=6 run --steps 240) $(CV_MASKED_MIN_ROWS=8192 CV_HD_MIN_ROWS=8192 CV_HD=6 run --steps 240)"
echo "CV_MASKED_MIN_ROWS=40000: $(CV_MASKED_MIN_ROWS=40000 run --steps 240)"
} 2>&1 | tee $O/masked_min_rows.txt
