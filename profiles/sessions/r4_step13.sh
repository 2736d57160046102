cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s13; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_gpu.log
tr() { timeout 600 python3 bench.py --mode train --steps 10 --warmup 3 "$@" 2>$O/err_train.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f ms/step, host enqueue %.2f, loss %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['final_loss']))"; }
{
echo "train fwd pairs (default): $(tr)"
echo "train fwd triples: $(CV_TRAIN_FWD_PIECES=3 tr)"
echo "train fwd pairs (default): $(tr)"
echo "train fwd triples: $(CV_TRAIN_FWD_PIECES=3 tr)"
} 2>&1 | tee $O/train_ab.txt
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
{
for m in 4 6 7 5; do echo "CV_HD=$m shape 2: $(CV_HD=$m run --steps 240) | $(CV_HD=$m run --steps 240)"; done
echo "CV_HD=6 shape 2 rows>=8192: $(CV_HD=6 CV_HD_MIN_ROWS=8192 run --steps 240)"
echo "CV_HD=4 shape 2 rows>=8192: $(CV_HD=4 CV_HD_MIN_ROWS=8192 run --steps 240)"
} 2>&1 | tee $O/hd_masks.txt
