cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s18; mkdir -p $O
tr() { timeout 600 python3 bench.py --mode train --steps 10 --warmup 3 "$@" 2>$O/err_train.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f ms/step, host enqueue %.2f, loss %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['final_loss']))"; }
{
for rep in 1 2; do
echo "sorted training forward (default): $(tr)"
echo "caller-order forward (CV_TRAIN_SORTED=0): $(CV_TRAIN_SORTED=0 tr)"
done
} 2>&1 | tee $O/train_sorted_ab.txt
python -m pytest tests/test_train_gpu.py tests/test_production_size_gpu.py tests/test_bf16_gpu.py tests/test_layer_grads_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_train.log
