# round 5, step 37: training tests with the sorted twin as the default
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s37
mkdir -p $O
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_production_size_gpu.py tests/test_layer_grads_gpu.py tests/test_bf16_gpu.py tests/test_concurrency_gpu.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED" | head -20 > $O/pytest.txt
cat $O/pytest.txt
