cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s14; mkdir -p $O
for p in 3 2; do echo "== CV_TRAIN_FWD_PIECES=$p"; CV_TRAIN_FWD_PIECES=$p python -m pytest "tests/test_train_gpu.py::test_minkunet_training_step_gradients_match_oracle" -m gpu -x -q 2>&1 | grep -E "assert|Error|passed|failed" | head -8; done 2>&1 | tee $O/train_test_modes.txt
python -m pytest tests/test_train_gpu.py tests/test_bf16_gpu.py tests/test_sparse_gpu.py tests/test_production_size_gpu.py tests/test_scene_call_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_subset.log
