# round 5, step 29: the training tests with the hl forward as the default
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s29
mkdir -p $O
timeout 2400 python -m pytest tests/test_train_gpu.py -m gpu -q 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
