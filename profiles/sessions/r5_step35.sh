# round 5, step 35: the whole GPU suite on the training changes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s35
mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
