# round 5, step 24: the whole GPU suite and the training step after conv_rows_x6 was removed; then step 25 (wgrad tile heights)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s24
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest.txt
cat $O/pytest.txt
bash profiles/sessions/r5_step25.sh
