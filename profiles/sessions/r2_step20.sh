# round 2, step 20: split-K reduced by the last-arriving workgroup (no finish launch on the coarse levels) A/B
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s20
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "CV_HL_FUSE_FINISH=0" "CV_HL_FUSE_FINISH=1"; do
  n=$(echo $v | tr ' =' '__')
  for i in 1 2; do
  echo "$v run $i: one $(env $v timeout 200 python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms']['net'],3))")  six $(env $v timeout 200 python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")"
  done
done
timeout 600 bash profiles/trace_one.sh r2s20 > /dev/null 2>&1
grep -c "conv_finish" $O/trace_tail.csv
