# round 2, step 2: new decode (geometry in compact, one-barrier greedy, finalize in the last backproject workgroup, pinned results)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s2
mkdir -p $O
timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_production_size_gpu.py tests/test_sparse_gpu.py tests/test_proposals_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
bash profiles/trace_one.sh r2s2
