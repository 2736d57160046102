# round 2, step 1: new production-size parity tests + the new default bench line
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s1
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; tail -30 $O/pytest.log
python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json; tail -5 $O/bench.err
python bench.py --steps 20 --warmup 3 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_20steps.json
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
