# everything the round-end driver runs, in one call: GPU tests, smoke(), default bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/check
mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-300
