#!/bin/bash
# vote lists (count / scan / fill over chunks, 16-entry hand-outs): parity + per-kernel times; decode per-kernel times
O=gpurun_out/r3f; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_vote_gpu.py tests/test_decode_gpu.py tests/test_production_size_gpu.py tests/test_proposals_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
stats() { python - "$1" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print('%-40s calls %4s avg_us %9.1f'%(r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
for l in 1 0; do
  for sz in "" "--large"; do
  (cd /tmp && rm -rf /tmp/pv && CV_HV_LISTS=$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $GRAFT_REPO_ROOT/profiles/vote_time.py $sz > /tmp/pv.log 2>&1; echo "== CV_HV_LISTS=$l $sz"; grep "event ms" /tmp/pv.log; stats $(find /tmp/pv -name "*kernel_stats.csv" | head -1)) >> $O/vote_kernels.txt 2>&1
  done
done
cat $O/vote_kernels.txt
(cd /tmp && rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python $GRAFT_REPO_ROOT/profiles/decode_time.py > /tmp/pd.log 2>&1; grep dbg /tmp/pd.log; stats $(find /tmp/pd -name "*kernel_stats.csv" | head -1)) > $O/decode_kernels.txt 2>&1
cat $O/decode_kernels.txt
python profiles/vote_time.py --ticks 2>&1 | grep hv_fwd > $O/ticks.txt; cat $O/ticks.txt
