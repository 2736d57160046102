#!/bin/bash
# vote work queue (parts sized by the arc-length weight of each tile): parity, per-kernel times, bench
O=gpurun_out/r3i; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_vote_gpu.py tests/test_production_size_gpu.py tests/test_proposals_gpu.py tests/test_decode_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
stats() { python - "$1" "$2" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if sys.argv[2] in r['Name']: print('   %-40s calls %4s avg_us %9.1f'%(r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
for cfg in "1:" "2:" "2:--large" "1:--large"; do
  l=${cfg%%:*}; sz=${cfg#*:}
  (cd /tmp && rm -rf /tmp/pv && CV_HV_LISTS=$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $GRAFT_REPO_ROOT/profiles/vote_time.py $sz > /tmp/pv.log 2>&1; echo "== CV_HV_LISTS=$l $sz"; grep "event ms" /tmp/pv.log; stats $(find /tmp/pv -name "*kernel_stats.csv" | head -1) hv_) >> $O/vote_kernels.txt 2>&1
done
cat $O/vote_kernels.txt
python bench.py --streams 1 --steps 40 --warmup 5 --cpu-scenes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms_median'], d['roofline']['frac'])"
python bench.py --steps 240 --warmup 5 --cpu-scenes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms_isolated'], d['roofline']['isolated_frac'])"
