#!/bin/bash
# decode greedy-walk variants (compile-time) + vote counts-only mode
O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp
stats() { python - "$1" "$2" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if sys.argv[2] in r['Name']: print('   %-40s calls %4s avg_us %9.1f'%(r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
for defs in "-DDEC_BLOCKED=1 -DDEC_CACHED=1 -DDEC_BBOX=1" "-DDEC_BLOCKED=0 -DDEC_CACHED=1 -DDEC_BBOX=0" "-DDEC_BLOCKED=0 -DDEC_CACHED=0 -DDEC_BBOX=0" "-DDEC_BLOCKED=1 -DDEC_CACHED=0 -DDEC_BBOX=0"; do
  CV_DEC_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  (cd /tmp && rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python $GRAFT_REPO_ROOT/profiles/decode_time.py > /tmp/pd.log 2>&1; echo "== $defs"; grep dbg /tmp/pd.log; stats $(find /tmp/pd -name "*kernel_stats.csv" | head -1) dec_greedy) >> $O/decode_variants.txt 2>&1
done
cat $O/decode_variants.txt
python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
python -m pytest tests/test_vote_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for l in 1 0; do
  (cd /tmp && rm -rf /tmp/pv && CV_HV_LISTS=$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $GRAFT_REPO_ROOT/profiles/vote_time.py > /tmp/pv.log 2>&1; echo "== CV_HV_LISTS=$l"; grep "event ms" /tmp/pv.log; stats $(find /tmp/pv -name "*kernel_stats.csv" | head -1) hv_) >> $O/vote_kernels.txt 2>&1
done
cat $O/vote_kernels.txt
