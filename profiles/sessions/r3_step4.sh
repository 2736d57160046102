#!/bin/bash
# vote lists: per-kernel times (rocprofv3 --kernel-trace --stats) with and without the lists; decode tests with the cached-best walk
O=gpurun_out/r3e; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_decode_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest_dec.log 2>&1; tail -3 $O/pytest_dec.log
for l in 1 0; do
  (cd /tmp && rm -rf /tmp/pv$l && CV_HV_LISTS=$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv$l -- python $GRAFT_REPO_ROOT/profiles/vote_time.py > /tmp/pv$l.log 2>&1; f=$(find /tmp/pv$l -name "*kernel_stats.csv" | head -1); echo "== CV_HV_LISTS=$l"; grep "event ms" /tmp/pv$l.log; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('%-40s calls %4s avg_us %9.1f'%(r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
  ) >> $O/vote_kernels.txt 2>&1
done
cat $O/vote_kernels.txt
python profiles/decode_time.py > $O/decode_time.txt 2>&1; grep -v amdgpu $O/decode_time.txt | tail -5
