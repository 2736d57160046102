# probe: two mask groups through the generic (argsort) orders, conv kernel sums from the per-dispatch trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
touch canonicalvoting_amd/csrc/sparse_conv.hip
CV_SC_DEFS="-DCV_WP_NPRE=14" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
for g in 3 2; do
  rm -rf /tmp/pg
  (cd /tmp && CV_NET_MASK_GROUPS=$g rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 8 --warmup 3 --cpu-scenes 0 > /tmp/pg.log 2>&1)
  f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
  python - "$f" $g <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def per(pat): return sum(float(r['TotalDurationNs']) for r in rows if pat in r['Name']) / 1e3 / 11
print("groups", sys.argv[2], "conv_hl<3> %.0f conv_hl<1> %.0f conv_hl<2> %.0f finish_small %.0f finish %.0f total %.0f" % (per('conv_hl<3'), per('conv_hl<1'), per('conv_hl<2'), per('conv_finish_small'), per('conv_finish('), per('conv_hl') + per('conv_finish')))
PY
  tail -1 /tmp/pg.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms'])"
done
