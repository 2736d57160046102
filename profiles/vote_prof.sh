cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/profiles/vote_time.py --ticks "$@" 2>&1 | grep "hv_fwd_tiles\|^vote"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $GRAFT_REPO_ROOT/profiles/vote_time.py "$@" 2>/dev/null | grep "^vote"
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "hv_" in r["Name"] or "fill" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min %.1f max %.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
