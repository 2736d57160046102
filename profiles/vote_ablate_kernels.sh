# kernel-only time of hv_fwd_tiles under its ablation variants (21 no LDS atomics, 22 no dense phase, 23 no record
# streaming, 25 no normalise / store), from rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pva
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pva -- python $GRAFT_REPO_ROOT/profiles/tmp_ab/abl.py 2>/dev/null | grep "^algo"
f=$(find /tmp/pva -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "hv_" in r["Name"] or "minmax" in r["Name"]:
        print("%-60s calls %s avg_us %.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
