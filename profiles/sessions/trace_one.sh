# per-dispatch trace + kernel stats of the default workload (teacher-fed decode), one scene in flight
# usage: bash profiles/trace_one.sh <outdir-name> [extra bench args]
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 8 --warmup 3 --cpu-scenes 0 "$@" > /tmp/p1.log 2>&1; tail -2 /tmp/p1.log; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_streams1.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python - "$t" $O/trace_tail.csv <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-1000:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,gap_us,grid,wg,lds,vgpr,name\n")
    prev = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%.1f,%.1f,%.1f,%s,%s,%s,%s,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r["Kernel_Name"][:90].replace(",", ";")))
        prev = e
PY
)
ls -la $O
