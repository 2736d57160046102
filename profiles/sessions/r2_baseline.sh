# round-2 baseline of the round-1 code on this round's box: gpu tests, bench, per-dispatch trace of one-in-flight steps
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2base
mkdir -p $O
nproc > $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --teacher-forced --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_tf_streams1.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 6 --warmup 3 --cpu-scenes 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_streams1.csv; t=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); python - "$t" $O/trace_tail.csv <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-900:]
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
