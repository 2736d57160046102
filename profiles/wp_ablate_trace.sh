# timing ablations of conv_rows_wp inside the real network program: per-dispatch kernel durations (rocprofv3
# --kernel-trace) of the conv kernels of the last forward, one column per CV_WP_ABL value
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/wp_ablate_trace; mkdir -p $O
for abl in ${ABLS:-0 2 12 15}; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-D${ABL_MACRO:-CV_WP_ABL}=$abl $EXTRA_DEFS" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  rm -rf /tmp/pa
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pa -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 6 --warmup 2 --cpu-scenes 0 > /tmp/pa.log 2>&1)
  t=$(find /tmp/pa -name "*kernel_trace.csv" | head -1)
  python - "$t" $O/abl_$abl.csv <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "conv_stem" in r["Kernel_Name"]]
a = stems[-2]; b = stems[-1]
with open(sys.argv[2], "w") as f:
    for r in rows[a:b]:
        n = r["Kernel_Name"]
        if "conv_" in n or "head_joint" in n:
            f.write("%s,%.2f,%s\n" % (n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", ";"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
        if "head_joint" in n:
            break
PY
done
python - <<'PY'
import os
abls = [int(x) for x in os.environ.get("ABLS", "0 2 12 15").split()]
cols = {}
for a in abls:
    cols[a] = [l.strip().split(",") for l in open("gpurun_out/wp_ablate_trace/abl_%d.csv" % a)]
n = min(len(c) for c in cols.values())
print("idx kernel grid | us per CV_WP_ABL =", abls)
tot = {a: 0.0 for a in abls}
for i in range(n):
    print("%3d %-24s %9s " % (i, cols[abls[0]][i][0][:24], cols[abls[0]][i][2]), " ".join("%7.1f" % float(cols[a][i][1]) for a in abls))
    for a in abls: tot[a] += float(cols[a][i][1])
print("total", " ".join("%8.1f" % tot[a] for a in abls))
PY
