cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python $R/bench.py --mode train --steps 1 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter()
for r in rows:
    if 'conv_rows_x6' in r['Kernel_Name']:
        c[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X'))] += 1
for k, v in c.items(): print(v, k)
print(list(rows[0].keys()))
PY
