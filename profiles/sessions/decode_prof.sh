#!/bin/bash
# per-kernel times of the decode launches (rocprofv3 kernel stats of a one-in-flight run), 80k and 300k scenes
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-decode_prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "80k" "300k --large --points 300000"; do
  set -- $cfg; tag=$1; shift
  rm -rf /tmp/pd_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$tag -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 --train-steps 0 "$@" > /tmp/pd_$tag.log 2>&1
  f=$(find /tmp/pd_$tag -name "*kernel_stats.csv" | head -1)
  python - "$f" $tag <<'PY' | tee -a $O/decode_kernels.txt
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec_" in n or "hv_fwd" in n:
        k = re.search(r"(dec_\w+|hv_fwd\w+)", n).group(1)
        print("%-5s %-28s calls %4s avg %9.1f us" % (sys.argv[2], k, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
