# A/B of compile-time occupancy targets of conv_hl (CV_SC_DEFS), three runs each to see past the run-to-run noise
cd $GRAFT_REPO_ROOT
for defs in "" "$1"; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  for i in 1 2 3; do
    echo "defs '$defs' run $i: one $(timeout 200 python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms']['net'],3))")  six $(timeout 200 python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")"
  done
done
