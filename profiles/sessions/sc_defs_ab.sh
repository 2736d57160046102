# compile-time variants of sparse_conv.hip (CV_SC_DEFS) against the default build: layer times and bench, rebuilt on the box
cd $GRAFT_REPO_ROOT
for defs in "" "$1"; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "== defs: '$defs'"
  python profiles/layer_times.py 2>&1 | tail -1
  python bench.py --streams 1 --cpu-scenes 0 --steps 120 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one in flight', round(d['value'],1), round(d['stage_ms']['net'],3))"
  python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('six in flight', round(d['value'],1))"
done
