# vote tile shapes under six scenes in flight (the LDS footprint of a vote workgroup decides how it co-resides with the conv workgroups)
cd $GRAFT_REPO_ROOT
for defs in "" "-DHV_TW=4" "-DHV_TX=8 -DHV_TW=8" "-DHV_TX=8 -DHV_TW=4"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  for i in 1 2; do
    echo "defs '$defs' run $i: six $(timeout 200 python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'vote in region', round(d['stage_ms']['vote'],3), 'isolated', round(d['stage_ms_isolated']['vote'],3))")"
  done
done
