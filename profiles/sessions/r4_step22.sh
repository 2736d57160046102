cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s22; mkdir -p $O
v() { timeout 300 python3 bench.py --streams 1 --stage vote_decode --steps 120 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('vote op %.4f kernel %.4f frac %.3f' % (d['stage_ms']['vote'], d['roofline']['avg_ms'], d['roofline']['frac']))"; }
{
for defs in "-DHV_SLOT_RANGES=1" "-DHV_SLOT_RANGES=0" "-DHV_SLOT_RANGES=1" "-DHV_SLOT_RANGES=0"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip; CV_HV_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "$defs: $(v) | $(v)"
done
touch canonicalvoting_amd/csrc/hv_vote.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
} | tee $O/vote_slot_ranges.txt
