cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s16; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 --scene-call py "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f vote %.3f' % (i['net'], i['vote']))"; }
{
for defs in "" "-DHV_LDS_PAD=16384" "-DHV_TX=16 -DHV_TW=4" "-DHV_TX=32 -DHV_TW=16"; do
  touch canonicalvoting_amd/csrc/hv_vote.hip
  CV_HV_DEFS="$defs" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "vote defs '$defs': $(run --steps 240) | $(run --steps 240)"
done
touch canonicalvoting_amd/csrc/hv_vote.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
} 2>&1 | tee $O/vote_occupancy_mix.txt
