# conv_hd skeleton: what is left without the unit loop / without the epilogue (timing ablations, one scene in flight)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s23; mkdir -p $O
run1() { timeout 400 python3 bench.py --steps 60 --streams 1 --scene-call py --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'net %.3f' % (d['stage_ms']['net']))"; }
run8() { timeout 400 python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for abl in 0 7 8 16 24; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DCV_HD_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "CV_HD_ABL=$abl: one in flight $(run1) | eight in flight $(run8)"
done 2>&1 | tee $O/hd_skeleton.txt
touch canonicalvoting_amd/csrc/sparse_conv.hip; python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
