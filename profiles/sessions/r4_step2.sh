# round 4, second GPU session: what binds the throughput with eight scenes in flight?
#  (a) the in-launch split-K reduction (write-through publish) on / off
#  (b) marginal cost of whole stages (timing ablations: results wrong)
#  (c) conv_hl phases removed at compile time (CV_HL_ABL: 1 gathers, 2 MFMAs, 4 weight tiles, 8 epilogue, 16 map reads)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s2; mkdir -p $O
run8() { timeout 400 python3 bench.py --steps 240 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f vote %.3f decode %.3f' % (i['net'], i['vote'], i['decode']))"; }
{
echo "== (a) split-K reduction: in-launch (write-through publish + ticket) vs finish launches; scenes/s at 240 steps, 8 in flight | one-in-flight stage ms"
for f in 1 0 1 0; do echo "CV_HL_FUSE_FINISH=$f: $(CV_HL_FUSE_FINISH=$f run8)"; done
echo "== (b) stages removed (8 in flight)"
echo "full: $(run8)"
echo "no vote, no decode: $(run8 --ablate novote)"
echo "no decode: $(run8 --ablate nodecode)"
echo "vote + decode only: $(run8 --stage vote_decode)"
echo "finish launches skipped (CV_HL_FUSE_FINISH=0): $(CV_HL_FUSE_FINISH=0 run8 --ablate finish)"
echo "== (c) conv_hl phases removed at compile time (all with CV_HL_FUSE_FINISH=0)"
for abl in 0 1 2 4 8 16 3 7; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DCV_HL_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "CV_HL_ABL=$abl: $(CV_HL_FUSE_FINISH=0 run8)"
done
touch canonicalvoting_amd/csrc/sparse_conv.hip
python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
} 2>&1 | tee $O/throughput_ablations.txt
# the GPU suite on the tree with the in-launch reduction on
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
