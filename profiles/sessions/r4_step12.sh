cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s12; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']))"; }
grid() {
for rep in 1 2 3; do
for sc in py c; do
echo "$1 $sc hl 240: $(CV_HD=0 run --scene-call $sc --steps 240)"
echo "$1 $sc hd2 240: $(CV_HD=4 CV_HD_SHAPE=2 run --scene-call $sc --steps 240)"
echo "$1 $sc hl 20: $(CV_HD=0 run --scene-call $sc --gpus 1 --steps 20 --warmup 5)"
echo "$1 $sc hd2 20: $(CV_HD=4 CV_HD_SHAPE=2 run --scene-call $sc --gpus 1 --steps 20 --warmup 5)"
done; done
}
{
grid early1
touch canonicalvoting_amd/csrc/sparse_conv.hip
CV_SC_DEFS="-DHD_EARLY=0" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
grid early0
} 2>&1 | tee $O/hd2_grid.txt
