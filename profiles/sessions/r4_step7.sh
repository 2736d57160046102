# one C call per scene (cv_detect_scene_f32) against the call-by-call pipeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s7; mkdir -p $O
timeout 1200 python -m pytest tests/test_scene_call_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_scene_call.log
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d.get('stage_ms_isolated') or d['stage_ms']
print(round(d['value'],1), 'iso net %.3f' % (i['net']), 'host median %.2f max %.2f' % (d['step_host_ms']['median'], d['step_host_ms']['max']), d['detections_per_scene'])"; }
{
for rep in 1 2 3; do
for sc in py c; do
echo "--scene-call $sc, 20 steps: $(run --scene-call $sc --gpus 1 --steps 20 --warmup 5)"
done; done
for sc in py c py c; do echo "--scene-call $sc, 240 steps: $(run --scene-call $sc --steps 240)"; done
echo "--scene-call c, 240 steps, CV_HD=4 shape 1: $(CV_HD=4 CV_HD_SHAPE=1 run --scene-call c --steps 240)"
echo "--scene-call c, 20 steps, CV_HD=4 shape 1: $(CV_HD=4 CV_HD_SHAPE=1 run --scene-call c --gpus 1 --steps 20 --warmup 5)"
echo "--scene-call c, 20 steps, no stagger: $(run --scene-call c --gpus 1 --steps 20 --warmup 5 --stagger-us 0)"
echo "--scene-call c, 20 steps, stagger 200: $(run --scene-call c --gpus 1 --steps 20 --warmup 5 --stagger-us 200)"
} 2>&1 | tee $O/scene_call_ab.txt
tail -5 $O/err.txt
