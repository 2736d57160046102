# knobs once more under the round-4 defaults (one-call scenes, conv_hd on the 96-column fine-level launches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s17; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
{
for st in 6 8 10 12; do echo "streams $st: 240 steps $(run --steps 240 --streams $st) $(run --steps 240 --streams $st) | 20 steps $(run --gpus 1 --steps 20 --warmup 5 --streams $st) $(run --gpus 1 --steps 20 --warmup 5 --streams $st) $(run --gpus 1 --steps 20 --warmup 5 --streams $st)"; done
for sg in 0 200 400 800; do echo "stagger $sg: 20 steps $(run --gpus 1 --steps 20 --warmup 5 --stagger-us $sg) $(run --gpus 1 --steps 20 --warmup 5 --stagger-us $sg) $(run --gpus 1 --steps 20 --warmup 5 --stagger-us $sg)"; done
for sp in 128 256 384 512; do echo "split target $sp: 240 steps $(run --steps 240 --split-target $sp) $(run --steps 240 --split-target $sp) | 20 steps $(run --gpus 1 --steps 20 --warmup 5 --split-target $sp) $(run --gpus 1 --steps 20 --warmup 5 --split-target $sp)"; done
for g in 2 3 4; do echo "mask groups $g: 240 steps $(CV_NET_MASK_GROUPS=$g run --steps 240) $(CV_NET_MASK_GROUPS=$g run --steps 240)"; done
} 2>&1 | tee $O/knobs_r4.txt
