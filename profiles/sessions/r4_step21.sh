cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s21; mkdir -p $O
run() { timeout 400 python3 bench.py --cpu-scenes 0 --train-steps 0 "$@" 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
{
for rep in 1 2 3 4 5; do
echo "20 steps: adaptive $(run --gpus 1 --steps 20 --warmup 5) fixed $(run --gpus 1 --steps 20 --warmup 5 --adaptive-split 0)"
done
echo "240 steps: adaptive $(run --steps 240) $(run --steps 240) fixed $(run --steps 240 --adaptive-split 0) $(run --steps 240 --adaptive-split 0)"
echo "streams 1: adaptive $(run --steps 60 --streams 1) fixed $(run --steps 60 --streams 1 --adaptive-split 0)"
} 2>&1 | tee $O/adaptive_split.txt
tail -3 $O/err.txt
