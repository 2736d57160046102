# host ceiling of the scene loop: small scenes (the kernels are short, the scene rate is the host's: Python + launches under the GIL)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --steps 480 --warmup 5 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; }
for pts in 3000 20000; do for st in 1 4 8; do echo "points=$pts streams=$st: $(run --points $pts --streams $st)"; done; done
