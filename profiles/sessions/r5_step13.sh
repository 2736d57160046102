# round 5, step 13: zskip (no partial tiles for rows without a neighbour in a mask group): tests, A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s13
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py tests/test_scene_call_gpu.py -x -q > $O/pytest_net.log 2>&1; tail -4 $O/pytest_net.log
: > $O/zskip.txt
for z in 1 0 1 0; do
  v1=$(CV_ZSKIP=$z python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms_isolated']['net'],3))")
  v2=$(CV_ZSKIP=$z python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  v3=$(CV_ZSKIP=$z python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "zskip=$z: 240 steps $v1 | one in flight $v2 | driver command $v3" >> $O/zskip.txt
done
cat $O/zskip.txt
