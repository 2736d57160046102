# round 5, step 40: raw stream handle / no-op device guard in the Python layer: host side and step time; tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s40
mkdir -p $O
python profiles/train_host_profile.py 2000 2>&1 | grep -E "^step|host enqueue" > $O/host.txt
for i in 1 2 3; do
  timeout 600 python bench.py --mode train --steps 16 --warmup 4 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train step', round(d['ms_per_step'],2), 'ms, host enqueue', round(d['host_enqueue_ms_per_step'],2))" >> $O/host.txt
done
cat $O/host.txt
timeout 2400 python -m pytest tests/test_train_gpu.py tests/test_concurrency_gpu.py tests/test_scene_call_gpu.py tests/test_sparse_gpu.py -m gpu -q 2>&1 | tail -2
