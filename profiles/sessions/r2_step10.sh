# round 2, step 10: 256-row workgroups for conv_hl on the fine levels (weight tile shared by twice the rows) A/B
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s11
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in "CV_HL_NW8=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1_$n.json
  env $v python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_$n.json
done
bash profiles/trace_one.sh r2s11 > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s11/bench*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
