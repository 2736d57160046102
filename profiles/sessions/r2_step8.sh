# round 2, step 8: conv_hl with NS unit slots (loads of unit u + NS - 1 in flight), A/B against 3 slots and the XCD-aware tiles
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s8
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in "CV_HL_DEEP=0 CV_XCD_TILES=0" "CV_HL_DEEP=1 CV_XCD_TILES=0" "CV_HL_DEEP=0 CV_XCD_TILES=1" "CV_HL_DEEP=1 CV_XCD_TILES=1"; do
  n=$(echo $v | tr ' =' '__')
  env $v python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1_$n.json
  env $v python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_$n.json
done
CV_XCD_TILES=0 bash profiles/trace_one.sh r2s8 > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s8/bench*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
