# round 2, step 5: device row sort (cv_sp_sort_rows), lazy caller-order manager, permutation folded into the stem map
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s5
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
bash profiles/trace_one.sh r2s5 > /dev/null 2>&1
python profiles/layer_times.py > $O/layer_times.txt 2>&1
python - <<'PY'
import json
for f in ("bench_streams1","bench"):
    d=json.load(open("gpurun_out/r2s5/%s.json"%f))
    print(f, round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()}, d.get("stage_ms_isolated"))
PY
