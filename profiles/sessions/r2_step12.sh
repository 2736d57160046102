# round 2, step 12: cv_sp_scene_plan (sort + levels + maps in one call, level counts on a side stream), two-digit sort for single scenes
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s14
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
bash profiles/trace_one.sh r2s14 > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s14/bench*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
