# round 2, step 7: XCD-aware tile numbering + mask orders sorted inside the eighths of the spatial row order
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s7
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
CV_XCD_TILES=0 python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1_noxcd.json
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
bash profiles/trace_one.sh r2s7 > /dev/null 2>&1
bash profiles/conv_l2_pmc.sh > $O/conv_l2_pmc.txt 2>&1
bash profiles/groups_micro.sh > $O/groups_micro.txt 2>&1
python - <<'PY'
import json
for f in ("bench_streams1_noxcd", "bench_streams1","bench"):
    try:
        d=json.load(open("gpurun_out/r2s7/%s.json"%f))
        print(f, round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()}, d.get("parity"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/conv_l2_pmc.txt $O/groups_micro.txt
