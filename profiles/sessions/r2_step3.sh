# round 2, step 3: vote with (y, x-strip) record bins
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s3
mkdir -p $O
timeout 1200 python -m pytest tests/test_vote_gpu.py tests/test_production_size_gpu.py tests/test_decode_gpu.py tests/test_proposals_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --streams 1 --cpu-scenes 0 --predictions network 2>/dev/null | tail -1 > $O/bench_streams1_net.json
python bench.py --streams 1 --cpu-scenes 0 --large --points 300000 --steps 40 2>/dev/null | tail -1 > $O/bench_streams1_300k.json
python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY'
import json
for f in ("bench_streams1","bench_streams1_net","bench_streams1_300k","bench"):
    d=json.load(open("gpurun_out/r2s3/%s.json"%f))
    print(f, round(d["value"],1), {k:round(v,3) for k,v in d["stage_ms"].items()}, "frac", round(d["roofline"]["frac"],3), d["roofline"].get("isolated_frac"))
PY
