# round 2, step 18: occupancy bitmap in front of the hash probes of the level-0 kernel maps, A/B
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2s19
mkdir -p $O
timeout 1800 python -m pytest tests/test_sparse_gpu.py tests/test_production_size_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "CV_MAP_BITMAP=1"; do
  n=$(echo $v | tr ' =' '__')
  env $v python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1_$n.json
  env $v python bench.py --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_$n.json
done
bash profiles/trace_one.sh r2s19 > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s19/bench*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], round(d["value"],1), {k:round(v,3) for k,v in (d.get("stage_ms_isolated") or d["stage_ms"]).items()})
    except Exception as e: print(f, "ERR", e)
PY
grep -i "build_kernel_maps\|bitmap" $O/kernel_stats_streams1.csv | cut -c1-140
