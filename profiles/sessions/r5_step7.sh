# round 5, step 7: conv_win v5 (unrolled chunk, immediates): tests, micro timing, bench, skeleton ablation on a real rebuild
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5s7}
mkdir -p $O
timeout 900 python -m pytest tests/test_windows_gpu.py -x -q > $O/pytest_windows.log 2>&1; tail -5 $O/pytest_windows.log
python profiles/win_micro.py 20 80000 1 > $O/micro.txt 2>&1
python profiles/win_micro.py 20 80000 2 >> $O/micro.txt 2>&1
MICRO_CIN=32 MICRO_COUT=32 python profiles/win_micro.py 20 80000 2 >> $O/micro.txt 2>&1
MICRO_CIN=128 python profiles/win_micro.py 20 80000 2 >> $O/micro.txt 2>&1
timeout 300 python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 2>$O/bench_s1_win1.err | tail -1 > $O/bench_s1_win1.json
timeout 300 python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 2>$O/bench_240_win1.err | tail -1 > $O/bench_240_win1.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 > $O/bench_20_win1.json
for abl in 15 2 8; do
  rm -f canonicalvoting_amd/_C/obj/sparse_win.hip.o
  CV_WIN_DEFS="-DCV_WIN_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "ABL=$abl" >> $O/micro.txt
  python profiles/win_micro.py 20 80000 1 >> $O/micro.txt 2>&1
  python profiles/win_micro.py 20 80000 2 >> $O/micro.txt 2>&1
done
grep -v amdgpu.ids $O/micro.txt
python - <<P
import json,glob,os
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d['value'],1), d.get('stage_ms_isolated'), (d.get('parity') or {}).get('net_max_abs_err'))
    except Exception as e: print(f, 'ERR', e, open(f).read()[:300])
P
