# conv_hd vs conv_hl per kernel (rocprofv3 --kernel-trace --stats, one scene in flight), and conv_hd's own ablations
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4s4; mkdir -p $O
for hd in 0 7; do
  rm -rf /tmp/kt_$hd
  CV_HD=$hd timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$hd --output-format csv -- python $R/bench.py --streams 1 --steps 60 --warmup 3 --cpu-scenes 0 --train-steps 0 > /dev/null 2>&1
  f=$(find /tmp/kt_$hd -name "*kernel_stats.csv" | head -1)
  echo "== CV_HD=$hd"; head -12 $f | cut -c1-160
  cp $f $O/kernel_stats_hd$hd.csv
done 2>&1 | tee $O/kernel_stats_summary.txt
cd $R
run1() { timeout 400 python3 bench.py --steps 60 --streams 1 --warmup 3 --cpu-scenes 0 --train-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['stage_ms']
print(round(d['value'],1), 'net %.3f' % (i['net']))"; }
for abl in 0 1 2 4 3 7; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DCV_HD_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "CV_HD_ABL=$abl (CV_HD=7, one in flight): $(CV_HD=7 run1)"
done 2>&1 | tee $O/hd_ablations.txt
