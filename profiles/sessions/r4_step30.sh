# scenes confined to XCD subsets (CU-masked streams)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s30; mkdir -p $O
./profiles/microbench/cu_mask_probe > $O/cu_mask_probe.txt 2>&1; cat $O/cu_mask_probe.txt
val() { tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f' % d['value'], d['stage_ms_isolated'] if 'stage_ms_isolated' in d else '', end=' ')"; }
for cfg in "--streams 8" "--streams 8 --cu-mask xcd1" "--streams 8 --cu-mask xcd1:blocked" "--streams 8 --cu-mask xcd2" "--streams 8 --cu-mask xcd2:blocked" "--streams 8 --cu-mask xcd4" "--streams 16 --cu-mask xcd1" "--streams 16 --cu-mask xcd1:blocked" "--streams 4 --cu-mask xcd2"; do
  echo -n "$cfg : 240 steps " >> $O/cu_mask.txt
  timeout 300 python3 bench.py --steps 240 --cpu-scenes 0 --train-steps 0 $cfg 2>>$O/err.txt | val >> $O/cu_mask.txt
  echo >> $O/cu_mask.txt
done
cat $O/cu_mask.txt; tail -5 $O/err.txt
