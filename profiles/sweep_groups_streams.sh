# mask-group / masked-min-rows sweep with the default six scenes in flight (sweep_groups.sh is the one-scene-in-flight sweep)
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_groups_streams6.txt
: > $O
run() { python bench.py --steps 160 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms'])" >> $O; }
for g in 2 3 4 5; do echo "G=$g" >> $O; CV_NET_MASK_GROUPS=$g CV_MASK_GROUPS=$g run; done
for r in 8192 40000; do echo "MINROWS=$r" >> $O; CV_MASKED_MIN_ROWS=$r run; done
echo "G=4 again" >> $O; run
cat $O
