# mask-group / masked-min-rows sweep with one and six scenes in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_groups_streams6.txt
: > $O
run() { python bench.py --steps 160 --cpu-scenes 0 --streams $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $1:', round(d['value'],1), 'net', round(d['stage_ms']['net'],3))" >> $O; }
for g in 3 4 5 6; do echo "G=$g" >> $O; CV_NET_MASK_GROUPS=$g CV_MASK_GROUPS=$g run 1; CV_NET_MASK_GROUPS=$g CV_MASK_GROUPS=$g run 6; done
for r in 8192 40000; do echo "MINROWS=$r" >> $O; CV_MASKED_MIN_ROWS=$r run 1; CV_MASKED_MIN_ROWS=$r run 6; done
cat $O
