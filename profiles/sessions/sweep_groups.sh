for g in 3 4 5 6; do
  echo "G=$g" >> gpurun_out/sweep_groups.txt
  CV_NET_MASK_GROUPS=$g CV_MASK_GROUPS=$g python bench.py --streams 1 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])" >> gpurun_out/sweep_groups.txt
done
for r in 8192 40000 1000000; do
  echo "MINROWS=$r" >> gpurun_out/sweep_groups.txt
  CV_MASKED_MIN_ROWS=$r python bench.py --streams 1 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])" >> gpurun_out/sweep_groups.txt
done
