# wall-time ablations of hv_fwd_tiles on the headline workload's predictions (vote op ms per variant)
for a in 0 21 22 23 25 24; do
  python bench.py --streams 1 --steps 20 --cpu-scenes 0 --algo $a 2>gpurun_out/vote_ablate_$a.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('algo $a vote_ms', round(d['stage_ms']['vote'],4))" >> gpurun_out/vote_ablate.txt
  grep -i "tick\|phase" gpurun_out/vote_ablate_$a.err | tail -3 >> gpurun_out/vote_ablate.txt
done
