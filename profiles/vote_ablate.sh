# wall-time ablations of hv_fwd_tiles on the teacher predictions of the bench scene (vote op ms per variant, one scene in flight):
# 0 full, 21 no LDS atomics, 22 no dense phase (nothing kept by the cull), 23 no record streaming, 25 no normalise / store
O=gpurun_out/${1:-r3u}; mkdir -p $O
for a in 0 21 22 23 25; do
  python bench.py --streams 1 --steps 20 --cpu-scenes 0 --train-steps 0 --algo $a 2>$O/vote_ablate_$a.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('algo $a vote_ms', round(d['stage_ms_median']['vote'],4))" | tee -a $O/vote_ablate.txt
done
