# conv_hl unit slots (CV_HL_NS2: bit NB - 1 = two slots, else three) re-measured with seven scenes in flight (round 2 measured them one at a time)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O; : > $O/hl_slots_in_flight.txt
run() { tag="$1"; shift; env "$@" python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > /tmp/a.json
  env "$@" python bench.py --streams 1 --steps 60 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > /tmp/c.json
  python -c "
import json
a,c=[json.load(open('/tmp/%s.json'%k)) for k in 'ac']
print('%-34s 240 steps %.1f (net in region %.2f ms) | one in flight %.1f (net %.3f ms)' % ('$tag', a['value'], a['stage_ms']['net'], c['value'], c['stage_ms']['net']))" >> $O/hl_slots_in_flight.txt; tail -1 $O/hl_slots_in_flight.txt; }
run "CV_HL_NS2=7 (default)" CV_HL_NS2=7
run "CV_HL_NS2=5 (64 columns: 3 slots)" CV_HL_NS2=5
run "CV_HL_NS2=0 (all: 3 slots)" CV_HL_NS2=0
run "CV_HL_NS2=3 (96 columns: 3 slots)" CV_HL_NS2=3
run "CV_HL_NS2=7 again" CV_HL_NS2=7
