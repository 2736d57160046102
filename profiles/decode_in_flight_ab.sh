# decode stage with seven scenes in flight: the greedy walk with / without its LDS copy of the list x wave priority
# (stage_ms.decode is a DEVICE time since round 6: the event sits behind the decode's last launch, in front of the host's wait)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
for rep in 1 2; do
for walk in lds reg; do for prio in 0 1; do
  CV_DEC_GREEDY=$walk CV_DEC_PRIO=$prio python bench.py --steps ${STEPS:-120} --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/dec_${walk}_p${prio}_$rep.json
  python - <<PY
import json
d = json.load(open("$O/dec_${walk}_p${prio}_$rep.json"))
print("walk $walk prio $prio rep $rep: %.1f scenes/s  decode in region mean %.3f median %.3f  isolated %.3f  | vote %.3f net %.3f | host step median %.2f max %.2f"
      % (d["value"], d["stage_ms"]["decode"], d["stage_ms_median"]["decode"], d["stage_ms_isolated"]["decode"], d["stage_ms"]["vote"], d["stage_ms"]["net"],
         d["step_host_ms"]["median"], d["step_host_ms"]["max"]))
PY
done; done; done 2>&1 | tee $O/decode_in_flight.txt
