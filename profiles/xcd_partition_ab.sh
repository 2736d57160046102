# experiment: scene streams restricted to groups of XCDs (CU masks) - scenes/s over 240 steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O; : > $O/xcd_partition.txt
run() { python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-60s %.1f scenes/s  net %.2f vote %.2f decode %.2f ms in region' % (' '.join(sys.argv[1:]), d['value'], d['stage_ms']['net'], d['stage_ms']['vote'], d['stage_ms']['decode']))" "$@" >> $O/xcd_partition.txt; tail -1 $O/xcd_partition.txt; }
run --streams 7
run --streams 8 --xcd-partition 4
run --streams 8 --xcd-partition 2
run --streams 8 --xcd-partition 2 --split-target 128
run --streams 8 --xcd-partition 2 --split-target 64
run --streams 8 --xcd-partition 1
run --streams 8 --xcd-partition 1 --split-target 64
run --streams 12 --xcd-partition 2 --split-target 128
run --streams 16 --xcd-partition 1 --split-target 64
run --streams 4 --xcd-partition 2 --split-target 128
run --streams 7
