# round 5, step 32: the calls of bn_backward_apply4 in one training step (duration, grid) with and without the gradient twins
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s32
mkdir -p $O
for b in 0 1; do
(cd /tmp && rm -rf /tmp/pt && CV_TRAIN_BWD_HL=$b rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 4 --warmup 3 --cpu-scenes 0 > /tmp/pt.log 2>&1; t=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); python - "$t" > $O/apply_calls_$b.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
names = [r['Kernel_Name'] for r in rows]
marks = [i for i, n in enumerate(names) if 'nll_loss_forward' in n]
step = rows[marks[-2]:marks[-1]]
ap = [r for r in step if 'bn_backward_apply4' in r['Kernel_Name']]
print('calls', len(ap), 'total us', sum(r['e'] - r['s'] for r in ap) / 1e3)
for r in ap:
    # what else runs during this call
    others = [o['Kernel_Name'][:40] for o in step if o is not r and o['s'] < r['e'] and o['e'] > r['s']]
    print('%8.1f us grid %8s vgpr %s  overlapping: %s' % ((r['e'] - r['s']) / 1e3, r['Grid_Size_X'], r['VGPR_Count'], ', '.join(sorted(set(others)))[:100]))
P
)
done
head -30 $O/apply_calls_1.txt; echo; head -12 $O/apply_calls_0.txt
