# round 5, step 31: which assertion of the unshared-mask test fails on the hl forward; streams of a training step in the kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s31
mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q 2>&1 | grep -E "^E|passed|failed" | head -12 > $O/pytest.txt
cat $O/pytest.txt
(cd /tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 4 --warmup 3 --cpu-scenes 0 > /tmp/pt.log 2>&1; t=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); python - "$t" > $O/train_streams.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# the last step: from the last 'fused_adam'-like kernel backwards
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if 'nll_loss_forward' in n]      # one per step (the loss)
print('adam launches', len(adam))
if len(adam) >= 2:
    lo, hi = adam[-2] + 1, adam[-1] + 1
    step = rows[lo:hi]
    t0, t1 = step[0]['s'], max(r['e'] for r in step)
    print('last step: %d launches, %.2f ms' % (len(step), (t1 - t0) / 1e6))
    qs = {}
    for r in step:
        q = r.get('Queue_Id', '?'), r.get('Stream_Id', '?')
        d = qs.setdefault(q, [0, 0.0, 1e30, 0])
        d[0] += 1; d[1] += (r['e'] - r['s']) / 1e6; d[2] = min(d[2], (r['s'] - t0) / 1e6); d[3] = max(d[3], (r['e'] - t0) / 1e6)
    for q, d in qs.items():
        print('queue/stream', q, 'launches', d[0], 'busy %.2f ms' % d[1], 'first %.2f last %.2f ms' % (d[2], d[3]))
    # timeline of wgrad kernels vs main
    last_main = max(r['e'] for r in step if 'wgrad' not in r['Kernel_Name'])
    last_wg = max([r['e'] for r in step if 'wgrad' in r['Kernel_Name']] or [t0])
    print('last non-wgrad kernel ends at %.2f ms, last wgrad kernel at %.2f ms' % ((last_main - t0) / 1e6, (last_wg - t0) / 1e6))
P
)
cat $O/train_streams.txt
