"""idle time of the main stream inside one training step (kernel trace CSV of `bench.py --mode train` under rocprofv3): total,
and which launches the gaps sit in front of.  argv: kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [i for i, r in enumerate(rows) if 'nll_loss_forward' in r['Kernel_Name']]
step = rows[marks[-2]:marks[-1]]
main_q = collections.Counter(r['Queue_Id'] for r in step).most_common(1)[0][0]
main = [r for r in step if r['Queue_Id'] == main_q]
side = [r for r in step if r['Queue_Id'] != main_q]
t0, t1 = step[0]['s'], max(r['e'] for r in step)
busy = sum(r['e'] - r['s'] for r in main)
print('step %.2f ms: main stream %d launches busy %.2f ms, side stream %d launches busy %.2f ms (last ends %.2f ms)' % (
    (t1 - t0) / 1e6, len(main), busy / 1e6, len(side), sum(r['e'] - r['s'] for r in side) / 1e6,
    (max([r['e'] for r in side] or [t0]) - t0) / 1e6))
gaps = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
end = main[0]['e']
for r in main[1:]:
    g = r['s'] - end
    if g > 0:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-48:]
        gaps[k][0] += 1; gaps[k][1] += g / 1e3; tot += g / 1e3
    end = max(end, r['e'])
print('idle time of the main stream %.2f ms; in front of:' % (tot / 1e3))
for k, (n, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %-50s %4d gaps %8.1f us' % (k, n, us))
# thirds of the step: where the gaps are
third = (t1 - t0) / 6
buckets = [0.0] * 6
end = main[0]['e']
for r in main[1:]:
    g = r['s'] - end
    if g > 0:
        buckets[min(5, int((r['s'] - t0) / third))] += g / 1e3
    end = max(end, r['e'])
print('idle us per sixth of the step:', [round(b) for b in buckets])
