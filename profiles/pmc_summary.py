"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z_0-9]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in agg.values() for c in k})
print('%-28s %6s ' % ('kernel', 'n') + ' '.join('%22s' % c[-22:] for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    n = max(len(v) for v in d.values())
    print('%-28s %6d ' % (k[:28], n) + ' '.join('%22.4g' % (sum(d.get(c, [0])) / max(1, len(d.get(c, [0])))) for c in names))
