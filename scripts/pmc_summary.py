"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel name (sum over dispatches of the LAST step)."""
import csv, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
# keep only the last occurrence block: dispatch ids after the second-to-last sgd_kernel
ids = sorted(set(int(r['Dispatch_Id']) for r in rows))
sgd = sorted(set(int(r['Dispatch_Id']) for r in rows if 'sgd_kernel' in r['Kernel_Name']))
lo = sgd[-2] if len(sgd) >= 2 else -1
hi = sgd[-1] if sgd else 1 << 60
for r in rows:
    d = int(r['Dispatch_Id'])
    if not (lo < d <= hi):
        continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (d, k) not in seen:
        seen.add((d, k)); cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print('%-42s %5s ' % ('kernel', 'n') + ' '.join('%16s' % n[-16:] for n in names))
key = names[0]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print('%-42s %5d ' % (k, cnt[k]) + ' '.join('%16.4g' % v.get(n, 0) for n in names))
