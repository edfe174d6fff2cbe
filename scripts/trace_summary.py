"""Summarise a rocprofv3 kernel trace CSV: per-kernel totals for the LAST bench step (between the last two sgd launches)."""
import csv, collections, glob, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = (sgd[-2] + 1, sgd[-1] + 1) if len(sgd) >= 2 else (0, len(rows))
last = rows[a:b]
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(dur(r) for r in last)
span = int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])
print('launches/step %d   sum of kernel time %.2f ms   wall span %.2f ms' % (len(last), tot / 1e6, span / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    agg[k][0] += 1; agg[k][1] += dur(r)
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print('%-46s %5d %8.2f ms %5.1f%%  avg %8.1f us' % (k, c, t / 1e6, 100.0 * t / tot, t / c / 1e3))
