"""Two-stream timeline of the LAST bench step in a rocprofv3 kernel trace CSV: per-queue busy time, time with 0 / 1 / 2+
kernels in flight, and which kernel families run alone (nothing overlapping them)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = (sgd[-2] + 1, sgd[-1] + 1) if len(sgd) >= 2 else (0, len(rows))
last = rows[a:b]
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
ev = []
for i, r in enumerate(last):
    ev.append((int(r['Start_Timestamp']), 1, i)); ev.append((int(r['End_Timestamp']), -1, i))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth = 0; prev = t0; hist = collections.Counter(); alone = collections.Counter(); live = set()
for t, d, i in ev:
    if t > prev:
        hist[min(depth, 3)] += t - prev
        if depth == 1:
            alone[name(last[next(iter(live))])] += t - prev
    prev = t
    if d > 0: live.add(i)
    else: live.discard(i)
    depth += d
print('wall %.2f ms; in flight: none %.2f ms, one %.2f ms, two %.2f ms, three+ %.2f ms' % ((t1 - t0) / 1e6, hist[0] / 1e6, hist[1] / 1e6, hist[2] / 1e6, hist[3] / 1e6))
q = collections.Counter()
for r in last: q[r.get('Queue_Id', '?')] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('busy per queue (ms):', {k: round(v / 1e6, 2) for k, v in q.items()})
print('time spent running ALONE, by kernel (ms):')
for k, v in alone.most_common(25): print('  %-42s %.2f' % (k, v / 1e6))
