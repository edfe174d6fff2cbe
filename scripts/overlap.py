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
# largest idle intervals (nothing in flight) with the kernels around them
ends = sorted((int(r['End_Timestamp']), name(r)) for r in last)
starts = sorted((int(r['Start_Timestamp']), name(r)) for r in last)
gaps = []
depth = 0; prev_t = None; prev_name = None
ev2 = sorted([(int(r['Start_Timestamp']), 1, name(r)) for r in last] + [(int(r['End_Timestamp']), -1, name(r)) for r in last])
idle_from = None
for tt, dd, nm in ev2:
    if dd > 0:
        if depth == 0 and idle_from is not None: gaps.append((tt - idle_from[0], idle_from[1], nm))
        depth += 1
    else:
        depth -= 1
        if depth == 0: idle_from = (tt, nm)
gaps.sort(reverse=True)
print('idle intervals: %d, total %.2f ms; > 5 us: %d (%.2f ms); largest:' % (len(gaps), sum(g[0] for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for g in gaps[:12]: print('  %7.1f us  after %-40s before %s' % (g[0] / 1e3, g[1], g[2]))
# context of the three largest idle intervals
srt = sorted(last, key=lambda r: int(r['Start_Timestamp']))
big = []
depth = 0; idle_from = None
for tt, dd, idx in sorted([(int(r['Start_Timestamp']), 1, i) for i, r in enumerate(srt)] + [(int(r['End_Timestamp']), -1, i) for i, r in enumerate(srt)]):
    if dd > 0:
        if depth == 0 and idle_from is not None: big.append((tt - idle_from, idx))
        depth += 1
    else:
        depth -= 1
        if depth == 0: idle_from = tt
big.sort(reverse=True)
for gap, idx in big[:3]:
    print('gap %.1f us before launch #%d:' % (gap / 1e3, idx))
    for j in range(max(0, idx - 3), min(len(srt), idx + 3)):
        r = srt[j]
        print('   %s #%d q%s %-38s grid %s x %s  start +%.1f us  dur %.1f us' % ('>>' if j == idx else '  ', j, r.get('Queue_Id', '?'), name(r), r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'),
              (int(r['Start_Timestamp']) - int(srt[0]['Start_Timestamp'])) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
