#!/usr/bin/env python
"""Run the ADD forward launch list a few times (for rocprofv3 --kernel-trace) and, given --csv, summarise one forward by
(kernel, grid) inside the cell segment.  python scripts/fwd_trace.py [--mode eval] | --csv trace.csv"""
import argparse, os, sys, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def summarise(f, n):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    mark = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
    last = rows[mark[-2]:mark[-1]]
    dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(dur(r) for r in last)
    span = int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])
    gaps = sum(max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) for a, b in zip(last, last[1:]))
    print('launches %d  kernel time %.2f ms  span %.2f ms  idle gaps %.2f ms' % (len(last), tot / 1e6, span / 1e6, gaps / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in last:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
        key = (k, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        agg[key][0] += 1; agg[key][1] += dur(r)
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
        print('%-42s wg=%-6d y=%-3d z=%-3d %4d %8.2f ms %5.1f%%  avg %8.1f us' % (k[0], k[1], k[2], k[3], c, t / 1e6, 100.0 * t / tot, t / c / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='eval'); ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--csv'); ap.add_argument('--top', type=int, default=60)
    ap.add_argument('--real', action='store_true', help='the list the model really runs: level-ordered, batched, two streams')
    a = ap.parse_args()
    if a.csv:
        return summarise(a.csv, a.top)
    import numpy as np, torch
    import addk, addk.plan as P
    from addk.modeling.ADD import ADD
    from bench import NETWORK_ARCH, C_INDEX, make_args
    dev = torch.device('cuda:0')
    g0 = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
    torch.manual_seed(1)
    m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0).to(dev)
    m.train(a.mode == 'train')
    x = torch.randn(a.batch, 3, 1024, 2048, device=dev)
    g = P.Graph(dev, a.mode == 'train', False, None)
    act, inref = g.input_nchw(x)
    inref.bind(x)
    m.emit(g, act)
    if a.real:
        g.reorder = True
        g.finalize(int(os.environ.get('ADDK_STREAMS', '2')))
        for _ in range(4):
            g.run_parallel(g.fwd, None)
        torch.cuda.synchronize()
        return
    g.finalize()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4):
        g.run(g.fwd, st)
    torch.cuda.synchronize()

if __name__ == '__main__':
    main()
