#!/usr/bin/env python
"""How far is the two-stream schedule of the training step from what its dependency DAG allows?  

Builds the config-2 TrainStep as bench.py does, runs the step's level-ordered command list on ONE stream with a device event after every
command (eager, so a tiny command's figure includes the host's launch cost: an upper estimate), then reports
  * the sum of the commands' durations (the single-stream step),
  * the longest dependency chain of the DAG the planner derives from the commands' read / write regions (no schedule can beat it),
  * the makespan of a list schedule of that DAG on 2 / 3 / unlimited in-order streams IF overlapped commands did not slow each other
    (they do: the measured two-stream step is printed beside it), and which command kinds the longest chain is made of.
    python scripts/critical_path.py [--steps 3]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['ADDK_STREAMS'] = '1'
os.environ['ADDK_GRAPH'] = '0'
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402


def deps_of(cmds):
    from addk.plan import _overlap
    writers, readers, deps = {}, {}, []
    for i, c in enumerate(cmds):
        d = set()
        for k in c.rd:
            d.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
        for k in c.wr:
            d.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
            d.update(j for r, j in readers.get(k[0], ()) if _overlap(r, k))
        deps.append(sorted(d))
        for k in c.wr:
            writers[k[0]] = [(r, j) for r, j in writers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])] + [(k, i)]
            readers[k[0]] = [(r, j) for r, j in readers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])]
        for k in c.rd:
            readers.setdefault(k[0], []).append((k, i))
    return deps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    import addk                                   # noqa: F401
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    genotype = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab/genotype.npy'))
    model = ADD(bench.NETWORK_ARCH, bench.C_INDEX, genotype, 19, bench.make_args(20, sync_bn=False), 0)
    bench.init_weights(model)
    model.to(dev)
    ts = TrainStep(model, (2, 3, 1024, 2048), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, use_graph=False)
    x, t = bench.synthetic_batch(2, 1024, 2048, 1, dev)
    ts.load_batch(x, t)
    ts.step(); ts.step()
    torch.cuda.synchronize()
    g = ts.g
    cmds = list(g.fwd) + list(g.bwd)
    st = torch.cuda.current_stream()
    dur = np.zeros(len(cmds))
    for _ in range(a.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(cmds) + 1)]
        ev[0].record(st)
        for i, c in enumerate(cmds):
            rc = c.fn(*c.args, st.cuda_stream)
            assert not rc, c.name
            ev[i + 1].record(st)
        torch.cuda.synchronize()
        dur += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(cmds))])
    dur /= a.steps
    deps = deps_of(cmds)
    n = len(cmds)
    finish, pred = np.zeros(n), [-1] * n
    for i in range(n):
        s = 0.0
        for j in deps[i]:
            if finish[j] > s:
                s, pred[i] = finish[j], j
        finish[i] = s + dur[i]
    end = int(np.argmax(finish))
    chain = []
    while end >= 0:
        chain.append(end); end = pred[end]
    chain.reverse()
    kinds = collections.Counter(); kt = collections.Counter()
    for i in chain:
        kinds[cmds[i].name] += 1; kt[cmds[i].name] += dur[i]
    print('commands %d (forward %d, backward %d); sum of durations %.2f ms; longest dependency chain %.2f ms over %d commands' % (
        n, len(g.fwd), len(g.bwd), dur.sum(), finish.max(), len(chain)))
    print('the chain by command kind (ms, count): ' + ', '.join('%s %.2f/%d' % (k, v, kinds[k]) for k, v in kt.most_common(12)))

    def makespan(ns):          # in-order streams, list order, a command starts when its stream and its dependencies are free; no interference
        free = [0.0] * ns
        fin = np.zeros(n)
        for i in range(n):
            ready = max([fin[j] for j in deps[i]], default=0.0)
            s = min(range(ns), key=lambda q: max(free[q], ready))
            fin[i] = max(free[s], ready) + dur[i]
            free[s] = fin[i]
        return fin.max()
    print('list schedule without interference: 1 stream %.2f ms, 2 streams %.2f, 3 streams %.2f, 8 streams %.2f' % (makespan(1), makespan(2), makespan(3), makespan(8)))
    # the planner's own stream assignment (plan.schedule: follow the most recent dependency's stream while it is that stream's tail, else the least
    # recently used stream), forward and backward lists separately as TrainStep replays them, against an earliest-start assignment of the same order
    # and against a critical-path-first order (longest remaining chain first), all in the no-interference model
    import copy
    from addk import plan as P

    def simulate(lst, d, assign):
        m = len(lst)
        dp = deps_of(lst)
        fin = np.zeros(m); free = collections.defaultdict(float)
        for i in range(m):
            ready = max([fin[j] for j in dp[i]], default=0.0)
            q = assign(i, ready, free)
            fin[i] = max(free[q], ready) + d[i]
            free[q] = fin[i]
        return fin.max()
    nf = len(g.fwd)
    res = {}
    for label, lst, d in (('forward', list(g.fwd), dur[:nf]), ('backward', list(g.bwd), dur[nf:])):
        cp = [copy.copy(c) for c in lst]
        P.schedule(cp, 2)
        res[label, 'planner'] = simulate(lst, d, lambda i, ready, free: cp[i].stream)
        res[label, 'earliest'] = simulate(lst, d, lambda i, ready, free: min((0, 1), key=lambda q: max(free[q], ready)))
        # critical-path-first: topological order by decreasing bottom level
        dp = deps_of(lst); m = len(lst)
        succ = [[] for _ in range(m)]
        for i in range(m):
            for j in dp[i]:
                succ[j].append(i)
        bl = np.zeros(m)
        for i in range(m - 1, -1, -1):
            bl[i] = d[i] + max([bl[k] for k in succ[i]], default=0.0)
        indeg = [len(dp[i]) for i in range(m)]
        import heapq
        heap = [(-bl[i], i) for i in range(m) if indeg[i] == 0]
        heapq.heapify(heap)
        order = []
        while heap:
            _, i = heapq.heappop(heap)
            order.append(i)
            for k in succ[i]:
                indeg[k] -= 1
                if indeg[k] == 0:
                    heapq.heappush(heap, (-bl[k], k))
        lst2 = [lst[i] for i in order]; d2 = d[order]
        res[label, 'cp_first'] = simulate(lst2, d2, lambda i, ready, free: min((0, 1), key=lambda q: max(free[q], ready)))
    for pol in ('planner', 'earliest', 'cp_first'):
        print('two streams, no interference, %-9s forward %.2f + backward %.2f = %.2f ms' % (pol + ':', res['forward', pol], res['backward', pol], res['forward', pol] + res['backward', pol]))
    big = [(dur[i], cmds[i].name) for i in range(n) if dur[i] > 0.15]
    print('commands above 150 us: %d, together %.2f ms (chip-filling: these do not overlap with each other for free)' % (len(big), sum(d for d, _ in big)))
    ts.close()


if __name__ == '__main__':
    main()
