#!/usr/bin/env python
"""Every dense-conv FORWARD launch of the config-2 training plan, timed alone (HIP events, 10 repetitions) with its shape and achieved rate:
which shapes sit far from their kernel family's best.   python scripts/conv_table.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from addk.modeling.ADD import ADD
from addk.train import TrainStep
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
m = ADD(bench.NETWORK_ARCH, bench.C_INDEX, g, 19, bench.make_args(20), 0)
bench.init_weights(m); m.to(dev)
ts = TrainStep(m, (2, 3, 1024, 2048), use_graph=False)
x, t = bench.synthetic_batch(2, 1024, 2048, 1, dev)
ts.load_batch(x, t); ts.step(); torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for e in ts.g.meta:
    if e['kind'] != 'conv_fwd' or e['cmd'].name != 'conv_fwd':
        continue
    N, H, W, Cin, Cout, k, stride, dil = e['shape']
    if k == 1:
        continue
    dt = bench.time_launch(e['cmd'], reps=10)
    key = (H, W, Cin, Cout, k, stride, dil, bool(e.get('halo')))
    r = rows[key]; r[0] += 1; r[1] += dt; r[2] += e['flops']; r[3] += e['bytes']
print('%-44s %3s %9s %9s %8s %8s' % ('H W Cin Cout k stride dil halo', 'n', 'us/launch', 'GF', 'TF/s', 'GB/s'))
for key, (n, dt, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%-44s %3d %9.1f %9.2f %8.1f %8.0f' % (' '.join(str(v) for v in key), n, dt / n * 1e6, fl / n / 1e9, fl / dt / 1e12, by / dt / 1e9))
