#!/usr/bin/env python
"""Every conv DATA-GRADIENT launch of the config-2 training plan that is not a pointwise-kernel or batched launch, timed alone with its shape:
which ones still run on the generic fp32 kernel.   python scripts/dgrad_table.py"""
import os, sys, collections, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import addk._lib as L
from addk.modeling.ADD import ADD
from addk.train import TrainStep
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
m = ADD(bench.NETWORK_ARCH, bench.C_INDEX, g, 19, bench.make_args(20), 0)
bench.init_weights(m); m.to(dev)
ts = TrainStep(m, (2, 3, 1024, 2048), use_graph=False)
x, t = bench.synthetic_batch(2, 1024, 2048, 1, dev)
ts.load_batch(x, t); ts.step(); torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for c in ts.g.bwd:
    if c.name != 'conv_dgrad':
        continue
    a = c.args[0]._obj
    dt = bench.time_launch(c, reps=10)
    fl = 2.0 * a.N * a.OH * a.OW * a.Cout * a.KH * a.KW * a.dst.C
    key = (a.H, a.W, a.dst.C, a.Cout, a.KH, a.stride, a.dil, a.w_choff, a.cin_total, bool(a.wpack))
    r = rows[key]; r[0] += 1; r[1] += dt; r[2] += fl
print('%-56s %3s %9s %9s %8s' % ('H W C(dst) Cout k stride dil w_choff cin_total wpack', 'n', 'us/launch', 'GF', 'TF/s'))
for key, (n, dt, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print('%-56s %3d %9.1f %9.2f %8.1f' % (' '.join(str(v) for v in key), n, dt / n * 1e6, fl / n / 1e9, fl / dt / 1e12))
