"""Layer-by-layer forward difference between the fused and the unfused SepConv path (train-mode forward, full size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import addk, addk.plan as P
from addk.modeling.ADD import ADD
from addk.synth import fill_params
from bench import NETWORK_ARCH, C_INDEX, make_args, synthetic_batch
dev = torch.device('cuda:0')
g0 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'searched_arch', 'autodeeplab', 'genotype.npy'))
H, W = int(os.environ.get('HH', '1024')), int(os.environ.get('WW', '2048'))
mode = os.environ.get('MODE', 'train')
x, t = synthetic_batch(2, H, W, 1, dev)
tr = {}
keep = []
for fuse in ('0', '1'):
    os.environ['ADDK_FUSE_SEP'] = fuse
    m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0)
    fill_params(m, 1001); m.to(dev)
    m.train(mode == 'train')
    P.TRACE_BN = []
    with torch.no_grad():
        ys = m(x)
    torch.cuda.synchronize()
    names = {mod: n for n, mod in m.named_modules()}
    tr[fuse] = [(names[mod], raw.view()) for mod, raw in P.TRACE_BN]
    P.TRACE_BN = None
    keep.append((m, ys))
d0 = {}
for n, v in tr['0']:
    d0.setdefault(n, []).append(v)
seen = {}
for n, v in tr['1']:
    i = seen.get(n, 0); seen[n] = i + 1
    u = d0[n][i]
    e = float((u.double() - v.double()).norm() / u.double().norm())
    mx = float((u - v).abs().max() / u.abs().max())
    print('%-40s %-16s rel-L2 %.2e  max %.2e' % (n, 'x'.join(str(s) for s in v.shape[1:]), e, mx))
for a, b in zip(keep[0][1], keep[1][1]):
    print('logits rel-L2 %.2e' % float((a.double() - b.double()).norm() / a.double().norm()))
