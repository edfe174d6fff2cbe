import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import addk
from addk.modeling.ADD import ADD
from addk.train import TrainStep
from addk.synth import fill_params
from bench import NETWORK_ARCH, C_INDEX, make_args, synthetic_batch
dev = torch.device('cuda:0')
g0 = np.load('/root/repo/searched_arch/autodeeplab/genotype.npy')
H, W = int(os.environ.get('HH', '1024')), int(os.environ.get('WW', '2048'))
x, t = synthetic_batch(2, H, W, 1, dev)
res = {}
for fuse in ('0', '1'):
    os.environ['ADDK_FUSE_SEP'] = fuse
    m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0)
    fill_params(m, 1001); m.to(dev)
    ts = TrainStep(m, (2, 3, H, W), use_graph=False)
    ts.load_batch(x, t)
    ts.forward_backward_only()
    torch.cuda.synchronize()
    res[fuse] = (ts.loss.item(), ts.flat_g.clone(), {id(p): n for n, p in m.named_parameters()}, ts)
    print('fuse', fuse, 'loss %.7f' % ts.loss.item(), 'gnorm %.6e' % float(ts.flat_g.double().norm()))
a, b = res['0'], res['1']
print('grad rel diff', float((a[1].double() - b[1].double()).norm() / a[1].double().norm()))
# per-parameter
ts0, ts1 = a[3], b[3]
names = [n for n, p in ts0.model.named_parameters()]
g0_ = ts0.grads(); g1_ = ts1.grads()
rows = []
for (n, p0), (_, p1) in zip(ts0.model.named_parameters(), ts1.model.named_parameters()):
    ga, gb = g0_.get(p0), g1_.get(p1)
    if ga is None or gb is None: continue
    d = float((ga.double() - gb.double()).norm() / (ga.double().norm() + 1e-30))
    rows.append((d, n))
rows.sort(reverse=True)
grp = {}
order = []
for d, n in rows:
    key = '.'.join(n.split('.')[:2]) if n.startswith('cells.') else n.split('.')[0]
    if key not in grp: order.append(key)
    grp.setdefault(key, []).append(d)
def keyf(k):
    return (0, 0) if k.startswith('stem') else ((1, int(k.split('.')[1])) if k.startswith('cells') else (2, 0))
for key in sorted(grp, key=keyf):
    v = sorted(grp[key])
    print('%-14s n=%3d  max %.3e  median %.3e  min %.3e' % (key, len(v), v[-1], v[len(v) // 2], v[0]))
