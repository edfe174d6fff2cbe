"""Debug: whole-net training gradients — addk vs oracle(fp32) vs oracle(fp64)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn as nn
import addk, oracle
from _util import *
from addk.modeling.ADD import ADD

dev = torch.device('cuda:0')
def run(Fv, arch, hw, n=2):
    args = (arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(Fv), arch['low_level_layer'])
    o32 = oracle.ADD(*args); fill_params(o32, 600)
    o64 = oracle.ADD(*args); o64.load_state_dict(o32.state_dict()); o64.double()
    a = ADD(*args); a.load_state_dict(o32.state_dict()); a.to(dev)
    x = rand_tensor(61, 'dbg_x', (n, 3) + hw)
    tgt = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (n,) + hw)).long()
    for m in (o32, o64, a): m.train()
    res = {}
    for name, m, xx, tt in (('o32', o32, x, tgt), ('o64', o64, x.double(), tgt), ('addk', a, x.to(dev), tgt.to(dev))):
        ys = m(xx)
        loss = sum(nn.functional.cross_entropy(y, tt, ignore_index=255) for y in ys) / len(ys)
        loss.backward()
        res[name] = ({k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None},
                     [y.detach().double().cpu() for y in ys], loss.item())
    print('config F=%d C_index=%s hw=%s  loss o32 %.6f o64 %.6f addk %.6f' % (Fv, arch['C_index'], hw, res['o32'][2], res['o64'][2], res['addk'][2]))
    for i in range(len(res['o64'][1])):
        r = res['o64'][1][i]
        print('  logits%d: o32-vs-64 %.2e   addk-vs-64 %.2e' % (i, (res['o32'][1][i]-r).abs().max()/r.abs().max(), (res['addk'][1][i]-r).abs().max()/r.abs().max()))
    groups = {}
    for k, g64 in res['o64'][0].items():
        e32 = float((res['o32'][0][k]-g64).abs().max()/(g64.abs().max()+1e-30))
        ea = float((res['addk'][0][k]-g64).abs().max()/(g64.abs().max()+1e-30))
        gk = '.'.join(k.split('.')[:2])
        w = groups.setdefault(gk, [0, 0])
        w[0] = max(w[0], e32); w[1] = max(w[1], ea)
    for gk in sorted(groups):
        print('  %-28s o32-vs-64 %.2e   addk-vs-64 %.2e' % (gk, groups[gk][0], groups[gk][1]))

run(4, ARCH_C2, (65, 129))
run(4, dict(network_arch=ARCH_C2['network_arch'], C_index=[], low_level_layer=0), (65, 129))
run(20, ARCH_C2, (129, 257))
