"""Intrinsic spread of fp32 evaluations of the train-mode gradient at the even size: the fp32 ORACLE on inputs perturbed by 6e-8 relative
(one ulp-level nudge) against the fp64 gradient of the unperturbed input.  CPU only.   python tests/tools/chaos_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn as nn
import oracle
from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor
torch.set_num_threads(8)
args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4), 0)
crit = nn.CrossEntropyLoss(ignore_index=255)
hw = (64, 128)
def target(hw, seed):
    t = torch.from_numpy(np.random.default_rng(seed).integers(0, 19, (2,) + hw)).long(); t[0, :2, :5] = 255; return t
def grads(m, x, t):
    m.train()
    for p in m.parameters(): p.grad = None
    ys = m(x); (sum(crit(y, t) for y in ys) / len(ys)).backward()
    return {n: p.grad.detach().double() for n, p in m.named_parameters() if p.grad is not None}
def rel(ga, g64):
    den = sum(float((g64[n] ** 2).sum()) for n in g64) ** 0.5
    return sum(float(((ga[n] - g64[n]) ** 2).sum()) for n in g64) ** 0.5 / den
for k in range(4):
    mo = oracle.ADD(*args); fill_params(mo, 600 + k)
    m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
    x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw); t = target(hw, 180 + 2 * k)
    g64 = grads(m64, x.double(), t)
    errs = [rel(grads(mo, x, t), g64)]
    g = torch.Generator().manual_seed(99 + k)
    for j in range(5):
        xp = x * (1 + 6e-8 * torch.randn(x.shape, generator=g))
        errs.append(rel(grads(mo, xp, t), g64))
    print('draw %d: fp32 oracle %.2e | fp32 oracle on inputs perturbed by 6e-8 relative: %s' % (k, errs[0], ' '.join('%.2e' % e for e in errs[1:])), flush=True)
