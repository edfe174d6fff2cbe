"""Frozen-BN whole-network gradients at 512x1024 (F=20): ours vs the fp64 oracle, next to the fp32 oracle vs the fp64 oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests')): sys.path.insert(0, p)
import numpy as np, torch, torch.nn as nn
import addk, oracle
from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor, rel_err
from addk.modeling.ADD import ADD
torch.set_num_threads(16)
dev = torch.device('cuda:0')
args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), 0)
mo = oracle.ADD(*args); fill_params(mo, 600)
ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
m64 = oracle.ADD(*args).double(); m64.load_state_dict(mo.state_dict())
for m in (mo, ma, m64): m.eval()
hw = (512, 1024)
x = rand_tensor(61, 'frozen_x', (2, 3) + hw)
tgt = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (2,) + hw)).long()
crit = nn.CrossEntropyLoss(ignore_index=255)
t0 = time.time()
(sum(crit(y, tgt) for y in mo(x)) / 2).backward(); print('oracle fp32 %.1fs' % (time.time() - t0), flush=True)
t0 = time.time()
(sum(crit(y, tgt) for y in m64(x.double())) / 2).backward(); print('oracle fp64 %.1fs' % (time.time() - t0), flush=True)
(sum(crit(y, tgt.to(dev)) for y in ma(x.to(dev))) / 2).backward(); torch.cuda.synchronize()
pa, p64 = dict(ma.named_parameters()), dict(m64.named_parameters())
rows = []
for k, p in mo.named_parameters():
    if p.dim() == 4 and p.grad is not None:
        rows.append((rel_err(pa[k].grad.cpu().double(), p64[k].grad), rel_err(p.grad.double(), p64[k].grad), k))
rows.sort(reverse=True)
for e_us, e_or, k in rows[:12]:
    print('%-44s ours vs fp64 %.2e   oracle32 vs fp64 %.2e' % (k, e_us, e_or))
print('max ours %.2e  max oracle32 %.2e  median ours %.2e  median oracle32 %.2e' % (
    max(r[0] for r in rows), max(r[1] for r in rows), sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2]))
