#!/usr/bin/env python
"""Study behind VERDICT r02 item 1: where does the even-size train-mode gradient deficit come from?

For several weight/input draws of the F=4 network at an even (64x128) and an odd (65x129) size it prints the rel-L2 error of
the whole-network gradients against an fp64 evaluation of the same graph, for the fp32 oracle and for addk in two forms of
the BatchNorm backward:  centred  dy = G + c1' + c2*(x - mean)  (ATen's order, batchnorm.py:51-53) and folded
dy = G + c1 + c2*x with c1 = c1' - c2*mean (round 2).  Then, with BatchNorm FROZEN (no amplification), the per-layer ratio of
addk's conv-weight gradient error to the fp32 oracle's at the even size.

    python tests/tools/even_size_study.py [--draws 4] > gpurun_out/r03_even_size_study.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import numpy as np            # noqa: E402
import torch                  # noqa: E402
import torch.nn as nn         # noqa: E402

import addk                   # noqa: E402
import oracle                 # noqa: E402
from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor, rel_err   # noqa: E402


def target(hw, seed):
    t = torch.from_numpy(np.random.default_rng(seed).integers(0, 19, (2,) + hw)).long()
    t[0, :2, :5] = 255
    return t


def grads_of(m, x, t, train):
    crit = nn.CrossEntropyLoss(ignore_index=255)
    m.train(train)
    for p in m.parameters():
        p.grad = None
    ys = m(x)
    (sum(crit(y, t) for y in ys) / len(ys)).backward()
    return {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}


def rel_l2(ga, g64):
    den = sum(float((g64[n] ** 2).sum()) for n in g64) ** 0.5
    return sum(float(((ga[n] - g64[n]) ** 2).sum()) for n in g64) ** 0.5 / den


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--draws', type=int, default=4)
    ap.add_argument('--F', type=int, default=4)
    a = ap.parse_args()
    from addk.modeling.ADD import ADD
    dev = torch.device('cuda:0')
    addk.load()
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(a.F), 0)
    for hw in ((64, 128), (65, 129)):
        rows = {'o32': [], 'centred': [], 'folded': []}
        for k in range(a.draws):
            mo = oracle.ADD(*args)
            fill_params(mo, 600 + k)
            m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
            x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
            t = target(hw, 180 + 2 * k)
            g64 = grads_of(m64, x.double(), t, True)
            rows['o32'].append(rel_l2(grads_of(mo, x, t, True), g64))
            for name, env in (('centred', '1'), ('folded', '0')):
                os.environ['ADDK_BN_CENTERED'] = env
                ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
                rows[name].append(rel_l2(grads_of(ma, x.to(dev), t.to(dev), True), g64))
                del ma
        print('train-mode gradient rel-L2 vs fp64 at %dx%d (F=%d), %d draws' % (hw + (a.F, a.draws)))
        for name in ('o32', 'centred', 'folded'):
            print('  %-8s %s   ratio to fp32 oracle: %s' % (name, ' '.join('%.2e' % v for v in rows[name]),
                                                            ' '.join('%.2f' % (v / o) for v, o in zip(rows[name], rows['o32']))))
        sys.stdout.flush()
    # frozen BatchNorm, even size: per-layer ratio of the conv-weight gradient errors
    os.environ['ADDK_BN_CENTERED'] = '1'
    hw = (64, 128)
    for k in range(min(2, a.draws)):
        mo = oracle.ADD(*args)
        fill_params(mo, 600 + k)
        m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
        ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
        x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
        t = target(hw, 180 + 2 * k)
        g64, g32, ga = grads_of(m64, x.double(), t, False), grads_of(mo, x, t, False), grads_of(ma, x.to(dev), t.to(dev), False)
        groups = {}
        for n in g64:
            if g64[n].dim() != 4:
                continue
            key = '.'.join(n.split('.')[:2]) if n.startswith('cells.') else n.split('.')[0]
            groups.setdefault(key, []).append((rel_err(ga[n], g64[n]), rel_err(g32[n], g64[n])))
        print('frozen BatchNorm at %dx%d draw %d: whole-net rel-L2 addk %.2e  fp32 oracle %.2e' % (hw + (k, rel_l2(ga, g64), rel_l2(g32, g64))))
        for key in sorted(groups):
            ea, eo = max(v[0] for v in groups[key]), max(v[1] for v in groups[key])
            print('  %-16s n=%3d  addk %.2e  fp32 oracle %.2e  ratio %.2f' % (key, len(groups[key]), ea, eo, ea / max(eo, 1e-30)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
