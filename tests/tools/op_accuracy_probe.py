#!/usr/bin/env python
"""Per-operator rounding error against fp64, addk vs the fp32 oracle, on the tiny maps where the even-size network shows a
forward deficit (tests/tools/even_size_study.py).  L2-relative errors of the train-mode output and of the BatchNorm running
statistics, median over draws.     python tests/tools/op_accuracy_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch                  # noqa: E402
import torch.nn as nn         # noqa: E402

import addk                   # noqa: E402
import oracle                 # noqa: E402
from _util import fill_params, rand_tensor   # noqa: E402

KW = dict(eps=1e-5, momentum=0.1, affine=True)


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def main():
    from addk.modeling.operations import OPS, ReLUConvBN
    dev = torch.device('cuda:0')
    addk.load()
    draws = int(os.environ.get('DRAWS', '8'))
    cases = [(prim, C, hw) for prim in ('sep_conv_3x3', 'sep_conv_5x5', 'dil_conv_3x3', 'dil_conv_5x5', 'rcb1x1')
             for C, hw in ((16, (4, 8)), (16, (16, 32)), (40, (16, 32)), (40, (64, 128)))]
    for prim, C, hw in cases:
        eo, ea, so, sa = [], [], [], []
        for k in range(draws):
            mk_a = (lambda: ReLUConvBN(C, C, 1, 1, 0, nn.BatchNorm2d, **KW)) if prim == 'rcb1x1' else (lambda: OPS[prim](C, 1, nn.BatchNorm2d, **KW))
            mk_o = (lambda: oracle.ReLUConvBN(C, C, 1, 1, 0, nn.BatchNorm2d, **KW)) if prim == 'rcb1x1' else (lambda: oracle.OPS[prim](C, 1, nn.BatchNorm2d, **KW))
            mo = mk_o(); fill_params(mo, 40 + k)
            m64 = mk_o(); m64.load_state_dict(mo.state_dict()); m64.double()
            ma = mk_a(); ma.load_state_dict(mo.state_dict()); ma.to(dev)
            # inputs like the network's: a positive mean (post-ReLU sums) on top of unit noise
            x = rand_tensor(50 + k, 'probe_x', (2, C) + hw) + 0.7
            for m in (mo, m64, ma):
                m.train()
            with torch.no_grad():
                yo, y64, ya = mo(x), m64(x.double()), ma(x.to(dev))
            eo.append(l2(yo, y64)); ea.append(l2(ya, y64))
            n = [k2 for k2 in mo.state_dict() if k2.endswith('running_var')][-1]
            so.append(l2(mo.state_dict()[n], m64.state_dict()[n])); sa.append(l2(ma.state_dict()[n], m64.state_dict()[n]))
        med = lambda v: sorted(v)[len(v) // 2]
        print('%-14s C=%-3d %3dx%-3d  output: addk %.2e  fp32 oracle %.2e  ratio %.2f | last running_var: addk %.2e  oracle %.2e  ratio %.2f' % (
            prim, C, hw[0], hw[1], med(ea), med(eo), med(ea) / med(eo), med(sa), med(so), med(sa) / med(so)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
