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
    ap.add_argument('--grads', type=int, default=1)
    ap.add_argument('--detail', type=int, default=0)
    ap.add_argument('--trace', type=int, default=0)
    ap.add_argument('--gradtrace', type=int, default=-1)
    a = ap.parse_args()
    from addk.modeling.ADD import ADD
    dev = torch.device('cuda:0')
    addk.load()
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(a.F), 0)
    if a.gradtrace >= 0:
        layer_trace(args, (64, 128), a.gradtrace, dev)
        grad_trace(args, (64, 128), a.gradtrace, dev)
        return
    if a.trace:
        layer_trace(args, (64, 128), 0, dev)
        return
    for hw in (((64, 128), (65, 129)) if a.grads else ()):
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
    # Where does the FORWARD pass start to differ?  After one train-mode forward every BatchNorm's running_mean / running_var
    # carries its batch statistics: per layer (network order), the error of those statistics against fp64, addk vs fp32 oracle.
    os.environ['ADDK_BN_CENTERED'] = '1'
    for hw in ((64, 128), (65, 129)):
        for k in range(min(2, a.draws)):
            mo = oracle.ADD(*args)
            fill_params(mo, 600 + k)
            m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
            ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
            x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
            for m, xx in ((mo, x), (m64, x.double()), (ma, x.to(dev))):
                m.train()
                with torch.no_grad():
                    m(xx)
            s64, s32, sa = m64.state_dict(), mo.state_dict(), ma.state_dict()
            print('running statistics after one train-mode forward at %dx%d draw %d (max-abs error / max-abs value, per BatchNorm in module order)' % (hw + (k,)))
            groups = {}
            for n in s64:
                if not n.endswith('running_var'):
                    continue
                key = '.'.join(n.split('.')[:2]) if n.startswith('cells.') else n.split('.')[0]
                groups.setdefault(key, []).append((rel_err(sa[n], s64[n]), rel_err(s32[n], s64[n])))
            for key in groups:
                ea = sorted(v[0] for v in groups[key]); eo = sorted(v[1] for v in groups[key])
                print('  %-16s n=%3d  running_var: addk max %.2e med %.2e | fp32 oracle max %.2e med %.2e | ratio max %.2f med %.2f' % (
                    key, len(ea), ea[-1], ea[len(ea) // 2], eo[-1], eo[len(eo) // 2], ea[-1] / max(eo[-1], 1e-30), ea[len(ea) // 2] / max(eo[len(eo) // 2], 1e-30)))
            if a.detail and hw == (64, 128) and k == 0:
                for n in s64:
                    if n.endswith('running_var') and (n.startswith('cells.0.') or n.startswith('cells.1.') or n.startswith('cells.2.') or n.startswith('cells.3.')):
                        nm = n.replace('running_var', 'running_mean')
                        print('    %-44s var addk %.2e o32 %.2e | mean addk %.2e o32 %.2e' % (
                            n[:-12], rel_err(sa[n], s64[n]), rel_err(s32[n], s64[n]), rel_err(sa[nm], s64[nm]), rel_err(s32[nm], s64[nm])))
            sys.stdout.flush()
            del ma


def layer_trace(args, hw, k, dev):
    """Raw conv outputs (the tensors entering each BatchNorm) of one train-mode forward, addk and fp32 oracle against fp64, in
    execution order: where does addk's error leave the oracle's?"""
    import addk.plan as P
    from addk.modeling.ADD import ADD
    mo = oracle.ADD(*args)
    fill_params(mo, 600 + k)
    m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
    ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
    x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
    caps = {}
    for tag, m in (('o32', mo), ('o64', m64)):
        names = {mod: n for n, mod in m.named_modules()}
        store = caps.setdefault(tag, [])
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.register_forward_pre_hook(lambda md, inp, store=store, names=names: store.append((names[md], inp[0].detach().double())))
        m.train()
        with torch.no_grad():
            m(x if tag == 'o32' else x.double())
    P.TRACE_BN = []
    ma.train()
    with torch.no_grad():
        ma(x.to(dev))
    torch.cuda.synchronize()
    names = {mod: n for n, mod in ma.named_modules()}
    got = [(names[mod], raw.view().permute(0, 3, 1, 2).double().cpu()) for mod, raw in P.TRACE_BN]
    P.TRACE_BN = None
    # addk emits in its own order: match by (module name, occurrence)
    def index(lst):
        seen, out = {}, {}
        for n, t in lst:
            i = seen.get(n, 0); seen[n] = i + 1
            out[(n, i)] = t
        return out
    ia, i32 = index(got), index(caps['o32'])
    print('raw conv outputs entering each BatchNorm, train-mode forward at %dx%d draw %d: L2-relative error vs fp64' % (hw + (k,)))
    seen = {}
    for n, t64 in caps['o64']:
        i = seen.get(n, 0); seen[n] = i + 1
        a_, o_ = ia.get((n, i)), i32[(n, i)]
        if a_ is None or tuple(a_.shape) != tuple(t64.shape):
            print('  %-44s (no addk match)' % n)
            continue
        den = float(t64.norm())
        ea, eo = float((a_ - t64).norm()) / den, float((o_ - t64).norm()) / den
        print('  %-44s %-18s addk %.2e  fp32 oracle %.2e  ratio %.2f' % (n, 'x'.join(str(v) for v in t64.shape[1:]), ea, eo, ea / max(eo, 1e-30)))
    sys.stdout.flush()


def grad_trace(args, hw, k, dev):
    """Per-parameter gradient error (rel-L2 of that parameter's gradient vs fp64), addk and fp32 oracle, train mode, listed in
    BACKWARD order (decoder first): where does addk's backward leave the oracle's?"""
    from addk.modeling.ADD import ADD
    mo = oracle.ADD(*args)
    fill_params(mo, 600 + k)
    m64 = oracle.ADD(*args); m64.load_state_dict(mo.state_dict()); m64.double()
    ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
    x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
    t = target(hw, 180 + 2 * k)
    g64, g32, ga = grads_of(m64, x.double(), t, True), grads_of(mo, x, t, True), grads_of(ma, x.to(dev), t.to(dev), True)
    print('train-mode gradients at %dx%d draw %d: whole-net rel-L2 addk %.2e  fp32 oracle %.2e' % (hw + (k, rel_l2(ga, g64), rel_l2(g32, g64))))
    names = [n for n, _ in mo.named_parameters() if n in g64]
    def key(n):
        if n.startswith('stem'):
            return (0, int(n[4]), n)
        if n.startswith('cells.'):
            return (1, int(n.split('.')[1]), n)
        return (2, 0, n)
    for n in sorted(names, key=key, reverse=True):
        if g64[n].dim() != 4:
            continue
        den = float(g64[n].norm())
        ea, eo = float((ga[n] - g64[n]).norm()) / den, float((g32[n] - g64[n]).norm()) / den
        print('  %-48s |g| %.2e  addk %.2e  fp32 oracle %.2e  ratio %.2f' % (n, den, ea, eo, ea / max(eo, 1e-30)))
    sys.stdout.flush()


if __name__ == '__main__':
    main()
