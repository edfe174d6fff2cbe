#!/usr/bin/env python
"""Which kernel family moves the frozen-BatchNorm gradients of config 5's architecture (F=40, genotype_1, 2x256x512)?

The fp32 / fp64 oracle gradients are computed once (stage `oracle`, cached in an .npz under $TMPDIR), then every variant runs in
its own process (the toggles are read when the library loads / the plan is built):

    python tests/tools/f40_frozen_probe.py > gpurun_out/r03_f40_frozen_probe.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

CACHE = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'f40_frozen_oracle.npz')       # ~150 MB: not under gpurun_out/ (64 MiB merge limit)
VARIANTS = [
    ('default', {}),
    ('unfused SepConv (fwd + bwd)', {'ADDK_FUSE_SEP': '0', 'ADDK_FUSE_SEP_BWD': '0'}),
    ('unfused SepConv backward only', {'ADDK_FUSE_SEP_BWD': '0'}),
    ('fp32 matrix kernels', {'ADDK_PROBE_PRECISION': 'fp32'}),
    ('stem2 on the generic fp32 kernel', {'ADDK_C3B_STRIDE2': '0'}),
]


def setup(gname):
    import numpy as np
    import torch
    import torch.nn as nn
    import oracle
    from _util import ARCH_C2, fill_params, make_args, rand_tensor
    geno = np.load(os.path.join(ROOT, 'searched_arch', '40_5e_38_lr', gname + '.npy'))
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], geno, 19, make_args(40), ARCH_C2['low_level_layer'])
    mo = oracle.ADD(*args)
    fill_params(mo, 900)
    hw = (256, 512)
    x = rand_tensor(61, 'f40_frozen_x', (2, 3) + hw)
    t = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (2,) + hw)).long()
    t[torch.from_numpy(np.random.default_rng(63).random((2,) + hw) < 0.05)] = 255      # tests/test_gpu_configs.py:_target
    return args, mo, x, t, nn.CrossEntropyLoss(ignore_index=255)


def stage_oracle(gname):
    import numpy as np
    import oracle
    args, mo, x, t, crit = setup(gname)
    mo.eval()
    (sum(crit(y, t) for y in mo(x)) / 2).backward()
    m64 = oracle.ADD(*args).double()
    m64.load_state_dict(mo.state_dict()); m64.eval()
    (sum(crit(y, t) for y in m64(x.double())) / 2).backward()
    out = {}
    for (k, p), (_, q) in zip(mo.named_parameters(), m64.named_parameters()):
        if p.dim() == 4 and p.grad is not None:
            out['g32/' + k] = p.grad.numpy()
            out['g64/' + k] = q.grad.numpy()
    os.makedirs(os.path.dirname(CACHE), exist_ok=True)
    np.savez(CACHE, **out)
    # the fp32 oracle's OWN sensitivity: the same network on an input nudged by one part in 2^24 / 2^23 (frozen BatchNorm: no batch
    # statistics to amplify anything, only ReLU / max-pool decisions flipping), held against the SAME fp64 gradients
    import torch
    from _util import rel_err
    for eps in (0.0, 2.0 ** -24, -2.0 ** -24, 2.0 ** -23):
        for p in mo.parameters():
            p.grad = None
        (sum(crit(y, t) for y in mo(x * (1.0 + eps))) / 2).backward()
        errs = sorted(rel_err(p.grad.double(), torch.from_numpy(out['g64/' + k])) for k, p in mo.named_parameters() if 'g64/' + k in out)
        print('fp32 oracle, input * (1 %+.1e):      %d gradients: max %.2e median %.2e  p90 %.2e' % (eps, len(errs), errs[-1], errs[len(errs) // 2], errs[int(len(errs) * 0.9)]))
        sys.stdout.flush()


def stage_variant(gname, label):
    import numpy as np
    import torch
    import addk
    from addk.modeling.ADD import ADD
    from _util import rel_err
    addk.load()
    if os.environ.get('ADDK_PROBE_PRECISION'):
        addk.set_precision(os.environ['ADDK_PROBE_PRECISION'])
    args, mo, x, t, crit = setup(gname)
    dev = torch.device('cuda:0')
    ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev); ma.eval()
    (sum(crit(y, t.to(dev)) for y in ma(x.to(dev))) / 2).backward()
    torch.cuda.synchronize()
    ref = np.load(CACHE)
    rows = []
    for k, p in ma.named_parameters():
        if 'g64/' + k in ref:
            g64 = torch.from_numpy(ref['g64/' + k])
            rows.append((rel_err(p.grad.cpu().double(), g64), rel_err(torch.from_numpy(ref['g32/' + k]).double(), g64), k, tuple(p.shape)))
    ours = sorted(r[0] for r in rows); theirs = sorted(r[1] for r in rows)
    print('%-34s %d gradients: addk max %.2e median %.2e p90 %.2e | fp32 oracle max %.2e median %.2e p90 %.2e' % (
        label, len(rows), ours[-1], ours[len(ours) // 2], ours[int(len(ours) * 0.9)], theirs[-1], theirs[len(theirs) // 2], theirs[int(len(theirs) * 0.9)]))
    for e, o, k, shp in sorted(rows, reverse=True)[:6]:
        print('      %-56s %-18s addk %.2e  oracle %.2e' % (k, 'x'.join(map(str, shp)), e, o))
    if os.environ.get('ADDK_PROBE_FULL'):          # every gradient, in backward order (heads first)
        def key(r):
            n = r[2]
            return (0, int(n[4]), n) if n.startswith('stem') else (1, int(n.split('.')[1]), n) if n.startswith('cells.') else (2, 0, n)
        for e, o, k, shp in sorted(rows, key=key, reverse=True):
            print('   %-56s %-18s addk %.2e  oracle %.2e  ratio %.1f' % (k, 'x'.join(map(str, shp)), e, o, e / max(o, 1e-30)))
    sys.stdout.flush()


def main():
    gname = 'genotype_1'
    if len(sys.argv) > 1 and sys.argv[1] == 'oracle':
        return stage_oracle(gname)
    if len(sys.argv) > 1 and sys.argv[1] == 'variant':
        return stage_variant(gname, sys.argv[2])
    if not os.path.exists(CACHE):
        subprocess.check_call([sys.executable, __file__, 'oracle'])
    only = os.environ.get('ADDK_PROBE_ONLY')
    for label, env in VARIANTS:
        if only and label != only:
            continue
        e = dict(os.environ); e.update(env)
        subprocess.call([sys.executable, __file__, 'variant', label], env=e)


if __name__ == '__main__':
    main()
