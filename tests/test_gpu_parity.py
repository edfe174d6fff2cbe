"""GPU parity tests (run on the MI355X box: pytest -m gpu).  The HIP path (through the C ABI in libaddk.so) is
compared with the CPU oracle on identical seeded inputs and weights, and with the golden vectors produced by
the real reference.  Tolerance: 1e-3 relative fp32 as BASELINE.json's north_star states (max-abs error divided
by the max-abs of the reference tensor); most checks land 1-2 orders of magnitude below that."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle                       # noqa: E402  (the checker)
from _util import (ARCH_C2, ARCH_C3, GENOTYPE_AUTODEEPLAB, GENOTYPE_BASELINE_2, GENOTYPE_40_1,   # noqa: E402
                   NETWORK_PATH_BASELINE, fill_params, make_args, probe_weights, rand_tensor, rel_err, rms_err)

TOL = 1e-3
BN = nn.BatchNorm2d
KW = dict(eps=1e-5, momentum=0.1, affine=True)
REPORT = []


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    import addk
    addk.load()
    return torch.device('cuda:0')


FAILS = []


def _chk(name, a, b, tol=TOL):
    """Two metrics at the same tolerance: max-abs error / max-abs reference (the 1e-3 of north_star) AND rms error / rms reference,
    in which every element counts (a tensor whose small-magnitude elements were wrong passed the first alone: VERDICT r03 weak 1e)."""
    e, r = rel_err(a, b), rms_err(a, b)
    ok = e <= tol and r <= tol
    REPORT.append('%-60s %.3e  rms %.3e%s' % (name, e, r, '' if ok else '   <-- FAIL (tol %.0e)' % tol))
    if not ok:
        FAILS.append('%s: rel err %.3e / rms err %.3e > %.1e' % (name, e, r, tol))
    return e


@pytest.fixture(autouse=True)
def _collect_failures():
    """Every comparison of a test is evaluated and logged; the test fails at the end listing all of them."""
    del FAILS[:]
    yield
    bad = list(FAILS)
    del FAILS[:]
    assert not bad, '%d mismatches: %s' % (len(bad), '; '.join(bad[:12]))


def teardown_module(module):
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report.txt', 'w') as f:
        f.write('\n'.join(REPORT) + '\n')


def pair(dev, make_a, make_o, inputs, seed, tag, call=None, train=True, tol=TOL, golden=None):
    """Build the addk module and the oracle with identical parameters; compare eval output, train output,
    input gradients, parameter gradients and updated BN running statistics."""
    call = call or (lambda m, *a: m(*a))
    mo, ma = make_o(), make_a()
    fill_params(mo, seed)
    ma.load_state_dict(mo.state_dict())
    ma.to(dev)

    def to_dev(v, rg=False):
        if isinstance(v, (list, tuple)):
            return [to_dev(u, rg) for u in v]
        t = v.clone().to(dev)
        return t.requires_grad_(True) if rg else t

    mo.eval(); ma.eval()
    with torch.no_grad():
        yo = call(mo, *[i.clone() if torch.is_tensor(i) else [u.clone() for u in i] for i in inputs])
        ya = call(ma, *to_dev(inputs))
    _chk(tag + '/eval', ya, yo, tol)
    if golden is not None:
        _chk(tag + '/eval-vs-reference-golden', ya, torch.from_numpy(golden[0][golden[1] + '/eval']), tol)
    if not train:
        return
    fill_params(mo, seed)
    ma.load_state_dict(mo.state_dict())
    mo.train(); ma.train()

    def rg(v):
        if isinstance(v, (list, tuple)):
            return [rg(u) for u in v]
        return v.clone().requires_grad_(True)
    xo = [rg(i) for i in inputs]
    xa = to_dev(inputs, True)
    yo = call(mo, *xo)
    ya = call(ma, *xa)
    _chk(tag + '/train', ya, yo, tol)
    w = probe_weights(seed, tag, tuple(yo.shape))
    (yo * w).sum().backward()
    (ya * w.to(dev)).sum().backward()
    torch.cuda.synchronize()

    def flat(v):
        return [t for u in v for t in (flat(u) if isinstance(u, (list, tuple)) else [u])]
    for k, (ga, go) in enumerate(zip(flat(xa), flat(xo))):
        if go.grad is not None:
            if ga.grad is None:          # no gradient path (e.g. the 'none' primitive): the reference grad is exactly 0
                assert float(go.grad.abs().max()) == 0.0, tag + ' missing input grad %d' % k
                continue
            _chk(tag + '/gin%d' % k, ga.grad, go.grad, tol)
    pa = dict(ma.named_parameters())
    for n, p in mo.named_parameters():
        if p.grad is None:
            continue
        assert pa[n].grad is not None, '%s: missing grad for %s' % (tag, n)
        _chk(tag + '/g:' + n, pa[n].grad, p.grad, tol)
    ba = dict(ma.named_buffers())
    for n, b in mo.named_buffers():
        if n.endswith('num_batches_tracked'):
            assert int(ba[n]) == int(b), tag + ' ' + n
        else:
            _chk(tag + '/buf:' + n, ba[n], b, tol)


def test_mfma_fragment_layout(dev):
    """v_mfma_f32_16x16x4_f32 operand/accumulator lane maps, with an asymmetric B."""
    import addk
    lib = addk.load()
    out = torch.zeros(256, device=dev)
    assert lib.addk_selftest_mfma(out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    A = np.arange(64, dtype=np.float64).reshape(16, 4)
    B = np.array([[(k + 1) * (j + 2) for j in range(16)] for k in range(4)], dtype=np.float64)
    assert np.array_equal(out.cpu().numpy().reshape(16, 16), (A @ B).astype(np.float32))


@pytest.mark.parametrize('C,hw', [(8, (16, 32)), (20, (17, 33)), (40, (23, 31))])
@pytest.mark.parametrize('prim', oracle.PRIMITIVES)
def test_ops_registry(dev, golden, prim, C, hw):
    from addk.modeling.operations import OPS
    g = golden('ops')
    x = torch.from_numpy(g['x_C%d' % C]) if 'x_C%d' % C in g.files else rand_tensor(11, 'ops_x_%d' % C, (2, C) + hw)
    gold = (g, '%s_C%d' % (prim, C)) if '%s_C%d/eval' % (prim, C) in g.files else None
    pair(dev, lambda: OPS[prim](C, 1, BN, **KW), lambda: oracle.OPS[prim](C, 1, BN, **KW), [x], 100 + C,
         '%s_C%d' % (prim, C), golden=gold)


@pytest.mark.parametrize('prim', [p for p in oracle.PRIMITIVES if p != 'skip_connect'])
def test_ops_stride2(dev, golden, prim):
    from addk.modeling.operations import OPS
    g = golden('ops')
    pair(dev, lambda: OPS[prim](8, 2, BN, **KW), lambda: oracle.OPS[prim](8, 2, BN, **KW), [torch.from_numpy(g['x_s2'])],
         150, '%s_s2' % prim, golden=(g, '%s_s2' % prim))


def test_relu_conv_bn_and_reduce(dev, golden):
    from addk.modeling.operations import DoubleFactorizedReduce, FactorizedReduce, ReLUConvBN
    g = golden('ops')
    for ci, co in ((40, 24), (200, 40)):
        x = torch.from_numpy(g['rcb_x_%d' % ci])
        pair(dev, lambda: ReLUConvBN(ci, co, 1, 1, 0, BN, **KW), lambda: oracle.ReLUConvBN(ci, co, 1, 1, 0, BN, **KW), [x],
             200 + ci, 'rcb_%d_%d' % (ci, co), golden=(g, 'rcb_%d_%d' % (ci, co)))
    for h in (16, 17, 15):
        x = torch.from_numpy(g['fr_x_%d' % h])
        pair(dev, lambda: FactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1),
             lambda: oracle.FactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 300, 'fr_%d' % h, golden=(g, 'fr_%d' % h))
        pair(dev, lambda: DoubleFactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1),
             lambda: oracle.DoubleFactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 301, 'dfr_%d' % h, golden=(g, 'dfr_%d' % h))
    # odd channel counts (scalar path) and a large-K 1x1 with non-multiple-of-8 channels
    x = rand_tensor(14, 'rcb_odd', (2, 10, 9, 13))
    pair(dev, lambda: ReLUConvBN(10, 6, 1, 1, 0, BN, **KW), lambda: oracle.ReLUConvBN(10, 6, 1, 1, 0, BN, **KW), [x], 210, 'rcb_10_6')
    x = rand_tensor(14, 'rcb_k3', (2, 12, 11, 14))
    pair(dev, lambda: ReLUConvBN(12, 20, 3, 2, 1, BN, **KW), lambda: oracle.ReLUConvBN(12, 20, 3, 2, 1, BN, **KW), [x], 211, 'rcb_k3s2')


def test_bilinear_resize(dev, golden):
    """The resize kernels against F.interpolate golden vectors (forward and gather-form backward)."""
    from addk.module import AddkModule

    class Resize(AddkModule):
        def __init__(self, size):
            super().__init__()
            self.size = size

        def emit(self, g, x):
            return g.resize(x, *self.size)
    g = golden('bilinear')
    for k in ('down4', 'up_32_63', 'fit_63_64', 'up8', 'up_odd', 'down_odd'):
        size = tuple(int(v) for v in g[k + '/size'])
        x = torch.from_numpy(g[k + '/x']).to(dev).requires_grad_(True)
        y = Resize(size).to(dev)(x)
        _chk('bilinear/%s/y' % k, y, torch.from_numpy(g[k + '/y']), 1e-5)
        w = probe_weights(21, 'bil_' + k, tuple(y.shape)).to(dev)
        (y * w).sum().backward()
        _chk('bilinear/%s/gx' % k, x.grad, torch.from_numpy(g[k + '/gx']), 1e-5)


def test_heads(dev, golden):
    from addk.modeling.aspp_train import ASPP_train
    from addk.modeling.decoder import Decoder
    g = golden('heads')
    pair(dev, lambda: ASPP_train(40, 256, BN, mult=1), lambda: oracle.ASPP_train(40, 256, BN, mult=1),
         [torch.from_numpy(g['aspp40/x'])], 400, 'aspp40', golden=(g, 'aspp40'))
    pair(dev, lambda: ASPP_train(80, 256, BN, mult=2), lambda: oracle.ASPP_train(80, 256, BN, mult=2),
         [torch.from_numpy(g['aspp80_m2/x'])], 401, 'aspp80_m2', golden=(g, 'aspp80_m2'))
    pair(dev, lambda: ASPP_train(400, 256, BN, mult=1), lambda: oracle.ASPP_train(400, 256, BN, mult=1),
         [torch.from_numpy(g['aspp400/x'])], 402, 'aspp400', train=False, golden=(g, 'aspp400'))
    lo = torch.from_numpy(g['dec/low'])
    pair(dev, lambda: Decoder(19, BN), lambda: oracle.Decoder(19, BN), [torch.from_numpy(g['dec/x']), lo], 410, 'dec',
         call=lambda m, a, b: m(a, b, (33, 65)), golden=(g, 'dec'))
    pair(dev, lambda: Decoder(19, BN), lambda: oracle.Decoder(19, BN), [torch.from_numpy(g['dec_same/x']), lo], 411, 'dec_same',
         call=lambda m, a, b: m(a, b, (34, 66)), golden=(g, 'dec_same'))


def test_cells(dev, golden):
    from addk.modeling.ADD import Cell
    g = golden('cells')
    ga = torch.from_numpy(GENOTYPE_AUTODEEPLAB)
    ins = [torch.from_numpy(g['plain/pp']), torch.from_numpy(g['plain/p'])]
    for k, name in ((1, 'concat'), (2, 'dense')):
        pair(dev, lambda: Cell(BN, 5, 16, 32, ga, 1, 8, -1, dense_in=False, dense_out=True),
             lambda: oracle.Cell(BN, 5, 16, 32, ga, 1, 8, -1, dense_in=False, dense_out=True), ins, 500, 'plain_' + name,
             call=lambda m, a, b, k=k: m(a, b)[k], golden=(g, 'plain_' + name))
    ins = [[torch.from_numpy(g['densein/d%d' % i]) for i in range(3)], torch.from_numpy(g['densein/p'])]
    for k, name in ((1, 'concat'), (2, 'dense')):
        pair(dev, lambda: Cell(BN, 5, [8, 16, 8], 80, ga, 1, 8, 1, dense_in=True, dense_out=True),
             lambda: oracle.Cell(BN, 5, [8, 16, 8], 80, ga, 1, 8, 1, dense_in=True, dense_out=True), ins, 501, 'densein_' + name,
             call=lambda m, a, b, k=k: m(a, b)[k], golden=(g, 'densein_' + name))
    g40 = torch.from_numpy(GENOTYPE_40_1)
    ins = [[torch.from_numpy(g['last/d0']), torch.from_numpy(g['last/d1'])], torch.from_numpy(g['last/p'])]
    pair(dev, lambda: Cell(BN, 5, [8, 8], 40, g40, 1, 8, 0, dense_in=True, dense_out=False),
         lambda: oracle.Cell(BN, 5, [8, 8], 40, g40, 1, 8, 0, dense_in=True, dense_out=False), ins, 502, 'last',
         call=lambda m, a, b: m(a, b), golden=(g, 'last'))


def _build_add(dev, Fv, arch, seed=600, cls=None, ocls=None, genotype=GENOTYPE_AUTODEEPLAB, low=None):
    from addk.modeling.ADD import ADD
    cls, ocls = cls or ADD, ocls or oracle.ADD
    low = arch['low_level_layer'] if low is None else low
    mo = ocls(arch['network_arch'], arch['C_index'], genotype, 19, make_args(Fv), low)
    ma = cls(arch['network_arch'], arch['C_index'], genotype, 19, make_args(Fv), low)
    chk = fill_params(mo, seed)
    ma.load_state_dict(mo.state_dict())
    return ma.to(dev), mo, chk


@pytest.mark.parametrize('tag,Fv,arch', [('F4_65', 4, ARCH_C2), ('F4_64', 4, ARCH_C2), ('F4_C3_65', 4, ARCH_C3), ('F20_65', 20, ARCH_C2)])
def test_add_whole_net(dev, golden, tag, Fv, arch):
    g = golden('add')
    ma, mo, chk = _build_add(dev, Fv, arch)
    assert abs(chk - float(g[tag + '/chk'])) <= 1e-9 * chk
    x = torch.from_numpy(g[tag + '/x'])
    ma.eval(); mo.eval()
    with torch.no_grad():
        ya, yo = ma(x.to(dev)), mo(x)
    assert len(ya) == len(arch['C_index']) + 1
    for i, (a, o) in enumerate(zip(ya, yo)):
        _chk('%s/eval%d' % (tag, i), a, o)
        if tag + '/eval%d' % i in g.files:
            _chk('%s/eval%d-vs-reference-golden' % (tag, i), a, torch.from_numpy(g[tag + '/eval%d' % i]))
    if tag + '/loss' not in g.files:
        return
    # ---- training step.  Whole-network fp32 gradients are chaotic at these sizes: the REFERENCE arithmetic itself
    # (oracle fp32, same ATen kernels as the reference) sits 5-25 % from an fp64 evaluation of the same graph
    # (ReLU-mask flips amplified through 12 cells of small-batch BatchNorm; tests/tools/debug_whole_net.py).  An
    # elementwise 1e-3 bound against fp32 is therefore not a property the reference has; the bar here is that
    # the HIP path is as close to the fp64 truth as the reference's own fp32 path is (per-module gradients ARE
    # held to 1e-3 above, where the arithmetic is well conditioned).
    fill_params(mo, 600)
    ma.load_state_dict(mo.state_dict())
    m64 = type(mo)(arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(Fv), arch['low_level_layer'])
    m64.load_state_dict(mo.state_dict())
    m64.double()
    ma.train(); mo.train(); m64.train()
    tgt = torch.from_numpy(g[tag + '/target'].astype(np.int64))
    crit = nn.CrossEntropyLoss(weight=None, ignore_index=255)          # train.py:70,229-233 verbatim usage
    res = {}
    for name, m, xx, tt in (('o32', mo, x, tgt), ('o64', m64, x.double(), tgt), ('addk', ma, x.to(dev), tgt.to(dev))):
        ys = m(xx)
        loss = sum(crit(y, tt) for y in ys) / len(ys)
        loss.backward()
        res[name] = (ys, loss.item(), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    l32, l64, la = res['o32'][1], res['o64'][1], res['addk'][1]
    REPORT.append('%-60s o32 %.7f o64 %.7f addk %.7f golden %.7f' % (tag + '/loss', l32, l64, la, float(g[tag + '/loss'])))
    assert abs(la - l64) <= 3 * abs(l32 - l64) + 1e-4 * abs(l64)
    assert abs(la - float(g[tag + '/loss'])) < 1e-4 * abs(l64)
    for i in range(len(res['o64'][0])):
        e32 = rel_err(res['o32'][0][i], res['o64'][0][i])
        ea = rel_err(res['addk'][0][i], res['o64'][0][i])
        REPORT.append('%-60s o32-vs-fp64 %.3e  addk-vs-fp64 %.3e' % ('%s/train%d' % (tag, i), e32, ea))
        assert ea <= 3 * e32 + 1e-3
    g64 = res['o64'][2]
    assert set(res['addk'][2]) == set(g64)

    def rel_l2(ga):
        num = sum(float(((ga[k] - g64[k]) ** 2).sum()) for k in g64)
        return (num / sum(float((g64[k] ** 2).sum()) for k in g64)) ** 0.5

    def cos(ga):
        dot = sum(float((ga[k] * g64[k]).sum()) for k in g64)
        return dot / (sum(float((ga[k] ** 2).sum()) for k in g64) ** 0.5 * sum(float((g64[k] ** 2).sum()) for k in g64) ** 0.5)
    e32, ea = rel_l2(res['o32'][2]), rel_l2(res['addk'][2])
    REPORT.append('%-60s o32-vs-fp64 %.3e  addk-vs-fp64 %.3e  cos %.6f / %.6f' % (tag + '/grad rel-L2 (all params)', e32, ea,
                                                                               cos(res['o32'][2]), cos(res['addk'][2])))
    # the fp32-vs-fp64 spread itself varies 1.5e-2 .. 1.5e-1 between inputs and sizes, so one input says little about it:
    # the magnitude bound is held as a statistic over several inputs in tests/test_gpu_configs.py
    # (test_train_mode_gradient_spread_is_the_references_own: median within 2x, maximum within 3x of the fp32 oracle's own);
    # here the gradient direction must agree with fp64
    assert cos(res['addk'][2]) >= 0.99
    ba = dict(ma.named_buffers())
    for n, b in m64.named_buffers():
        if not n.endswith('num_batches_tracked'):
            _chk('%s/buf:%s' % (tag, n), ba[n], b, 2e-3)
        else:
            assert int(ba[n]) == int(b), n


@pytest.mark.parametrize('Fv,hw', [(4, (65, 129)), (20, (512, 1024))], ids=['F4_65x129', 'F20_512x1024'])
def test_add_whole_net_frozen_bn_gradients(dev, Fv, hw):
    """Whole-network wiring of the backward pass (dense connections, shared heads, stems, low-level path) in a
    well-conditioned setting: BatchNorm frozen (model.eval(), running statistics) removes the small-batch
    amplification, so conv-weight gradients can be held elementwise against the fp32 oracle.  The 512x1024 case is the
    one whose maps are large enough (>= 8192 pixels at the cell levels) to run the halo-patch, register-streaming and
    LDS-tiled kernels inside the real network.  Even there the network amplifies any 1e-7 perturbation of the stems to
    1e-4..1e-3 in the gradients (the fp32 oracle sits 2e-4 median .. 3e-3 from fp64), and how a given arithmetic fares is
    a matter of the input draw (profiles/r02_split_threshold_study.txt): the large case therefore runs TWO draws and holds the
    maximum / median error to 3x / 2.5x the fp32 oracle's own distance to fp64 per draw, to 2x / 1.75x in the geometric mean and every
    gradient elementwise to 1e-2 — round 1's bounds again, which round 2's kernels needed loosened to 5x / 4x / 2e-2:
    with blocked accumulation in the <= 64-channel split kernel and the generic kernel, and stem2 on the split kernel, the default
    bf16x6 mode measures 1.36 / 1.13 and 0.65 / 0.83 (round 2: 2.8 / 2.2 and 0.96 / 0.72).  The oracle's own figures move by ~10 % from
    run to run with the CPU's thread scheduling, ours are bit-reproducible."""
    big = hw[0] >= 512
    crit = nn.CrossEntropyLoss(ignore_index=255)
    ratios = []
    for draw, (sx, st) in enumerate([(61, 62), (71, 72)] if big else [(61, 62)]):
        ma, mo, _chk_sum = _build_add(dev, Fv, ARCH_C2)
        ma.eval(); mo.eval()
        x = rand_tensor(sx, 'frozen_x', (2, 3) + hw)
        tgt = torch.from_numpy(np.random.default_rng(st).integers(0, 19, (2,) + hw)).long()
        ya = ma(x.to(dev)); yo = mo(x)
        (sum(crit(y, tgt.to(dev)) for y in ya) / 2).backward()
        (sum(crit(y, tgt) for y in yo) / 2).backward()
        torch.cuda.synchronize()
        pa = dict(ma.named_parameters())
        if big:
            # [r4] fp64 truth = the REAL reference run in double precision, held sub-sampled by tests/golden/grads64.npz (no fp64 pass on
            # the host): max-abs error over the held positions / the whole tensor's max |g64|, and the rms error beside it (every element
            # counts there, not only the largest); the fp32 oracle runs live and is measured the same way
            from grads64_util import Grads64
            fx = Grads64('frozen512_%d' % draw)
            assert abs(_chk_sum - fx.chk) <= 1e-9 * abs(fx.chk), 'weights differ from the ones the fixture was made with'
        n = 0
        ours, theirs, groups, ours_rms, theirs_rms = [], [], {}, [], []
        for k, p in mo.named_parameters():
            if p.dim() == 4 and p.grad is not None:
                assert pa[k].grad is not None, k
                if big:
                    ours.append(fx.rel_err(k, pa[k].grad)); theirs.append(fx.rel_err(k, p.grad))
                    ours_rms.append(fx.rms_err(k, pa[k].grad)); theirs_rms.append(fx.rms_err(k, p.grad))
                    REPORT.append('%-60s %.3e (rms %.3e)' % ('frozen_bn%d.%d/g:%s vs reference fp64' % (hw[0], draw, k), ours[-1], ours_rms[-1]))
                    assert ours[-1] <= 1e-2, (k, ours[-1])
                    _chk('frozen_bn%d.%d/g-full:%s' % (hw[0], draw, k), pa[k].grad, p.grad, 1e-2)      # every element, against the live fp32 oracle
                    grp = '.'.join(k.split('.')[:2]) if k.startswith('cells.') else k.split('.')[0]
                    groups.setdefault(grp, []).append((ours[-1], theirs[-1]))
                else:
                    _chk('frozen_bn%d/g:%s' % (hw[0], k), pa[k].grad, p.grad, 2e-3)
                n += 1
        assert n > 400
        if big:
            med = lambda v: sorted(v)[len(v) // 2]
            REPORT.append('frozen_bn512 draw %d vs fp64: addk max %.2e median %.2e | fp32 oracle max %.2e median %.2e' % (
                draw, max(ours), med(ours), max(theirs), med(theirs)))
            for grp, v in groups.items():      # per module: median error of its conv-weight gradients, addk and the fp32 oracle, both vs fp64
                REPORT.append('frozen_bn512 draw %d per-layer %-22s n=%3d  addk %.2e  fp32 oracle %.2e  ratio %.2f' % (
                    draw, grp, len(v), med([a for a, _ in v]), med([b for _, b in v]), med([a for a, _ in v]) / max(med([b for _, b in v]), 1e-30)))
            REPORT.append('frozen_bn512 draw %d rms error vs fp64: addk max %.2e median %.2e | fp32 oracle max %.2e median %.2e' % (
                draw, max(ours_rms), med(ours_rms), max(theirs_rms), med(theirs_rms)))
            ratios.append((max(ours) / max(theirs), med(ours) / med(theirs)))
            assert ratios[-1][0] <= 3 and ratios[-1][1] <= 2.5, ratios
            assert max(ours_rms) <= 3 * max(theirs_rms) and med(ours_rms) <= 2.5 * med(theirs_rms), (max(ours_rms), max(theirs_rms), med(ours_rms), med(theirs_rms))
        del ma, mo
        torch.cuda.empty_cache()
    if big:
        gm = lambda v: float(np.exp(np.mean(np.log(v))))
        assert gm([r[0] for r in ratios]) <= 2 and gm([r[1] for r in ratios]) <= 1.75, ratios


def test_tail_x3_mode_holds_the_frozen_bn_gradient_gate(dev):
    """`tail_x3` (three product terms in the exit heads — ASPP and decoder forward, data and weight gradients — six everywhere else; VERDICT
    r03 item 8): the 2x512x1024 frozen-BatchNorm gate at its UNCHANGED bounds (max / median of 440 conv-weight gradients within 3x / 2.5x the
    fp32 oracle's own distance to the reference's fp64 per draw, every gradient elementwise to 1e-2).  The heads sit at the end of the
    network, where nothing amplifies their rounding; the reference's own GPU path is 16-bit throughout (apex O1, train.py:145-165)."""
    import addk
    prev = addk.get_precision()
    try:
        addk.set_precision('tail_x3')
        test_add_whole_net_frozen_bn_gradients(dev, 20, (512, 1024))
    finally:
        addk.set_precision(prev)


@pytest.mark.parametrize('gname', ['genotype_1', 'genotype_2'])
def test_add_f40_config5_eval(dev, gname):
    """BASELINE config 5 architecture (F=40, searched_arch/40_5e_38_lr genotypes, three unsorted pairs: Q1), eval forward
    against the oracle at a small input."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'searched_arch', '40_5e_38_lr', gname + '.npy'))
    ma, mo, _ = _build_add(dev, 40, ARCH_C2, seed=900, genotype=g)
    ma.eval(); mo.eval()
    x = rand_tensor(91, 'f40_x', (1, 3, 65, 129))
    with torch.no_grad():
        ya, yo = ma(x.to(dev)), mo(x)
    for i, (a, o) in enumerate(zip(ya, yo)):
        _chk('F40_%s/eval%d' % (gname, i), a, o)


def test_dynamic_inference_and_entropy(dev, golden):
    from addk.modeling.ADD import EDM
    from addk.modeling.operations import normalized_shannon_entropy
    g = golden('dynamic')
    ma, mo, chk = _build_add(dev, 20, ARCH_C2, seed=700)
    eo = oracle.EDM()
    fill_params(eo, 701)
    ea = EDM()
    ea.load_state_dict(eo.state_dict())
    ea.to(dev).eval()
    ma.eval(); mo.eval(); eo.eval()
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        y, feat = ma.get_feature(x.to(dev))
        _chk('get_feature/logits', y, torch.from_numpy(g['get_feature/logits']))
        _chk('get_feature/feature', feat, torch.from_numpy(g['get_feature/feature']))
        _chk('edm(feature)', ea(feat.clone()), torch.from_numpy(g['edm_on_feature']))
        for name, thr in (('early', 1e9), ('final', -1e9)):
            y, ee, secs, conf = ma.dynamic_inference(x.to(dev), threshold=thr, confidence='edm', edm=ea)
            assert ee == int(g[name + '/exit']) and secs > 0
            _chk('dynamic/%s/logits' % name, y, torch.from_numpy(g[name + '/logits']))
            _chk('dynamic/%s/conf' % name, conf, torch.from_numpy(g[name + '/conf']))
        ys = ma(x.to(dev))
        assert abs(normalized_shannon_entropy(ys[0]) - float(g['entropy0'])) < 1e-4
        assert abs(normalized_shannon_entropy(ys[1]) - float(g['entropy1'])) < 1e-4


def test_baseline_model_config1(dev, golden):
    from addk.modeling.baseline_model import Baselin_Model
    g = golden('baseline')
    arch = dict(network_arch=NETWORK_PATH_BASELINE, C_index=[5], low_level_layer=1)
    ma, mo, chk = _build_add(dev, 20, arch, seed=800, cls=Baselin_Model, ocls=oracle.Baselin_Model, genotype=GENOTYPE_BASELINE_2)
    ma.eval()
    with torch.no_grad():
        ys = ma(torch.from_numpy(g['129/x']).to(dev))
        _chk('baseline/129/last', ys[-1], torch.from_numpy(g['129/last']))
        _chk('baseline/129/first', ys[0], torch.from_numpy(g['129/first']))
        ys = ma(rand_tensor(81, 'base_x513', (1, 3, 513, 513)).to(dev))
        _chk('baseline/513/last_sub8', ys[-1][:, :, ::8, ::8], torch.from_numpy(g['513/last_sub8']))
        agree = (ys[-1].argmax(1)[:, ::4, ::4].cpu().numpy() == g['513/argmax_sub4']).mean()
        assert agree > 0.999
