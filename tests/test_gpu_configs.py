"""GPU parity at the BASELINE configurations the toy-sized suite did not reach:

 * the C=4 network (train.py:84-87: three 1x1 `conv_aspp` adapters) and config 5's F=40 architecture (both
   searched_arch/40_5e_38_lr genotypes) — eval logits, one train-mode step and frozen-BN gradients incl. backward;
 * config 2 at its full 2x1024x2048 shape (eval logits, first-step training loss) and config 4's dynamic inference at
   1x1024x2048 and 1x1025x2049, against the CPU oracle on the same seeded inputs;
 * the train-mode whole-network gradient spread as a STATISTIC over several inputs (replaces the single-case 0.12 floor).

Tolerance: 1e-3 relative fp32 (max-abs error / max-abs reference) unless a test states a different, measured bound."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import oracle                       # noqa: E402  (the checker)
from _util import (ARCH_C2, ARCH_C4, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor, rel_err)   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = []


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    import addk
    addk.load()
    return torch.device('cuda:0')


def teardown_module(module):
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_configs.txt', 'w') as f:
        f.write('\n'.join(REPORT) + '\n')


def _log(fmt, *a):
    REPORT.append(fmt % a)


def _geno(name):
    if name == 'autodeeplab':
        return GENOTYPE_AUTODEEPLAB
    return np.load(os.path.join(ROOT, 'searched_arch', '40_5e_38_lr', name + '.npy'))


def _build(dev, Fv, arch, geno, seed=600):
    from addk.modeling.ADD import ADD
    args = (arch['network_arch'], arch['C_index'], geno, 19, make_args(Fv), arch['low_level_layer'])
    mo = oracle.ADD(*args)
    chk = fill_params(mo, seed)
    ma = ADD(*args)
    ma.load_state_dict(mo.state_dict())
    return ma.to(dev), mo, chk, args


def _target(hw, n=2, seed=62):
    t = torch.from_numpy(np.random.default_rng(seed).integers(0, 19, (n,) + hw)).long()
    t[torch.from_numpy(np.random.default_rng(seed + 1).random((n,) + hw) < 0.05)] = 255
    return t


def _check_big(g, key, t, tol, what):
    t = t.detach().float().cpu()
    if key in g.files:
        e = rel_err(t, torch.from_numpy(g[key]))
    else:
        e = rel_err(t.reshape(-1)[::97], torch.from_numpy(g[key + '@sub97']))
        n = float(t.double().norm())
        assert abs(n - float(g[key + '@norm'])) <= tol * float(g[key + '@norm']), what + ' norm'
    _log('%-60s %.3e', what, e)
    assert e <= tol, '%s: %.3e > %.1e' % (what, e, tol)


CASES = {'F4_C4_65': (4, ARCH_C4, 'autodeeplab'), 'F40_g1_65': (40, ARCH_C2, 'genotype_1'), 'F40_g2_65': (40, ARCH_C2, 'genotype_2')}


@pytest.mark.parametrize('tag', list(CASES))
def test_add_configs_eval_and_train_step(dev, golden, tag):
    """Eval logits vs the oracle AND the reference golden; one train-mode step: loss vs the reference's, per-exit logits
    as close to an fp64 evaluation as the fp32 oracle is (the train-mode forward amplifies rounding)."""
    g = golden('configs')
    Fv, arch, gname = CASES[tag]
    ma, mo, chk, args = _build(dev, Fv, arch, _geno(gname))
    assert abs(chk - float(g[tag + '/chk'])) <= 1e-9 * chk
    x = rand_tensor(61, 'add_x_' + tag, (2, 3, 65, 129))
    ma.eval(); mo.eval()
    with torch.no_grad():
        ya, yo = ma(x.to(dev)), mo(x)
    assert len(ya) == len(arch['C_index']) + 1
    for i, (a, o) in enumerate(zip(ya, yo)):
        e = rel_err(a, o)
        _log('%-60s %.3e', '%s/eval%d vs oracle' % (tag, i), e)
        assert e <= 1e-3
        _check_big(g, '%s/eval%d' % (tag, i), a, 1e-3, '%s/eval%d vs reference golden' % (tag, i))
    fill_params(mo, 600)
    ma.load_state_dict(mo.state_dict())
    ma.train(); mo.train()
    tgt = _target((65, 129))
    crit = nn.CrossEntropyLoss(ignore_index=255)
    # [r4] the fp64 evaluation is the REAL reference's, run in double precision in the build container and held by
    # tests/golden/grads64.npz (loss, per-exit logits every 97th element, sub-sampled gradients): no fp64 pass on the host
    import numpy as np
    from grads64_util import Grads64, PATH
    fx = Grads64('cfg_' + tag)
    z = np.load(PATH)
    res = {}
    for name, m, xx, tt in (('o32', mo, x, tgt), ('addk', ma, x.to(dev), tgt.to(dev))):
        ys = m(xx)
        loss = sum(crit(y, tt) for y in ys) / len(ys)
        loss.backward()
        res[name] = (ys, loss.item(), {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None})
    torch.cuda.synchronize()
    l32, l64, la, lg = res['o32'][1], fx.loss64, res['addk'][1], float(g[tag + '/loss'])
    _log('%-60s o32 %.7f reference fp64 %.7f addk %.7f reference fp32 %.7f', tag + '/loss', l32, l64, la, lg)
    assert abs(la - lg) < 1e-4 * abs(lg)
    assert abs(la - l64) <= 3 * abs(l32 - l64) + 1e-4 * abs(l64)
    for i in range(len(res['o32'][0])):
        y64 = torch.from_numpy(z['cfg_%s/logits64_%d@sub97' % (tag, i)]).double()
        mx = float(z['cfg_%s/logits64_%d@maxabs' % (tag, i)])
        e32 = float((res['o32'][0][i].detach().double().reshape(-1)[::97] - y64).abs().max()) / mx
        ea = float((res['addk'][0][i].detach().double().cpu().reshape(-1)[::97] - y64).abs().max()) / mx
        _log('%-60s o32-vs-fp64 %.3e  addk-vs-fp64 %.3e', '%s/train%d' % (tag, i), e32, ea)
        assert ea <= 3 * e32 + 1e-3
    ga = res['addk'][2]
    assert set(ga) == set(fx.names())        # every parameter of the architecture received a gradient (backward wiring)
    cos = fx.cos(ga)
    _log('%-60s cos(addk, reference fp64) %.6f   (fp32 oracle: %.6f)', tag + '/grad', cos, fx.cos(res['o32'][2]))
    assert cos >= 0.99


@pytest.mark.parametrize('gname', ['genotype_1', 'genotype_2'])
def test_f40_frozen_bn_gradients(dev, gname):
    """Config 5's architecture, backward included, at 2x256x512 (cell maps >= 8k pixels: the large-map kernels engage):
    BatchNorm frozen, every conv-weight gradient held against the fp64 oracle relative to the fp32 oracle's own error.

    The statistic is the DISTRIBUTION over the ~500 gradients (median, 90th percentile, maximum), each against the fp32 oracle's own.
    The maximum alone is not a stable quantity even with BatchNorm frozen: the fp32 oracle's own maximum moves from 3.5e-3 to 1.1e-2
    (x3.1, p90 x1.9) when its INPUT is nudged by one ulp (tests/tools/f40_frozen_probe.py, profiles/r03_f40_frozen_bn_sensitivity_probe.txt:
    ReLU / max-pool decisions near ties flip and a cluster of level-1 cells moves together), and addk showed the same cluster —
    max 1.3e-2 with stem2 on the generic fp32-MFMA kernel, 2.2e-3 (below the oracle) with stem2 on the split kernel, both kernels
    individually within 2e-5 of fp64 per launch (test_gpu_fast_kernels.py).  A wrong kernel moves its layers by O(1), not by x3."""
    hw = (256, 512)
    ma, mo, _chk_sum, args = _build(dev, 40, ARCH_C2, _geno(gname), seed=900)
    ma.eval(); mo.eval()
    x = rand_tensor(61, 'f40_frozen_x', (2, 3) + hw)
    tgt = _target(hw)
    crit = nn.CrossEntropyLoss(ignore_index=255)
    ya = ma(x.to(dev)); yo = mo(x)
    for i, (a, o) in enumerate(zip(ya, yo)):
        e = rel_err(a, o)
        _log('%-60s %.3e', 'F40_%s_256x512/eval%d vs oracle' % (gname, i), e)
        assert e <= 1e-3
    (sum(crit(y, tgt.to(dev)) for y in ya) / 2).backward()
    (sum(crit(y, tgt) for y in yo) / 2).backward()
    torch.cuda.synchronize()
    # [r4] fp64 truth: the real reference in double precision, held by tests/golden/grads64.npz (no fp64 pass on the host)
    from grads64_util import Grads64
    fx = Grads64('f40_' + gname)
    assert abs(_chk_sum - fx.chk) <= 1e-9 * abs(fx.chk), 'weights differ from the ones the fixture was made with'
    pa = dict(ma.named_parameters())
    ours, theirs, ours_rms, theirs_rms = [], [], [], []
    for k, p in mo.named_parameters():
        if p.dim() == 4 and p.grad is not None:
            assert pa[k].grad is not None, k
            ours.append(fx.rel_err(k, pa[k].grad)); theirs.append(fx.rel_err(k, p.grad))
            ours_rms.append(fx.rms_err(k, pa[k].grad)); theirs_rms.append(fx.rms_err(k, p.grad))
    assert len(ours) > 400
    med = lambda v: sorted(v)[len(v) // 2]
    p90 = lambda v: sorted(v)[int(len(v) * 0.9)]
    _log('F40_%s frozen-BN 256x512, %d conv-weight gradients vs fp64: addk max %.2e p90 %.2e median %.2e | fp32 oracle max %.2e p90 %.2e median %.2e',
         gname, len(ours), max(ours), p90(ours), med(ours), max(theirs), p90(theirs), med(theirs))
    assert med(ours) <= max(2 * med(theirs), 2e-4) and p90(ours) <= max(5 * p90(theirs), 1e-3) and max(ours) <= max(10 * max(theirs), 2e-3)
    _log('F40_%s frozen-BN 256x512 rms error vs fp64: addk max %.2e p90 %.2e median %.2e | fp32 oracle max %.2e p90 %.2e median %.2e',
         gname, max(ours_rms), p90(ours_rms), med(ours_rms), max(theirs_rms), p90(theirs_rms), med(theirs_rms))
    # rms metric (every element counts): measured with the oracle on 16 threads (= the reference's own fp32 in the fixture: max 1.21e-3 p90 2.43e-4
    # median 7.3e-5 for genotype_1) addk sits at 2.17x the oracle's median for genotype_1 and 1.06x for genotype_2 — this architecture's gradients
    # move by x2-3 under a one-ulp input nudge (docstring), so the median bound is 2.5x here, the tail bounds as for the max-abs metric
    assert med(ours_rms) <= max(2.5 * med(theirs_rms), 2e-4) and p90(ours_rms) <= max(5 * p90(theirs_rms), 1e-3) and max(ours_rms) <= max(10 * max(theirs_rms), 2e-3)


@pytest.mark.parametrize('hw', [(65, 129), (64, 128)], ids=['odd65x129', 'even64x128'])
def test_train_mode_gradient_spread_is_the_references_own(dev, hw):
    """Train-mode whole-network gradients are ill-conditioned in the REFERENCE arithmetic itself: the fp32 oracle sits
    5e-2..1.5e-1 (rel-L2 over all parameters) from an fp64 evaluation of the same graph, at every batch / map size tried
    (2x65x129 5.1e-2, 2x129x257 1.1e-1, 4x129x257 9.5e-2, 2x257x513 1.5e-1, 8x65x129 6.8e-2: no well-conditioned size
    exists; the gradient norm grows 300x from the heads to the stems through 12 cells of small-batch BatchNorm).  A
    single input therefore says little (round 1's F4_64 case: addk 6.1e-2 vs fp32 1.5e-2 was one draw from this spread).
    The bound here is on the DISTRIBUTION over several inputs: addk's error against fp64 must look like the fp32
    oracle's own — median within 1.6x, geometric mean of the per-draw ratios within 2x — and the gradient direction must be as good
    as the fp32 oracle's.

    The even size is the better-conditioned one (its early resizes have lambda = 0.5 exactly, so the reference's own fp32 error
    sits at its floor, 1-2e-2).  Round 2's kernels were 1.2-2.9x the oracle there on every draw; round 3 traced it layer by layer
    (tests/tools/even_size_study.py --trace, profiles/r03_even_size_trace.txt): NOT the lazy-BatchNorm cancellation the r02 verdict
    suspected (the centred backward form changes no digit) but the forward error of stem1 (3x3, K = 576: 3.6e-7 vs the oracle's
    1.9e-7), inherited at 1.5-1.9x by every later layer — a k-ordered fp32 accumulation chain against the CPU library's blocked
    sums.  With blocked accumulation in the narrow split-bf16 kernel and the generic conv kernel every layer is back at 0.8-1.1x
    the oracle's forward error (profiles/r03_even_size_trace_blocked_accumulation.txt) and the MEDIAN gradient error equals the
    oracle's (1.58e-2 vs 1.57e-2).  The per-draw ratio is chaotic, though: 0.95 0.36 2.02 8.16 — the fp32 oracle itself moves by
    x5 on ONE draw when its input is nudged by 6e-8 (8.2e-3 .. 4.1e-2, profiles/r03_even_size_chaos_probe.txt), and draw 3's
    excess is born in the last cell / the 2-sample image-pool BatchNorm of the second exit, where every earlier layer is at or
    below the oracle's error (profiles/r03_even_size_draw3_forward_and_gradient_trace.txt).  Hence: the median and the geometric
    mean of the ratios are held tight, a single draw may sit up to 10x out."""
    from addk.modeling.ADD import ADD
    from grads64_util import Grads64
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4), 0)
    crit = nn.CrossEntropyLoss(ignore_index=255)
    e32s, eas, coss, cos32, eref = [], [], [], [], []
    # [r4] the fp64 truth is held by the REFERENCE (tests/golden/grads64.npz: the real reference run in double precision, sub-sampled,
    # tests/golden/make_golden_fp64.py) — no fp64 pass on the host any more, and every even-size draw runs by default again (the fourth,
    # round 3's 8x outlier, had gone behind ADDK_LONG_TESTS for suite time).  rel-L2 and cosine are taken over the held positions
    # (<= 128 per tensor, 64 579 of ~2.5 M elements), for addk and the live fp32 oracle alike.
    ndraw = 4 if hw[0] % 2 == 0 else 2
    for k in range(ndraw):
        fx = Grads64('spread_%dx%d_%d' % (hw + (k,)))
        mo = oracle.ADD(*args)
        assert abs(fill_params(mo, 600 + k) - fx.chk) <= 1e-9 * abs(fx.chk), 'weights differ from the ones the fixture was made with'
        ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
        x = rand_tensor(170 + k, 'spread_x', (2, 3) + hw)
        tgt = _target(hw, seed=180 + 2 * k)
        grads = {}
        for name, m, xx, tt in (('o32', mo, x, tgt), ('addk', ma, x.to(dev), tgt.to(dev))):
            m.train()
            ys = m(xx)
            (sum(crit(y, tt) for y in ys) / len(ys)).backward()
            grads[name] = {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
        assert set(grads['addk']) == set(fx.names()) == set(grads['o32'])
        e32s.append(fx.rel_l2(grads['o32'])); eas.append(fx.rel_l2(grads['addk'])); eref.append(fx.ref32_rel_l2())
        coss.append(fx.cos(grads['addk'])); cos32.append(fx.cos(grads['o32']))
    _log('train-mode gradient rel-L2 vs the reference\'s fp64 at %s: the reference\'s own fp32 (8 threads, build container) %s', hw, ' '.join('%.2e' % v for v in eref))
    med = lambda v: sorted(v)[len(v) // 2]
    _log('train-mode gradient rel-L2 vs fp64 over %d inputs at %s: fp32 oracle %s | addk %s | cos addk %s | cos fp32 oracle %s', ndraw, hw,
         ' '.join('%.2e' % v for v in e32s), ' '.join('%.2e' % v for v in eas), ' '.join('%.5f' % v for v in coss),
         ' '.join('%.5f' % v for v in cos32))
    ratios = [a / o for a, o in zip(eas, e32s)]
    gmean = float(np.exp(np.mean(np.log(ratios))))
    _log('train-mode gradient ratios addk / fp32 oracle at %s: %s   median %.2f  geometric mean %.2f', hw, ' '.join('%.2f' % r for r in ratios),
         med(eas) / med(e32s), gmean)
    assert med(eas) <= 1.6 * med(e32s), (eas, e32s)
    assert gmean <= 2.0, (ratios, gmean)
    assert max(ratios) <= 10.0, (eas, e32s)
    # direction: as good as the fp32 oracle's own on every input (some weight draws are chaotic for the reference too:
    # measured rel-L2 0.9 / cos 0.7 for BOTH on 1 of the 4 draws, 2 of 6), and >= 0.99 wherever the reference manages that
    for ca, c32 in zip(coss, cos32):
        assert ca >= min(0.99, c32 - 0.05), (coss, cos32)


def _bench_model(dev, seed=1):
    """The model bench.py times: ADD searched-dense C=2 F=20, weights from fill_params (name-keyed, identical in the oracle)."""
    ma, mo, _, args = _build(dev, 20, ARCH_C2, GENOTYPE_AUTODEEPLAB, seed=1000 + seed)
    return ma, mo


def test_full_size_config2_eval_logits_and_first_step_loss(dev):
    """BASELINE config 2 at its real shape, 2x1024x2048: eval-mode logits of both exits (subsampled every 8th pixel + the
    L2 norm of the whole tensor) and the first training step's loss against the CPU oracle."""
    hw = (1024, 2048)
    ma, mo = _bench_model(dev)
    x = rand_tensor(201, 'full_x', (2, 3) + hw)
    ma.eval(); mo.eval()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        ya = ma(x.to(dev))
        ya = [y.cpu() for y in ya]
        yo = mo(x)
    for i, (a, o) in enumerate(zip(ya, yo)):
        assert tuple(a.shape) == (2, 19) + hw
        e = rel_err(a[:, :, ::8, ::8], o[:, :, ::8, ::8])
        en = abs(float(a.double().norm()) - float(o.double().norm())) / float(o.double().norm())
        _log('%-60s sub8 %.3e  norm %.3e', 'config2 2x1024x2048 eval logits exit %d vs oracle' % i, e, en)
        assert e <= 1e-3 and en <= 1e-4
    del ya, yo
    ma.train(); mo.train()
    tgt = _target(hw)
    crit = nn.CrossEntropyLoss(ignore_index=255)
    with torch.no_grad():
        lo = sum(crit(y, tgt) for y in mo(x)) / 2
    from addk.loss import CrossEntropyLoss
    ca = CrossEntropyLoss(ignore_index=255)
    with torch.no_grad():
        ys = ma(x.to(dev))
        la = sum(ca(y, tgt.to(dev)) for y in ys) / 2
    _log('%-60s oracle %.7f addk %.7f', 'config2 2x1024x2048 train-mode first-step loss', float(lo), float(la))
    assert abs(float(la) - float(lo)) <= 1e-4 * abs(float(lo))


@pytest.mark.parametrize('hw', [(1024, 2048), (1025, 2049)], ids=['1024x2048', '1025x2049'])
def test_full_size_config4_dynamic_inference(dev, hw):
    """BASELINE config 4 at its real shapes (bs=1): EDM-gated dynamic inference, early and final exit, against the oracle
    (reference ADD.py:379-438; 1025x2049 is the reference's padded eval size, 1024x2048 exercises the even-size quirk Q8)."""
    from addk.modeling.ADD import EDM
    ma, mo = _bench_model(dev, seed=2)
    eo = oracle.EDM(); fill_params(eo, 701)
    ea = EDM(); ea.load_state_dict(eo.state_dict()); ea.to(dev).eval()
    ma.eval(); mo.eval(); eo.eval()
    x = rand_tensor(202, 'dyn_full_x', (1, 3) + hw)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        for name, thr in (('early', 1e9), ('final', -1e9)):
            ya, ee_a, secs, conf_a = ma.dynamic_inference(x.to(dev), threshold=thr, confidence='edm', edm=ea)
            ya = ya.cpu()
            yo, ee_o, _, conf_o = mo.dynamic_inference(x, threshold=thr, confidence='edm', edm=eo)
            assert ee_a == ee_o == (1 if name == 'early' else 0)
            e = rel_err(ya[:, :, ::8, ::8], yo[:, :, ::8, ::8])
            ec = rel_err(conf_a, conf_o)
            _log('%-60s logits sub8 %.3e  confidence %.3e  (%.1f ms)', 'config4 %dx%d dynamic %s exit vs oracle' % (hw + (name,)), e, ec, secs * 1e3)
            assert tuple(ya.shape) == (1, 19) + hw and e <= 1e-3 and ec <= 1e-3


@pytest.mark.skipif(__import__('os').environ.get('ADDK_STUDY') != '1', reason='measurement study (ADDK_STUDY=1), not a parity gate')
def test_study_split_threshold_over_input_draws(dev):
    """Whole-network frozen-BN conv-weight gradients at 2x512x1024 over several input draws, for three arithmetics of the
    halo kernels: exact fp32, split-bf16 above 64 output channels only (round 2's first rule) and split-bf16 everywhere (shipped).  The network
    amplifies any 1e-7 perturbation of the stems to 1e-4..1e-3 in the gradients (the fp32 oracle itself sits there), so one
    draw cannot rank the arithmetics; the report lists median and maximum error vs fp64 relative to the fp32 oracle's."""
    import addk
    import addk._lib as L
    from test_gpu_parity import _build_add
    lib = L.load()
    hw = (512, 1024)
    crit = nn.CrossEntropyLoss(ignore_index=255)
    rows = []
    for draw in range(int(__import__('os').environ.get('ADDK_STUDY_DRAWS', '3'))):
        x = rand_tensor(900 + draw, 'study_x', (2, 3) + hw)
        tgt = torch.from_numpy(np.random.default_rng(950 + draw).integers(0, 19, (2,) + hw)).long()
        _, mo, _ = _build_add(dev, 20, ARCH_C2)
        mo.eval()
        (sum(crit(y, tgt) for y in mo(x)) / 2).backward()
        m64 = oracle.ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), ARCH_C2['low_level_layer']).double()
        m64.load_state_dict(mo.state_dict()); m64.eval()
        (sum(crit(y, tgt) for y in m64(x.double())) / 2).backward()
        p64 = dict(m64.named_parameters())
        keys = [k for k, p in mo.named_parameters() if p.dim() == 4 and p.grad is not None]
        theirs = sorted(rel_err(dict(mo.named_parameters())[k].grad.double(), p64[k].grad) for k in keys)
        for name, prec, minc in (('fp32', 'fp32', 65), ('wide>64', 'bf16x6', 65), ('all', 'bf16x6', 0)):
            addk.set_precision(prec)
            lib.addk_set_split_min_channels(minc)
            ma, _, _ = _build_add(dev, 20, ARCH_C2)
            ma.eval()
            (sum(crit(y, tgt.to(dev)) for y in ma(x.to(dev))) / 2).backward()
            torch.cuda.synchronize()
            pa = dict(ma.named_parameters())
            ours = sorted(rel_err(pa[k].grad.cpu().double(), p64[k].grad) for k in keys)
            rows.append('draw %d %-8s median %.2e (%.2f x oracle)  max %.2e (%.2f x oracle)' % (
                draw, name, ours[len(ours) // 2], ours[len(ours) // 2] / theirs[len(theirs) // 2], ours[-1], ours[-1] / theirs[-1]))
            del ma
            torch.cuda.empty_cache()
    addk.set_precision('bf16x6'); lib.addk_set_split_min_channels(-1)
    REPORT.extend(rows)
    print('\n'.join(rows))
