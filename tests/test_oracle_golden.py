"""Pin the CPU oracle (oracle/) against golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only; no GPU, no /root/reference."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle
from _util import (ARCH_C2, ARCH_C3, GENOTYPE_AUTODEEPLAB, GENOTYPE_BASELINE_2, GENOTYPE_40_1,
                   NETWORK_PATH_BASELINE, fill_params, make_args, probe_weights, rand_tensor)

BN = nn.BatchNorm2d
KW = dict(eps=1e-5, momentum=0.1, affine=True)
TOL = 2e-5        # oracle and reference run the same ATen kernels; only summation order may differ


def close(a, ref, tol=TOL, what=''):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    err = np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12) if ref.size else 0.0
    assert err <= tol, '%s: rel err %.3e > %.1e' % (what, err, tol)


def check_big(g, key, t, tol=TOL):
    if key in g.files:
        close(t.detach().numpy(), g[key], tol, key)
    else:
        close(t.detach().reshape(-1)[::97].numpy(), g[key + '@sub97'], tol, key)
        close(float(t.detach().double().norm()), g[key + '@norm'], tol, key + '@norm')


def check_module(g, mod, inputs, seed, tag, call=None, train=True, tol=TOL):
    call = call or (lambda m, *a: m(*a))
    chk = fill_params(mod, seed)
    assert abs(chk - float(g[tag + '/chk'])) <= 1e-9 * max(1.0, chk), 'weight generator drifted'
    mod.eval()
    with torch.no_grad():
        close(call(mod, *[i.clone() for i in inputs]).numpy(), g[tag + '/eval'], tol, tag + '/eval')
    if not train:
        return
    fill_params(mod, seed)
    mod.train()
    xs = [i.clone().requires_grad_(True) for i in inputs]
    y = call(mod, *xs)
    close(y.detach().numpy(), g[tag + '/train'], tol, tag + '/train')
    (y * probe_weights(seed, tag, tuple(y.shape))).sum().backward()
    for k, x in enumerate(xs):
        close(x.grad.numpy(), g[tag + '/gin%d' % k], 5 * tol, tag + '/gin')
    for n, p in mod.named_parameters():
        if p.grad is not None:
            check_big(g, tag + '/g:' + n, p.grad, 5 * tol)
    for n, b in mod.named_buffers():
        if not n.endswith('num_batches_tracked'):
            close(b.numpy(), g[tag + '/buf:' + n], tol, tag + '/buf:' + n)


@pytest.mark.parametrize('C', [8, 20])
@pytest.mark.parametrize('prim', oracle.PRIMITIVES)
def test_ops_registry(golden, prim, C):
    g = golden('ops')
    x = torch.from_numpy(g['x_C%d' % C])
    check_module(g, oracle.OPS[prim](C, 1, BN, **KW), [x], 100 + C, '%s_C%d' % (prim, C))


@pytest.mark.parametrize('prim', [p for p in oracle.PRIMITIVES if p != 'skip_connect'])
def test_ops_stride2(golden, prim):
    g = golden('ops')
    check_module(g, oracle.OPS[prim](8, 2, BN, **KW), [torch.from_numpy(g['x_s2'])], 150, '%s_s2' % prim)


def test_relu_conv_bn_and_reduce(golden):
    g = golden('ops')
    for ci, co in ((40, 24), (200, 40)):
        x = torch.from_numpy(g['rcb_x_%d' % ci])
        check_module(g, oracle.ReLUConvBN(ci, co, 1, 1, 0, BN, **KW), [x], 200 + ci, 'rcb_%d_%d' % (ci, co))
    for h in (16, 17, 15):
        x = torch.from_numpy(g['fr_x_%d' % h])
        check_module(g, oracle.FactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 300, 'fr_%d' % h)
        check_module(g, oracle.DoubleFactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 301, 'dfr_%d' % h)


def test_heads(golden):
    g = golden('heads')
    check_module(g, oracle.ASPP_train(40, 256, BN, mult=1), [torch.from_numpy(g['aspp40/x'])], 400, 'aspp40')
    check_module(g, oracle.ASPP_train(80, 256, BN, mult=2), [torch.from_numpy(g['aspp80_m2/x'])], 401, 'aspp80_m2')
    check_module(g, oracle.ASPP_train(400, 256, BN, mult=1), [torch.from_numpy(g['aspp400/x'])], 402, 'aspp400',
                 train=False)
    lo = torch.from_numpy(g['dec/low'])
    check_module(g, oracle.Decoder(19, BN), [torch.from_numpy(g['dec/x']), lo], 410, 'dec',
                 call=lambda m, a, b: m(a, b, (33, 65)))
    check_module(g, oracle.Decoder(19, BN), [torch.from_numpy(g['dec_same/x']), lo], 411, 'dec_same',
                 call=lambda m, a, b: m(a, b, (34, 66)))


def test_cells(golden):
    g = golden('cells')
    ga = torch.from_numpy(GENOTYPE_AUTODEEPLAB)
    c = oracle.Cell(BN, 5, 16, 32, ga, 1, 8, -1, dense_in=False, dense_out=True)
    ins = [torch.from_numpy(g['plain/pp']), torch.from_numpy(g['plain/p'])]
    for k, name in ((1, 'concat'), (2, 'dense')):
        check_module(g, c, ins, 500, 'plain_' + name, call=lambda m, a, b, k=k: m(a, b)[k])
    c = oracle.Cell(BN, 5, [8, 16, 8], 80, ga, 1, 8, 1, dense_in=True, dense_out=True)
    ins = [torch.from_numpy(g['densein/d%d' % i]) for i in range(3)] + [torch.from_numpy(g['densein/p'])]
    for k, name in ((1, 'concat'), (2, 'dense')):
        check_module(g, c, ins, 501, 'densein_' + name, call=lambda m, a, b, c_, p_, k=k: m([a, b, c_], p_)[k])
    c = oracle.Cell(BN, 5, [8, 8], 40, torch.from_numpy(GENOTYPE_40_1), 1, 8, 0, dense_in=True, dense_out=False)
    ins = [torch.from_numpy(g['last/d0']), torch.from_numpy(g['last/d1']), torch.from_numpy(g['last/p'])]
    check_module(g, c, ins, 502, 'last', call=lambda m, a, b, p_: m([a, b], p_))


@pytest.mark.parametrize('tag,Fv,arch', [('F4_65', 4, ARCH_C2), ('F4_64', 4, ARCH_C2), ('F4_C3_65', 4, ARCH_C3),
                                          ('F20_65', 20, ARCH_C2)])
def test_add_whole_net(golden, tag, Fv, arch):
    g = golden('add')
    m = oracle.ADD(arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(Fv), arch['low_level_layer'])
    chk = fill_params(m, 600)
    assert abs(chk - float(g[tag + '/chk'])) <= 1e-9 * chk
    x = torch.from_numpy(g[tag + '/x'])
    m.eval()
    with torch.no_grad():
        ys = m(x)
    assert len(ys) == len(arch['C_index']) + 1
    for i, y in enumerate(ys):
        check_big(g, tag + '/eval%d' % i, y, 1e-4)
    if tag + '/loss' not in g.files:
        return
    fill_params(m, 600)
    m.train()
    tgt = torch.from_numpy(g[tag + '/target'].astype(np.int64))
    ys = m(x)
    loss = oracle.cross_entropy_mean_exits(ys, tgt)
    loss.backward()
    assert abs(loss.item() - float(g[tag + '/loss'])) < 1e-5 * abs(float(g[tag + '/loss']))
    for i, y in enumerate(ys):
        check_big(g, tag + '/train%d' % i, y, 1e-4)
    pd = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(tag + '/g:'):
            name = k[len(tag) + 3:].split('@')[0]
            check_big(g, tag + '/g:' + name, pd[name].grad, 1e-3)
    gn = sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
    assert abs(gn - float(g[tag + '/gnorm'])) < 1e-4 * gn
    bd = dict(m.named_buffers())
    for k in g.files:
        if k.startswith(tag + '/buf:'):
            close(bd[k[len(tag) + 5:]].numpy(), g[k], 1e-4, k)


def test_dynamic_inference(golden):
    g = golden('dynamic')
    m = oracle.ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), 0)
    assert abs(fill_params(m, 700) - float(g['chk'])) < 1e-3
    edm = oracle.EDM()
    assert abs(fill_params(edm, 701) - float(g['chk_edm'])) < 1e-6
    m.eval(); edm.eval()
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        y, feat = m.get_feature(x)
        close(y.numpy(), g['get_feature/logits'], 1e-4, 'get_feature logits')
        close(feat.numpy(), g['get_feature/feature'], 1e-4, 'feature')
        close(edm(feat.clone()).numpy(), g['edm_on_feature'], 1e-4, 'edm')
        for name, thr in (('early', 1e9), ('final', -1e9)):
            y, ee, secs, conf = m.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)
            assert ee == int(g[name + '/exit'])
            close(y.numpy(), g[name + '/logits'], 1e-4, name)
            close(conf.numpy(), g[name + '/conf'], 1e-4, name + ' conf')
        ys = m(x)
        assert abs(oracle.normalized_shannon_entropy(ys[0]) - float(g['entropy0'])) < 1e-5
        assert abs(oracle.normalized_shannon_entropy(ys[1]) - float(g['entropy1'])) < 1e-5
        assert abs(oracle.confidence_max(ys[0], 0.2) - float(g['confmax0'])) < 1e-6


def test_baseline_model_config1(golden):
    """BASELINE config 1: searched_baseline network_path + genotype_2, low_level_layer=1, exit=last."""
    g = golden('baseline')
    m = oracle.Baselin_Model(NETWORK_PATH_BASELINE, [5], GENOTYPE_BASELINE_2, 19, make_args(20), 1)
    assert abs(fill_params(m, 800) - float(g['chk'])) < 1e-3
    m.eval()
    with torch.no_grad():
        ys = m(torch.from_numpy(g['129/x']))
        close(ys[-1].numpy(), g['129/last'], 1e-4, 'last')
        close(ys[0].numpy(), g['129/first'], 1e-4, 'first')
        ys = m(rand_tensor(81, 'base_x513', (1, 3, 513, 513)))
        close(ys[-1][:, :, ::8, ::8].numpy(), g['513/last_sub8'], 1e-4, '513 sub8')
        agree = (ys[-1].argmax(1)[:, ::4, ::4].numpy() == g['513/argmax_sub4']).mean()
        assert agree > 0.999


def test_bilinear_golden(golden):
    g = golden('bilinear')
    for k in ('down4', 'up_32_63', 'fit_63_64', 'up8', 'up_odd', 'down_odd'):
        x = torch.from_numpy(g[k + '/x'])
        y = F.interpolate(x, [int(v) for v in g[k + '/size']], mode='bilinear', align_corners=False)
        close(y.numpy(), g[k + '/y'], 1e-6, k)


def test_syncbn_definition_and_evaluator(golden):
    g = golden('misc')
    x = torch.from_numpy(g['syncbn/x'])
    rm, rv = torch.from_numpy(g['syncbn/rm']).clone(), torch.from_numpy(g['syncbn/rv']).clone()
    ys = oracle.global_batch_norm(list(x), rm, rv, torch.from_numpy(g['syncbn/w']), torch.from_numpy(g['syncbn/b']))
    close(torch.cat(ys).numpy(), g['syncbn/y'], 1e-6, 'global bn')
    ev = oracle.Evaluator(19)
    ev.add_batch(torch.from_numpy(g['eval/gt'].astype(np.int64)), torch.from_numpy(g['eval/pred'].astype(np.int64)))
    close(ev.confusion_matrix.numpy(), g['eval/cm'], 0, 'cm')
    assert abs(ev.Mean_Intersection_over_Union() - float(g['eval/miou'])) < 1e-7
    assert abs(float(ev.Pixel_Accuracy()) - float(g['eval/pa'])) < 1e-7
    assert abs(float(ev.Pixel_Accuracy_Class()) - float(g['eval/pac'])) < 1e-7
    assert abs(float(ev.Frequency_Weighted_Intersection_over_Union()) - float(g['eval/fwiou'])) < 1e-7


def _config_case(tag):
    import os
    sa = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'searched_arch', '40_5e_38_lr')
    from _util import ARCH_C4
    return {'F4_C4_65': (4, ARCH_C4, GENOTYPE_AUTODEEPLAB), 'F40_g1_65': (40, ARCH_C2, np.load(os.path.join(sa, 'genotype_1.npy'))),
            'F40_g2_65': (40, ARCH_C2, np.load(os.path.join(sa, 'genotype_2.npy')))}[tag]


@pytest.mark.parametrize('tag', ['F4_C4_65', 'F40_g1_65', 'F40_g2_65'])
def test_add_configs(golden, tag):
    """C=4 network (train.py:84-87) and config 5's F=40 architecture: oracle vs the reference's eval logits and one
    train-mode step (tests/golden/make_golden.py gen_configs)."""
    g = golden('configs')
    Fv, arch, geno = _config_case(tag)
    m = oracle.ADD(arch['network_arch'], arch['C_index'], geno, 19, make_args(Fv), arch['low_level_layer'])
    chk = fill_params(m, 600)
    assert abs(chk - float(g[tag + '/chk'])) <= 1e-9 * chk
    x = rand_tensor(61, 'add_x_' + tag, (2, 3, 65, 129))
    m.eval()
    with torch.no_grad():
        ys = m(x)
    assert len(ys) == len(arch['C_index']) + 1
    for i, y in enumerate(ys):
        check_big(g, tag + '/eval%d' % i, y, 1e-4)
    fill_params(m, 600)
    m.train()
    tgt = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (2, 65, 129))).long()
    tgt[torch.from_numpy(np.random.default_rng(63).random((2, 65, 129)) < 0.05)] = 255
    ys = m(x)
    loss = oracle.cross_entropy_mean_exits(ys, tgt)
    loss.backward()
    assert abs(loss.item() - float(g[tag + '/loss'])) < 1e-5 * abs(float(g[tag + '/loss']))
    for i, y in enumerate(ys):
        check_big(g, tag + '/train%d' % i, y, 1e-4)
    gn = sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
    assert abs(gn - float(g[tag + '/gnorm'])) < 1e-3 * float(g[tag + '/gnorm'])
