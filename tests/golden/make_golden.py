"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference on PyTorch-CPU in the build container).  Run once:

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden.py

Nothing here travels to the GPU box except the .npz outputs.  Inputs come from
the deterministic generators in tests/_util.py; weights are filled by
`fill_params` keyed on state_dict names (checksum stored in every fixture), so
the same call on the oracle / product reproduces them without RNG-order
dependence (SURVEY §8c caveat iv).
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, '/root/reference')

from _util import (ARCH_C2, ARCH_C3, ARCH_C4, GENOTYPE_AUTODEEPLAB, GENOTYPE_BASELINE_2, GENOTYPE_40_1,  # noqa: E402
                   NETWORK_PATH_BASELINE, fill_params, make_args, probe_weights, rand_tensor)

import modeling.operations as R_ops            # noqa: E402
from modeling.genotypes import PRIMITIVES       # noqa: E402
from modeling.ADD import ADD, EDM, Cell         # noqa: E402
from modeling.aspp_train import ASPP_train      # noqa: E402
from modeling.decoder import Decoder            # noqa: E402
from modeling.baseline_model import Baselin_Model  # noqa: E402
from utils.metrics import Evaluator             # noqa: E402

torch.set_num_threads(8)
BN = nn.BatchNorm2d
KW = dict(eps=1e-5, momentum=0.1, affine=True)


def npy(t):
    return t.detach().cpu().numpy().copy()      # copy: .grad and BN buffers are mutated in place later


def store_big(out, key, t):
    """Large tensors are pinned by an every-97th-element subsample + their L2 norm."""
    if t.numel() > 40000:
        out[key + '@sub97'] = npy(t.reshape(-1)[::97])
        out[key + '@norm'] = np.float64(float(t.double().norm()))
    else:
        out[key] = npy(t)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-28s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrs)))


def run_module(mod, inputs, seed, tag, out, call=None, train=True):
    """Record eval output, train output, input/param grads and post-step BN stats."""
    call = call or (lambda m, *a: m(*a))
    chk = fill_params(mod, seed)
    out[tag + '/chk'] = np.float64(chk)
    mod.eval()
    with torch.no_grad():
        y = call(mod, *[i.clone() for i in inputs])
    out[tag + '/eval'] = npy(y)
    if not train:
        return
    fill_params(mod, seed)
    mod.train()
    xs = [i.clone().requires_grad_(True) for i in inputs]
    y = call(mod, *xs)
    out[tag + '/train'] = npy(y)
    w = probe_weights(seed, tag, tuple(y.shape))
    (y * w).sum().backward()
    for k, x in enumerate(xs):
        out[tag + '/gin%d' % k] = npy(x.grad)
    for n, p in mod.named_parameters():
        if p.grad is None:
            continue
        store_big(out, tag + '/g:' + n, p.grad)
    for n, b in mod.named_buffers():
        if not n.endswith('num_batches_tracked'):
            out[tag + '/buf:' + n] = npy(b)


def gen_ops():
    out = {}
    for C, hw in ((8, (16, 32)), (20, (17, 33))):
        x = rand_tensor(11, 'ops_x_%d' % C, (2, C) + hw)
        out['x_C%d' % C] = npy(x)
        for prim in PRIMITIVES:
            m = R_ops.OPS[prim](C, 1, BN, **KW)
            run_module(m, [x], 100 + C, '%s_C%d' % (prim, C), out)
    # stride-2 registry entries (cold: the hot path always passes stride=1, ADD.py:61)
    x = rand_tensor(11, 'ops_x_s2', (2, 8, 16, 32))
    out['x_s2'] = npy(x)
    for prim in PRIMITIVES:
        if prim == 'skip_connect':
            continue
        m = R_ops.OPS[prim](8, 2, BN, **KW)
        run_module(m, [x], 150, '%s_s2' % prim, out)
    for ci, co, hw in ((40, 24, (9, 17)), (200, 40, (8, 16))):
        x = rand_tensor(12, 'rcb_x_%d' % ci, (2, ci) + hw)
        out['rcb_x_%d' % ci] = npy(x)
        run_module(R_ops.ReLUConvBN(ci, co, 1, 1, 0, BN, **KW), [x], 200 + ci, 'rcb_%d_%d' % (ci, co), out)
    for hw in ((16, 32), (17, 33), (15, 31)):
        x = rand_tensor(13, 'fr_x_%d' % hw[0], (2, 24) + hw)
        out['fr_x_%d' % hw[0]] = npy(x)
        run_module(R_ops.FactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 300, 'fr_%d' % hw[0], out)
        run_module(R_ops.DoubleFactorizedReduce(24, 16, BN, eps=1e-5, momentum=0.1), [x], 301, 'dfr_%d' % hw[0], out)
    save('ops', **out)


def gen_bilinear():
    out = {}
    cases = {'down4': ((2, 8, 32, 64), (8, 16)), 'up_32_63': ((2, 8, 32, 64), (63, 127)),
             'fit_63_64': ((1, 8, 63, 127), (64, 128)), 'up8': ((1, 19, 9, 17), (65, 129)),
             'up_odd': ((2, 4, 17, 33), (33, 65)), 'down_odd': ((1, 4, 65, 129), (17, 33))}
    for k, (shp, size) in cases.items():
        x = rand_tensor(21, 'bil_' + k, shp).requires_grad_(True)
        y = F.interpolate(x, list(size), mode='bilinear')
        w = probe_weights(21, 'bil_' + k, tuple(y.shape))
        (y * w).sum().backward()
        out[k + '/x'], out[k + '/y'], out[k + '/gx'] = npy(x), npy(y), npy(x.grad)
        out[k + '/size'] = np.array(size)
    x = rand_tensor(21, 'bil_ac', (2, 8, 1, 1))
    out['ac_true/x'] = npy(x)
    out['ac_true/y'] = npy(nn.Upsample((5, 7), mode='bilinear', align_corners=True)(x))
    save('bilinear', **out)


def gen_heads():
    out = {}
    x = rand_tensor(31, 'aspp_x40', (2, 40, 9, 17))
    out['aspp40/x'] = npy(x)
    run_module(ASPP_train(40, 256, BN, mult=1), [x], 400, 'aspp40', out)
    x = rand_tensor(31, 'aspp_x80', (2, 80, 20, 24))
    out['aspp80_m2/x'] = npy(x)
    run_module(ASPP_train(80, 256, BN, mult=2), [x], 401, 'aspp80_m2', out)
    x = rand_tensor(31, 'aspp_x400', (2, 400, 9, 17))
    out['aspp400/x'] = npy(x)
    run_module(ASPP_train(400, 256, BN, mult=1), [x], 402, 'aspp400', out, train=False)
    xa = rand_tensor(32, 'dec_x', (2, 256, 5, 9))
    lo = rand_tensor(32, 'dec_low', (2, 48, 9, 17))
    out['dec/x'], out['dec/low'] = npy(xa), npy(lo)
    run_module(Decoder(19, BN), [xa, lo], 410, 'dec', out, call=lambda m, a, b: m(a, b, (33, 65)))
    xb = rand_tensor(32, 'dec_x_same', (2, 256, 9, 17))
    out['dec_same/x'] = npy(xb)
    run_module(Decoder(19, BN), [xb, lo], 411, 'dec_same', out, call=lambda m, a, b: m(a, b, (34, 66)))
    save('heads', **out)


def gen_cells():
    out = {}
    g = torch.from_numpy(GENOTYPE_AUTODEEPLAB)
    # plain (cell-0 style: FactorizedReduce preprocess, resize of prev_prev)
    c = Cell(BN, 5, 16, 32, g, 1, 8, -1, dense_in=False, dense_out=True)
    pp = rand_tensor(41, 'c0_pp', (2, 16, 32, 64)); p = rand_tensor(41, 'c0_p', (2, 32, 16, 32))
    out['plain/pp'], out['plain/p'] = npy(pp), npy(p)
    for k, name in ((1, 'concat'), (2, 'dense')):
        run_module(c, [pp, p], 500, 'plain_' + name, out, call=lambda m, a, b, k=k: m(a, b)[k])
    # dense-in with upsample (downup_sample=+1) and mixed-resolution dense inputs
    c = Cell(BN, 5, [8, 16, 8], 80, g, 1, 8, 1, dense_in=True, dense_out=True)
    d = [rand_tensor(42, 'cd_%d' % i, s) for i, s in enumerate(((2, 8, 16, 32), (2, 16, 8, 16), (2, 8, 15, 31)))]
    p = rand_tensor(42, 'cd_p', (2, 80, 8, 16))
    for i, t in enumerate(d):
        out['densein/d%d' % i] = npy(t)
    out['densein/p'] = npy(p)
    for k, name in ((1, 'concat'), (2, 'dense')):
        run_module(c, d + [p], 501, 'densein_' + name, out,
                   call=lambda m, a, b, c_, p_, k=k: m([a, b, c_], p_)[k])
    # last (dense_out=False), unsorted 40_5e genotype (Q1: three swapped pairs)
    c = Cell(BN, 5, [8, 8], 40, torch.from_numpy(GENOTYPE_40_1), 1, 8, 0, dense_in=True, dense_out=False)
    d = [rand_tensor(43, 'cl_%d' % i, (2, 8, 9, 17)) for i in range(2)]
    p = rand_tensor(43, 'cl_p', (2, 40, 9, 17))
    out['last/d0'], out['last/d1'], out['last/p'] = npy(d[0]), npy(d[1]), npy(p)
    run_module(c, d + [p], 502, 'last', out, call=lambda m, a, b, p_: m([a, b], p_))
    save('cells', **out)


SENTINELS = ['stem0.0.weight', 'cells.0._ops.0.op.1.weight', 'cells.3.pre_preprocess.1.op.1.weight',
             'cells.5._ops.8.op.1.weight', 'cells.5._ops.9.op.5.weight', 'cells.11.preprocess.op.2.weight',
             'aspp.aspp3.weight', 'aspp.bn1.bias', 'decoder._conv.4.weight', 'decoder._conv.7.bias',
             'low_level_conv.1.weight', 'cells.4.preprocess.conv_2.weight']


def gen_add():
    out = {}
    for tag, Fv, arch, hw, n in (('F4_65', 4, ARCH_C2, (65, 129), 2), ('F4_64', 4, ARCH_C2, (64, 128), 2),
                                 ('F4_C3_65', 4, ARCH_C3, (65, 129), 2), ('F20_65', 20, ARCH_C2, (65, 129), 1)):
        m = ADD(arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(Fv), arch['low_level_layer'])
        out[tag + '/chk'] = np.float64(fill_params(m, 600))
        x = rand_tensor(61, 'add_x_' + tag, (n, 3) + hw)
        out[tag + '/x'] = npy(x)
        m.eval()
        with torch.no_grad():
            ys = m(x)
        for i, y in enumerate(ys):
            if tag in ('F4_65', 'F20_65'):
                out[tag + '/eval%d' % i] = npy(y)
            else:
                store_big(out, tag + '/eval%d' % i, y)
        if n < 2:
            continue
        fill_params(m, 600)
        m.train()
        tgt = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (n,) + hw)).long()
        tgt[torch.from_numpy(np.random.default_rng(63).random((n,) + hw) < 0.05)] = 255
        out[tag + '/target'] = npy(tgt).astype(np.uint8)
        ys = m(x)
        crit = nn.CrossEntropyLoss(weight=None, ignore_index=255)
        loss = sum(crit(y, tgt) for y in ys) / len(ys)
        loss.backward()
        out[tag + '/loss'] = np.float64(loss.item())
        for i, y in enumerate(ys):
            store_big(out, tag + '/train%d' % i, y)
        pd = dict(m.named_parameters())
        for s in SENTINELS:
            if s in pd and pd[s].grad is not None:
                store_big(out, tag + '/g:' + s, pd[s].grad)
        out[tag + '/gnorm'] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5)
        bd = dict(m.named_buffers())
        for s in ('stem1.1.running_mean', 'aspp.aspp5_bn.running_var', 'aspp.bn1.running_mean',
                  'decoder._conv.2.running_var', 'cells.7._ops.3.op.2.running_var'):
            out[tag + '/buf:' + s] = npy(bd[s])
    save('add', **out)


def gen_dynamic():
    out = {}
    torch.cuda.synchronize = lambda *a, **k: None     # ADD.py:380,436 call it unconditionally; no GPU here
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), 0)
    out['chk'] = np.float64(fill_params(m, 700))
    edm = EDM()
    out['chk_edm'] = np.float64(fill_params(edm, 701))
    m.eval(); edm.eval()
    x = rand_tensor(71, 'dyn_x', (1, 3, 65, 129))
    out['x'] = npy(x)
    with torch.no_grad():
        y, feat = m.get_feature(x)
        out['get_feature/logits'], out['get_feature/feature'] = npy(y), npy(feat)
        out['edm_on_feature'] = npy(edm(feat.clone()))
        for name, thr in (('early', 1e9), ('final', -1e9)):
            y, ee, _, conf = m.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)
            out[name + '/logits'], out[name + '/exit'], out[name + '/conf'] = npy(y), np.int64(ee), npy(conf)
        ys = m(x)
        out['entropy0'] = np.float64(R_ops.normalized_shannon_entropy(ys[0]))
        out['entropy1'] = np.float64(R_ops.normalized_shannon_entropy(ys[1]))
        out['confmax0'] = np.float64(R_ops.confidence_max(ys[0], 0.2))
        out['fwd0'] = npy(ys[0])
    save('dynamic', **out)


def gen_baseline():
    out = {}
    m = Baselin_Model(NETWORK_PATH_BASELINE, [5], GENOTYPE_BASELINE_2, 19, make_args(20), 1)   # eval_edm.py:56-60
    out['chk'] = np.float64(fill_params(m, 800))
    m.eval()
    for tag, hw in (('129', (129, 129)), ('513', (513, 513))):
        x = rand_tensor(81, 'base_x' + tag, (1, 3) + hw)
        with torch.no_grad():
            ys = m(x)
        if tag == '129':
            out['129/x'] = npy(x)
            out['129/last'] = npy(ys[-1])
            out['129/first'] = npy(ys[0])
        else:                                   # BASELINE config 1: input regenerated from its seed in the test
            out['513/last_sub8'] = npy(ys[-1][:, :, ::8, ::8])
            out['513/last_absmean'] = np.float64(ys[-1].abs().mean().item())
            out['513/argmax_sub4'] = npy(ys[-1].argmax(1)[:, ::4, ::4]).astype(np.uint8)
    save('baseline', **out)


def gen_misc():
    out = {}
    shards = [rand_tensor(91, 'sbn_%d' % r, (2, 12, 5, 7)) * (1 + 0.3 * r) + 0.2 * r for r in range(8)]
    rm, rv = torch.zeros(12), torch.ones(12)
    w, b = rand_tensor(91, 'sbn_w', (12,)) * 0.2 + 1, rand_tensor(91, 'sbn_b', (12,)) * 0.2
    y = F.batch_norm(torch.cat(shards), rm, rv, w, b, True, 0.1, 1e-5)
    out['syncbn/x'], out['syncbn/y'] = npy(torch.stack(shards)), npy(y)
    out['syncbn/w'], out['syncbn/b'], out['syncbn/rm'], out['syncbn/rv'] = npy(w), npy(b), npy(rm), npy(rv)
    ev = Evaluator(19)
    g = np.random.default_rng(92)
    gt = torch.from_numpy(g.integers(0, 19, (2, 33, 65))).long()
    gt[torch.from_numpy(g.random((2, 33, 65)) < 0.07)] = 255
    pr = torch.from_numpy(g.integers(0, 19, (2, 33, 65))).long()
    pr = torch.where(torch.from_numpy(g.random((2, 33, 65)) < 0.5), gt.clamp(max=18), pr)
    ev.add_batch(gt, pr)
    out['eval/gt'], out['eval/pred'] = npy(gt).astype(np.uint8), npy(pr).astype(np.uint8)
    out['eval/cm'] = npy(ev.confusion_matrix)
    out['eval/miou'] = np.float64(ev.Mean_Intersection_over_Union())
    out['eval/pa'] = np.float64(float(ev.Pixel_Accuracy()))
    out['eval/pac'] = np.float64(float(ev.Pixel_Accuracy_Class()))
    out['eval/fwiou'] = np.float64(float(ev.Frequency_Weighted_Intersection_over_Union()))
    save('misc', **out)


def gen_configs():
    """Round-2 additions: the C=4 network (train.py:84-87: three 1x1 conv_aspp adapters, level-3 cells feeding a level-2
    head) and BASELINE config 5's architecture (F=40, both searched_arch/40_5e_38_lr genotypes), eval logits and one
    train-mode step (loss, per-exit logits, sentinel gradients) from the real reference."""
    out = {}
    sa = os.path.join(HERE, '..', '..', 'searched_arch', '40_5e_38_lr')
    cases = (('F4_C4_65', 4, ARCH_C4, GENOTYPE_AUTODEEPLAB, (65, 129)),
             ('F40_g1_65', 40, ARCH_C2, np.load(os.path.join(sa, 'genotype_1.npy')), (65, 129)),
             ('F40_g2_65', 40, ARCH_C2, np.load(os.path.join(sa, 'genotype_2.npy')), (65, 129)))
    for tag, Fv, arch, geno, hw in cases:
        m = ADD(arch['network_arch'], arch['C_index'], geno, 19, make_args(Fv), arch['low_level_layer'])
        out[tag + '/chk'] = np.float64(fill_params(m, 600))
        x = rand_tensor(61, 'add_x_' + tag, (2, 3) + hw)
        m.eval()
        with torch.no_grad():
            ys = m(x)
        for i, y in enumerate(ys):
            store_big(out, tag + '/eval%d' % i, y)
        fill_params(m, 600)
        m.train()
        tgt = torch.from_numpy(np.random.default_rng(62).integers(0, 19, (2,) + hw)).long()
        tgt[torch.from_numpy(np.random.default_rng(63).random((2,) + hw) < 0.05)] = 255
        ys = m(x)
        crit = nn.CrossEntropyLoss(weight=None, ignore_index=255)
        loss = sum(crit(y, tgt) for y in ys) / len(ys)
        loss.backward()
        out[tag + '/loss'] = np.float64(loss.item())
        for i, y in enumerate(ys):
            store_big(out, tag + '/train%d' % i, y)
        out[tag + '/gnorm'] = np.float64(sum(float((p.grad.double() ** 2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5)
    save('configs', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'bilinear', 'heads', 'cells', 'add', 'dynamic', 'baseline', 'misc', 'configs']
    for w in which:
        globals()['gen_' + w]()
