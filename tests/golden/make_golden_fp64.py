"""Generate tests/golden/grads64.npz: whole-network gradients of the REAL reference evaluated in DOUBLE precision (and, beside them,
in its own fp32) for every draw the GPU suite's gradient gates use.  The reference is imported from /root/reference on PyTorch-CPU
in the build container; nothing of it travels — only sub-sampled numbers do.

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden_fp64.py [case ...]

Why: rounds 1-3 evaluated the CPU *oracle* in fp64 inside the GPU tests (2-3 minutes of host time per case on the GPU box, the bulk of a
500-670 s suite), and the fourth even-size draw had to go behind an env flag for suite time.  With these
fixtures the gates compare addk — and the fp32 oracle, which still runs in the test — against fp64 values held by the reference itself.

Per case (the parameter tensors that receive a gradient, in named_parameters() order; frozen-BatchNorm cases keep the conv weights only):
    <case>/names    their names, newline-joined          <case>/counts   samples held per tensor
    <case>/g64      the fp64 gradient at the sample positions `sample_index(numel)` of every tensor, concatenated (stored as float32:
                    6e-8 relative, three orders below the smallest error any gate looks at)
    <case>/d32      (the reference's OWN fp32 gradient - the fp64 one) at the same positions (pins the oracle's backward to the reference)
    <case>/stat     per tensor [max |g64|, sum g64^2, sum (g32 - g64)^2, max |g32 - g64|] over the WHOLE tensor
plus <case>/loss64, <case>/loss32 and <case>/chk (checksum of the name-keyed weights: the test rebuilds identical ones).
Inputs, targets and weights come from the deterministic generators of tests/_util.py with the seeds the tests use."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, '/root/reference')

from _util import ARCH_C2, ARCH_C4, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor      # noqa: E402
from grads64_util import SAMPLES, sample_index, target                                      # noqa: E402

from modeling.ADD import ADD                                                                # noqa: E402  (the reference)

torch.set_num_threads(8)
SA = os.path.join(HERE, '..', '..', 'searched_arch', '40_5e_38_lr')


def run_case(out, case, Fv, geno, hw, wseed, xseed, xname, tseed, train, only=None, ignore=True, conv_only=False, arch=ARCH_C2, logits=False, lstride=97):
    """One forward + CE (mean over exits) + backward of the reference in fp32 and in fp64 on the same weights / batch."""
    t0 = time.time()
    args = (arch['network_arch'], arch['C_index'], geno, 19, make_args(Fv), arch['low_level_layer'])
    x = rand_tensor(xseed, xname, (2, 3) + hw)
    tgt = target(hw, seed=tseed, ignore=ignore)
    crit = nn.CrossEntropyLoss(ignore_index=255)
    grads, losses = {}, {}
    for prec in ('32', '64'):
        m = ADD(*args)
        out[case + '/chk'] = np.float64(fill_params(m, wseed))
        if prec == '64':
            m = m.double()
        m.train(train)
        if only is not None:
            for n, p in m.named_parameters():
                p.requires_grad_(n in only)
        ys = m(x.double() if prec == '64' else x)
        loss = sum(crit(y, tgt) for y in ys) / len(ys)
        loss.backward()
        losses[prec] = float(loss)
        if logits:            # per-exit train-mode logits: every 97th element (as tests/golden/make_golden.py store_big) + the L2 norm
            for i, y in enumerate(ys):
                out['%s/logits%s_%d@sub%d' % (case, prec, i, lstride)] = y.detach().double().reshape(-1)[::lstride].float().numpy().copy()
                out['%s/logits%s_%d@maxabs' % (case, prec, i)] = np.float64(float(y.detach().abs().max()))
        grads[prec] = {n: p.grad.detach().double() for n, p in m.named_parameters() if p.grad is not None}
        del m, ys, loss
    out[case + '/loss64'], out[case + '/loss32'] = np.float64(losses['64']), np.float64(losses['32'])
    names, counts, a64, a32, stat = [], [], [], [], []
    for n, g64 in grads['64'].items():
        if conv_only and g64.dim() != 4:
            continue
        g32 = grads['32'][n]
        idx = torch.from_numpy(sample_index(g64.numel()))
        f64, f32 = g64.reshape(-1), g32.reshape(-1)
        d = f32 - f64
        names.append(n); counts.append(idx.numel())
        a64.append(f64[idx]); a32.append(d[idx])
        stat.append([float(f64.abs().max()), float((f64 ** 2).sum()), float((d ** 2).sum()), float(d.abs().max())])
    out[case + '/names'] = np.array('\n'.join(names))
    out[case + '/counts'] = np.array(counts, dtype=np.int32)
    out[case + '/g64'] = torch.cat(a64).float().numpy()
    out[case + '/d32'] = torch.cat(a32).float().numpy()
    out[case + '/stat'] = np.array(stat, dtype=np.float64)
    n_el = int(sum(counts))
    print('%-28s %4d tensors  %8d samples  loss64 %.7f loss32 %.7f  %.0f s' % (case, len(names), n_el, losses['64'], losses['32'], time.time() - t0), flush=True)


SENTINELS = ['stem1.0.weight', 'stem2.1.weight', 'cells.0._ops.1.op.2.weight', 'cells.0._ops.0.op.1.weight',
             'cells.3._ops.6.op.1.weight', 'cells.4.preprocess.conv_1.weight', 'cells.7.pre_preprocess_1x1.op.1.weight',
             'cells.11._ops.9.op.6.weight', 'low_level_conv.1.weight', 'aspp.aspp3.weight', 'aspp.conv1.weight',
             'decoder._conv.1.weight', 'decoder._conv.4.weight', 'decoder._conv.7.weight']


def cases():
    c = {}
    # tests/test_gpu_configs.py::test_train_mode_gradient_spread_is_the_references_own: F=4, train mode, 2 odd + 4 even draws
    for hw, nd in (((65, 129), 2), ((64, 128), 4)):
        for k in range(nd):
            c['spread_%dx%d_%d' % (hw + (k,))] = dict(Fv=4, geno=GENOTYPE_AUTODEEPLAB, hw=hw, wseed=600 + k, xseed=170 + k, xname='spread_x',
                                                      tseed=180 + 2 * k, train=True)
    # tests/test_gpu_configs.py::test_add_configs_eval_and_train_step: the C=4 network and config 5's architecture at 65x129, one train-mode step
    for tag, Fv, arch, g in (('F4_C4_65', 4, ARCH_C4, GENOTYPE_AUTODEEPLAB), ('F40_g1_65', 40, ARCH_C2, np.load(os.path.join(SA, 'genotype_1.npy'))),
                             ('F40_g2_65', 40, ARCH_C2, np.load(os.path.join(SA, 'genotype_2.npy')))):
        c['cfg_' + tag] = dict(Fv=Fv, geno=g, hw=(65, 129), wseed=600, xseed=61, xname='add_x_' + tag, tseed=62, train=True, arch=arch, logits=True)
    # tests/test_gpu_parity.py::test_add_whole_net_frozen_bn_gradients[F20_512x1024]: frozen BatchNorm, two draws
    for d, (sx, st) in enumerate([(61, 62), (71, 72)]):
        c['frozen512_%d' % d] = dict(Fv=20, geno=GENOTYPE_AUTODEEPLAB, hw=(512, 1024), wseed=600, xseed=sx, xname='frozen_x', tseed=st, train=False, ignore=False, conv_only=True)
    # tests/test_gpu_configs.py::test_f40_frozen_bn_gradients and tests/test_gpu_round3.py::test_bf16x3_whole_network_parity (config 5's architecture)
    for g in ('genotype_1', 'genotype_2'):
        c['f40_%s' % g] = dict(Fv=40, geno=np.load(os.path.join(SA, g + '.npy')), hw=(256, 512), wseed=900, xseed=61, xname='f40_frozen_x', tseed=62, train=False, conv_only=True)
    # tests/test_gpu_round3.py::test_full_size_frozen_bn_gradients_on_sentinel_convs: config 2 at 2x1024x2048, 14 sentinel convs
    c['full_sentinels'] = dict(Fv=20, geno=GENOTYPE_AUTODEEPLAB, hw=(1024, 2048), wseed=1003, xseed=203, xname='full_frozen_x', tseed=66, train=False, only=SENTINELS)
    # tests/test_gpu_round5.py::test_full_size_train_mode_gradients_on_sentinel_convs: the same shape and sentinels in TRAIN mode (batch statistics
    # in every BatchNorm, `train.py:227-240`): the headline step's backward held against the reference in double
    c['full_train_sentinels'] = dict(Fv=20, geno=GENOTYPE_AUTODEEPLAB, hw=(1024, 2048), wseed=1003, xseed=203, xname='full_frozen_x', tseed=66, train=True, only=SENTINELS,
                                     logits=True, lstride=9973)
    return c


def main():
    path = os.path.join(HERE, 'grads64.npz')
    out = dict(np.load(path)) if os.path.exists(path) else {}
    all_cases = cases()
    which = sys.argv[1:] or list(all_cases)
    for name in which:
        kw = dict(all_cases[name])
        for k in [k for k in out if k.startswith(name + '/')]:
            del out[k]
        run_case(out, name, **kw)
        np.savez_compressed(path, **out)
    print('%s: %.1f KB, %d arrays, sample cap %d' % (path, os.path.getsize(path) / 1024, len(out), SAMPLES))


if __name__ == '__main__':
    main()
