"""Checkpoint / resume and class-balanced weights (reference utils/saver.py:8-45, train.py:184-210,317-322,
utils/copy_state_dict.py, utils/calculate_weights.py:6-29).  CPU part: run-directory layout, best-run bookkeeping, tolerant
key copy, class weights vs the oracle's restatement.  GPU part: a TrainStep resumed from a checkpoint continues bit-exactly
and its optimizer state is a valid torch.optim.SGD state_dict."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from addk import checkpoint as ck


def _args(tmp):
    return SimpleNamespace(dataset='cityscapes', checkname='add', network='searched-dense')


def test_saver_layout_and_best_run_bookkeeping(tmp_path):
    root = str(tmp_path / 'run')
    s0 = ck.Saver(_args(tmp_path), root=root)
    assert s0.experiment_dir.endswith(os.path.join('cityscapes', 'add', 'experiment_0')) and os.path.isdir(s0.experiment_dir)
    s0.save_experiment_config()
    assert open(os.path.join(s0.experiment_dir, 'parameters.txt')).read() == 'network:searched-dense\ndatset:cityscapes\n'
    st = {'epoch': 3, 'state_dict': {'w': torch.ones(2)}, 'optimizer': None, 'best_pred': 0.5}
    f = s0.save_checkpoint(st, is_best=True)
    assert os.path.basename(f) == 'checkpoint.pth.tar' and float(open(os.path.join(s0.experiment_dir, 'best_pred.txt')).read()) == 0.5
    best = os.path.join(s0.directory, 'model_best.pth.tar')
    assert torch.load(best, weights_only=False)['best_pred'] == 0.5
    s1 = ck.Saver(_args(tmp_path), root=root)                     # second run: experiment_1, sees run 0
    assert s1.experiment_dir.endswith('experiment_1')
    s1.save_checkpoint(dict(st, best_pred=0.4), is_best=True)    # best of THIS run but worse than run 0: model_best stays
    assert torch.load(best, weights_only=False)['best_pred'] == 0.5
    s1.save_checkpoint(dict(st, best_pred=0.7), is_best=True)
    assert torch.load(best, weights_only=False)['best_pred'] == 0.7
    s1.save_checkpoint(dict(st, best_pred=0.1), is_best=False)
    assert torch.load(os.path.join(s1.experiment_dir, 'checkpoint.pth.tar'), weights_only=False)['best_pred'] == 0.1


def test_tolerant_copy_and_module_prefix(tmp_path):
    m = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    src = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    sd = {'module.' + k: v for k, v in src.state_dict().items() if not k.startswith('1.running_var')}
    path = str(tmp_path / 'c.pth.tar')
    torch.save({'epoch': 7, 'state_dict': sd, 'optimizer': None, 'best_pred': 0.25}, path)
    epoch, best, missing = ck.load_checkpoint(m, path, clean_module=True)
    assert (epoch, best, missing) == (7, 0.25, ['1.running_var'])
    assert torch.equal(m[0].weight, src[0].weight) and torch.equal(m[1].running_mean, src[1].running_mean)
    assert ck.load_checkpoint(m, path, clean_module=True, ft=True)[0] == 0          # --ft clears the start epoch
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(m, str(tmp_path / 'nope'))


def test_class_weights_match_oracle():
    g = np.random.default_rng(5)
    batches = [g.integers(0, 19, (2, 33, 65)) for _ in range(3)]
    for b in batches:
        b[g.random(b.shape) < 0.06] = 255
    w = ck.calculate_weights_labels([{'label': torch.from_numpy(b)} for b in batches], 19)
    ref = oracle.class_weights_from_labels(batches, 19)
    assert w.shape == (19,) and np.allclose(w, ref, rtol=1e-12)
    assert np.all(w > 0) and w.max() <= 1 / np.log(1.02) + 1e-9


@pytest.mark.gpu
def test_train_step_resumes_bit_exactly_and_speaks_torch_sgd(tmp_path):
    import addk  # noqa: F401
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args
    dev = torch.device('cuda:0')
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4), 0)

    def fresh():
        m = ADD(*args)
        fill_params(m, 33)
        return m.to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 65, 129, generator=g).to(dev)
    t = torch.randint(0, 19, (2, 65, 129), generator=g).to(dev)
    ma = fresh()
    sa = TrainStep(ma, (2, 3, 65, 129), lr=0.05, use_graph=False)
    for _ in range(2):
        sa.step(x, t)
    path = ck.Saver(SimpleNamespace(dataset='d', checkname='c'), root=str(tmp_path)).save_checkpoint(
        ck.save_state(ma, step=sa, epoch=5, best_pred=0.3), is_best=True)
    la = float(sa.step(x, t).item())                       # step 3 of the uninterrupted run
    mb = fresh()
    sb = TrainStep(mb, (2, 3, 65, 129), lr=0.01, use_graph=False)
    epoch, best, missing = ck.load_checkpoint(mb, path, step=sb, map_location=dev)
    assert (epoch, best, missing) == (5, 0.3, [])
    lb = float(sb.step(x, t).item())
    assert la == lb, (la, lb)
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k
    # the optimizer entry is a torch.optim.SGD state_dict over model.parameters()
    opt = torch.optim.SGD(mb.parameters(), lr=0.5, momentum=0.1)
    opt.load_state_dict(torch.load(path, weights_only=False)['optimizer'])
    pg = opt.param_groups[0]
    assert (pg['lr'], pg['momentum'], pg['weight_decay'], pg['nesterov']) == (pytest.approx(0.05), 0.9, 4e-5, True)
    p0 = next(mb.parameters())
    assert opt.state[p0]['momentum_buffer'].shape == p0.shape
