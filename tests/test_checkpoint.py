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


# ---- pinned to the reference's own utilities: fixtures written by tests/golden/make_golden_host.py, which imports
# /root/reference/utils/{saver,calculate_weights,copy_state_dict,lr_scheduler}.py in the build container ----
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def host():
    return np.load(os.path.join(GOLD, 'host.npz'), allow_pickle=False)


def test_class_weights_match_reference_fixture(host, tmp_path):
    for case in range(3):
        labels = host['cw%d_labels' % case]
        path = str(tmp_path / ('w%d.npy' % case))
        w = ck.calculate_weights_labels([{'label': torch.from_numpy(b)} for b in labels], 19, save_path=path)
        assert np.allclose(w, host['cw%d_weights' % case], rtol=1e-12, atol=0)
        assert np.allclose(np.load(path), host['cw%d_saved' % case], rtol=1e-12, atol=0)     # the .npy the reference leaves on disk
        assert np.allclose(oracle.class_weights_from_labels(list(labels), 19), host['cw%d_weights' % case], rtol=1e-12)


def test_lr_schedules_match_reference_fixture(host):
    from addk.optim import LR_Scheduler
    import ast
    n = 0
    while 'lr%d' % n in host.files:
        mode, kw, ngroups = (str(v) for v in host['lr%d_cfg' % n])
        kw = dict(ast.literal_eval(kw))
        sch = LR_Scheduler(mode, **kw)
        for row in host['lr%d' % n]:
            epoch, i, want = int(row[0]), int(row[1]), row[2:]
            opt = SimpleNamespace(param_groups=[{'lr': None} for _ in range(int(ngroups))])
            lr = sch(opt, i, epoch, 0.0)
            got = np.array([g['lr'] for g in opt.param_groups])
            assert np.allclose(got, want, rtol=1e-14, atol=0), (mode, kw, epoch, i, got, want)
            assert lr == got[0]
        n += 1
    assert n == 5
    fused = SimpleNamespace(lr=None)
    fused.set_lr = lambda v: setattr(fused, 'lr', v)
    LR_Scheduler('poly', 0.05, 4, 7)(fused, 3, 2)                 # the fused step takes the rate through set_lr
    assert fused.lr == pytest.approx(0.05 * (1 - 17 / 28) ** 0.9, rel=1e-14)


def test_copy_state_dict_matches_reference_fixture(host):
    dst = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4), nn.Conv2d(4, 5, 3))
    dst.load_state_dict({k: torch.from_numpy(host['copy_dst_before/' + k]) for k in dst.state_dict()})
    pre = {k[len('copy_src/'):]: torch.from_numpy(host[k]) for k in host.files if k.startswith('copy_src/')}
    missing, failed = ck.copy_state_dict(dst.state_dict(), pre, prefix='module.')
    after = dst.state_dict()
    keys = [str(k) for k in host['copy_keys']]
    assert list(after.keys()) == keys
    for k, changed in zip(keys, host['copy_changed']):
        assert np.array_equal(after[k].numpy(), host['copy_dst_after/' + k]), k
        assert bool(changed) == (k not in missing and k not in failed and not np.array_equal(host['copy_dst_before/' + k], host['copy_dst_after/' + k]))
    assert missing == ['1.running_var'] and sorted(failed) == ['2.bias', '2.weight']     # what the reference only prints


def test_saver_matches_reference_layout_and_reads_its_checkpoint(host, tmp_path):
    # the same two runs the reference Saver was driven through: same files, same text
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        args = _args(tmp_path)
        state = {'epoch': 6, 'state_dict': {'w': torch.ones(1)}, 'optimizer': None, 'best_pred': 0.5}
        s0 = ck.Saver(args)
        s0.save_experiment_config()
        s0.save_checkpoint(state, True)
        s1 = ck.Saver(args)
        s1.save_checkpoint(dict(state, best_pred=0.4), True)
        s1.save_checkpoint(dict(state, best_pred=0.7, epoch=9), True)
        listing = sorted(os.path.join(r, f) for r, _, fs in os.walk('run') for f in fs)
        assert listing == sorted(str(v) for v in host['saver_listing'])
        assert open(os.path.join(s0.experiment_dir, 'parameters.txt')).read() == str(host['saver_parameters_txt'])
        assert [open(os.path.join(s.experiment_dir, 'best_pred.txt')).read() for s in (s0, s1)] == [str(v) for v in host['saver_best_pred_txt']]
        assert torch.load(os.path.join(s0.directory, 'model_best.pth.tar'), weights_only=False)['best_pred'] == float(host['saver_model_best_pred'])
    finally:
        os.chdir(cwd)
    # a checkpoint FILE written by the reference's Saver loads through load_checkpoint: weights, epoch, best_pred, optimizer
    m = nn.Sequential(nn.Conv2d(3, 4, 3, bias=False), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 1))
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    epoch, best, missing = ck.load_checkpoint(m, os.path.join(GOLD, 'ref_checkpoint.pth.tar'), optimizer=opt)
    assert (epoch, best, missing) == (6, 0.5, [])
    m.eval()
    y = m(torch.from_numpy(host['saver_probe_x'])).detach().numpy()
    assert np.allclose(y, host['saver_probe_y'], rtol=1e-6, atol=1e-7)
    pg = opt.param_groups[0]
    assert (pg['lr'], pg['momentum'], pg['weight_decay'], pg['nesterov']) == (0.05, 0.9, 4e-5, True)
    assert all('momentum_buffer' in opt.state[p] for p in m.parameters())


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
