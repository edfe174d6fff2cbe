#!/usr/bin/env python
"""Fixtures for the host utilities around the path, produced by the REFERENCE's own code (run in the build container only):

    python tests/golden/make_golden_host.py

imports /root/reference/utils/{saver,calculate_weights,copy_state_dict,lr_scheduler}.py and writes
  tests/golden/host.npz                 class weights of fixed label batches, learning rates of the three schedules, the keys
                                        copy_state_dict copies / skips, the file layout Saver leaves behind
  tests/golden/ref_checkpoint.pth.tar   a checkpoint file written by the reference's Saver.save_checkpoint (tiny model + SGD)
Only `mypath.Path.db_root_dir` is redirected (the reference hard-codes a placeholder dataset directory to save into)."""
import contextlib
import io
import os
import shutil
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')

import mypath                                             # noqa: E402
from utils.calculate_weights import calculate_weigths_labels   # noqa: E402
from utils.copy_state_dict import copy_state_dict        # noqa: E402
from utils.lr_scheduler import LR_Scheduler              # noqa: E402
from utils.saver import Saver                            # noqa: E402

out = {}
tmp = tempfile.mkdtemp()
mypath.Path.db_root_dir = staticmethod(lambda dataset: tmp)

# ---- class-balanced weights (utils/calculate_weights.py:6-29) ----
g = np.random.default_rng(5)
for case, (nb, shape, pign) in enumerate(((3, (2, 33, 65), 0.06), (2, (4, 17, 19), 0.5), (1, (1, 8, 8), 0.0))):
    batches = [g.integers(0, 19, shape) for _ in range(nb)]
    for b in batches:
        b[g.random(b.shape) < pign] = 255
    if case == 1:
        batches[0][batches[0] == 7] = 3                   # a class that never occurs in one batch
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        w = calculate_weigths_labels('cityscapes', [{'label': torch.from_numpy(b)} for b in batches], 19)
    out['cw%d_labels' % case] = np.stack(batches).astype(np.int64)
    out['cw%d_weights' % case] = np.asarray(w, dtype=np.float64)
    out['cw%d_saved' % case] = np.load(os.path.join(tmp, 'cityscapes_classes_weights.npy'))

# ---- learning-rate schedules (utils/lr_scheduler.py:30-67) ----
class _Opt:                                               # what the scheduler writes into
    def __init__(self, ngroups):
        self.param_groups = [{'lr': None} for _ in range(ngroups)]

cases = [('poly', dict(base_lr=0.05, num_epochs=4, iters_per_epoch=7), 1), ('poly', dict(base_lr=0.1, num_epochs=3, iters_per_epoch=5, warmup_epochs=1), 2),
         ('cos', dict(base_lr=0.025, num_epochs=5, iters_per_epoch=3, min_lr=0.001), 1), ('step', dict(base_lr=0.01, num_epochs=6, iters_per_epoch=2, lr_step=2), 3),
         ('poly', dict(base_lr=0.05, num_epochs=2, iters_per_epoch=4, min_lr=0.02), 1)]
for ci, (mode, kw, ngroups) in enumerate(cases):
    with contextlib.redirect_stdout(io.StringIO()):
        sch = LR_Scheduler(mode, **kw)
        rows = []
        for epoch in range(kw['num_epochs']):
            for i in range(kw['iters_per_epoch']):
                opt = _Opt(ngroups)
                sch(opt, i, epoch, 0.0)
                rows.append([epoch, i] + [grp['lr'] for grp in opt.param_groups])
    out['lr%d' % ci] = np.asarray(rows, dtype=np.float64)
    out['lr%d_cfg' % ci] = np.asarray([mode, repr(sorted(kw.items())), str(ngroups)])

# ---- tolerant copy (utils/copy_state_dict.py:1-17) ----
torch.manual_seed(0)
src = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 3))
dst = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4), nn.Conv2d(4, 5, 3))      # layer 2 has another shape: "copy param failed"
pre = {'module.' + k: v.clone() for k, v in src.state_dict().items() if k != '1.running_var'}
before = {k: v.clone() for k, v in dst.state_dict().items()}
with contextlib.redirect_stdout(io.StringIO()):
    copy_state_dict(dst.state_dict(), pre, prefix='module.')
after = dst.state_dict()
out['copy_keys'] = np.asarray(list(after.keys()))
out['copy_changed'] = np.asarray([not torch.equal(before[k], after[k]) for k in after])
for k, v in pre.items():
    out['copy_src/' + k] = v.numpy()
for k, v in before.items():
    out['copy_dst_before/' + k] = v.numpy()
for k, v in after.items():
    out['copy_dst_after/' + k] = v.numpy()

# ---- Saver (utils/saver.py:8-45): two runs, best-run bookkeeping, and one checkpoint file kept as a fixture ----
cwd = os.getcwd()
os.chdir(tmp)
try:
    args = SimpleNamespace(dataset='cityscapes', checkname='add', network='searched-dense')
    torch.manual_seed(1)
    m = nn.Sequential(nn.Conv2d(3, 4, 3, bias=False), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 1))
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)
    for _ in range(2):
        opt.zero_grad()
        m(torch.randn(2, 3, 8, 8)).square().mean().backward()
        opt.step()
    s0 = Saver(args)
    s0.save_experiment_config()
    state = {'epoch': 6, 'state_dict': m.state_dict(), 'optimizer': opt.state_dict(), 'best_pred': 0.5}
    s0.save_checkpoint(state, True)
    s1 = Saver(args)
    s1.save_checkpoint(dict(state, best_pred=0.4), True)
    s1.save_checkpoint(dict(state, best_pred=0.7, epoch=9), True)
    listing = []
    for root, _, files in sorted(os.walk('run')):
        for f in sorted(files):
            listing.append(os.path.join(root, f))
    out['saver_listing'] = np.asarray(listing)
    out['saver_parameters_txt'] = np.asarray(open(os.path.join(s0.experiment_dir, 'parameters.txt')).read())
    out['saver_best_pred_txt'] = np.asarray([open(os.path.join(s.experiment_dir, 'best_pred.txt')).read() for s in (s0, s1)])
    out['saver_model_best_pred'] = np.asarray(torch.load(os.path.join(s0.directory, 'model_best.pth.tar'), weights_only=False)['best_pred'])
    shutil.copyfile(os.path.join(s0.experiment_dir, 'checkpoint.pth.tar'), os.path.join(HERE, 'ref_checkpoint.pth.tar'))
    x = torch.randn(2, 3, 8, 8)
    m.eval()
    out['saver_probe_x'] = x.numpy()
    out['saver_probe_y'] = m(x).detach().numpy()
finally:
    os.chdir(cwd)
    shutil.rmtree(tmp, ignore_errors=True)

np.savez_compressed(os.path.join(HERE, 'host.npz'), **out)
print('wrote host.npz (%d arrays) and ref_checkpoint.pth.tar' % len(out))
