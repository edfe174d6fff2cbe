"""Checkpoint / resume in the reference's own file format, and the class-balanced loss weights.

 * `Saver` — run directories and `checkpoint.pth.tar` / `model_best.pth.tar` / `best_pred.txt` exactly as
   utils/saver.py:8-45 lays them out (`run/<dataset>/<checkname>/experiment_<id>/`), so checkpoints written here are
   read by the reference's `--resume` and vice versa.  One fix: under DDP only rank 0 creates the run directory and
   writes files (the reference lets every rank glob and create `experiment_*`: SURVEY §5.2).
 * checkpoint dict `{'epoch', 'state_dict', 'optimizer', 'best_pred'}` (train.py:317-322).  `state_dict` keys are the
   module tree's (identical to the reference's); `optimizer` is a torch.optim.SGD state_dict (param_groups +
   per-parameter `momentum_buffer`), produced from / loaded into the fused step's flat momentum buffer
   (`optimizer_state_dict`, `load_optimizer_state_dict`), so `torch.optim.SGD.load_state_dict` accepts it unchanged.
 * `load_checkpoint` — train.py:184-210: tolerant key-wise copy (utils/copy_state_dict.py:1-17), optional 7-character
   `module.` prefix strip (`--clean-module`), optimizer state unless fine-tuning (`--ft`), returns (start_epoch, best_pred).
 * `calculate_weights_labels` — utils/calculate_weights.py:6-29: w_c = 1 / ln(1.02 + freq_c / total) over the labels of a
   loader, labels outside [0, num_classes) ignored.
"""
import glob
import os
import shutil
from collections import OrderedDict

import numpy as np
import torch


def _is_rank0():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class Saver(object):
    """utils/saver.py:8-45.  `args` needs .dataset and .checkname (and .network for save_experiment_config)."""

    def __init__(self, args, root='run'):
        self.args = args
        self.directory = os.path.join(root, args.dataset, args.checkname)
        self.runs = sorted(glob.glob(os.path.join(self.directory, 'experiment_*')))
        run_id = max([int(x.split('_')[-1]) for x in self.runs]) + 1 if self.runs else 0
        self.experiment_dir = os.path.join(self.directory, 'experiment_{}'.format(str(run_id)))
        self.writer = _is_rank0()
        if self.writer:
            os.makedirs(self.experiment_dir, exist_ok=True)

    def save_checkpoint(self, state, is_best, filename='checkpoint.pth.tar'):
        """Saves checkpoint to disk; on a new best also best_pred.txt and, if it beats every earlier run, model_best.pth.tar."""
        if not self.writer:
            return None
        filename = os.path.join(self.experiment_dir, filename)
        torch.save(state, filename)
        if is_best:
            best_pred = state['best_pred']
            with open(os.path.join(self.experiment_dir, 'best_pred.txt'), 'w') as f:
                f.write(str(best_pred))
            previous = [0.0]
            for run in self.runs:
                path = os.path.join(run, 'best_pred.txt')
                if os.path.exists(path):
                    with open(path, 'r') as f:
                        previous.append(float(f.readline()))
            if not self.runs or best_pred > max(previous):
                shutil.copyfile(filename, os.path.join(self.directory, 'model_best.pth.tar'))
        return filename

    def save_experiment_config(self):
        if not self.writer:
            return
        p = OrderedDict()
        p['network'] = getattr(self.args, 'network', None)
        p['datset'] = self.args.dataset           # the reference's key spelling (utils/saver.py:52)
        with open(os.path.join(self.experiment_dir, 'parameters.txt'), 'w') as f:
            for key, val in p.items():
                f.write(key + ':' + str(val) + '\n')


def copy_state_dict(cur_state_dict, pre_state_dict, prefix=''):
    """utils/copy_state_dict.py:1-17: copy every key found (under `prefix`), report the rest, never raise.  Returns the
    (missing, failed) key lists the reference only prints."""
    missing, failed = [], []
    for k in cur_state_dict.keys():
        v = pre_state_dict.get(prefix + k)
        if v is None:
            missing.append(k)
            continue
        try:
            cur_state_dict[k].copy_(v)
        except Exception:
            failed.append(k)
    return missing, failed


def optimizer_state_dict(step):
    """torch.optim.SGD-format state of a fused `TrainStep` (flat parameter / momentum buffers): parameter i of
    `model.parameters()` order gets state[i]['momentum_buffer'] shaped like the parameter."""
    mom, wd, nest = step.hyper
    state = {}
    off = 0
    started = step.steps > 0
    for i, p in enumerate(step.params):
        n = p.numel()
        if started:
            buf = torch.as_strided(step.mom_buf, p.shape, p.stride(), off).detach().clone().contiguous()
            state[i] = {'momentum_buffer': buf}
        off += (n + 3) // 4 * 4
    group = {'lr': float(step.lr_dev.item()), 'momentum': mom, 'dampening': 0, 'weight_decay': wd, 'nesterov': bool(nest),
             'maximize': False, 'foreach': None, 'differentiable': False, 'fused': None, 'params': list(range(len(step.params)))}
    return {'state': state, 'param_groups': [group]}


def load_optimizer_state_dict(step, sd):
    """Inverse of `optimizer_state_dict`; also accepts the state_dict of a real torch.optim.SGD over the same parameters."""
    groups = sd['param_groups']
    order = [i for g in groups for i in g['params']]
    assert len(order) == len(step.params), 'optimizer state has %d parameters, the step %d' % (len(order), len(step.params))
    g0 = groups[0]
    step.hyper = (g0.get('momentum', step.hyper[0]), g0.get('weight_decay', step.hyper[1]), int(bool(g0.get('nesterov', step.hyper[2]))))
    step.set_lr(g0['lr'])
    off = 0
    step.mom_buf.zero_()
    loaded = 0
    for i, p in zip(order, step.params):
        st = sd['state'].get(i, sd['state'].get(str(i)))
        if st is not None and st.get('momentum_buffer') is not None:
            torch.as_strided(step.mom_buf, p.shape, p.stride(), off).copy_(st['momentum_buffer'])
            loaded += 1
        off += (p.numel() + 3) // 4 * 4
    if loaded:
        step.steps = max(step.steps, 1)         # momentum buffers exist: the next step is not a "first step"
    return loaded


def save_state(model, step=None, optimizer=None, epoch=0, best_pred=0.0):
    """The checkpoint dict of train.py:317-322 (epoch is stored +1 there: pass what you want to resume from)."""
    opt = optimizer_state_dict(step) if step is not None else (optimizer.state_dict() if optimizer is not None else None)
    return {'epoch': epoch, 'state_dict': model.state_dict(), 'optimizer': opt, 'best_pred': best_pred}


def load_checkpoint(model, path, step=None, optimizer=None, clean_module=False, ft=False, map_location='cpu'):
    """train.py:184-210.  Returns (start_epoch, best_pred, missing_keys)."""
    if not os.path.isfile(path):
        raise RuntimeError("=> no checkpoint found at '{}'".format(path))
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = ck['state_dict']
    if clean_module:
        sd = OrderedDict((k[7:], v) for k, v in sd.items())      # remove 'module.' of DataParallel / DDP
    missing, failed = copy_state_dict(model.state_dict(), sd)
    if failed:
        raise RuntimeError('checkpoint tensors with wrong shapes: %s' % ', '.join(failed[:8]))
    if not ft and ck.get('optimizer') is not None:
        if step is not None:
            load_optimizer_state_dict(step, ck['optimizer'])
        elif optimizer is not None:
            optimizer.load_state_dict(ck['optimizer'])
    return (0 if ft else ck.get('epoch', 0)), ck.get('best_pred', 0.0), missing


def calculate_weights_labels(labels_iter, num_classes, save_path=None):
    """utils/calculate_weights.py:6-29.  `labels_iter` yields label tensors/arrays (or the reference's sample dicts with a
    'label' entry).  The histogram runs on whatever device the labels live on (one bincount per batch)."""
    z = torch.zeros(num_classes, dtype=torch.float64)
    for y in labels_iter:
        if isinstance(y, dict):
            y = y['label']
        y = torch.as_tensor(y).reshape(-1).long()
        y = y[(y >= 0) & (y < num_classes)]
        z += torch.bincount(y, minlength=num_classes).double().cpu()
    freq = (z / z.sum()).numpy()
    ret = 1.0 / np.log(1.02 + freq)
    if save_path is not None:
        np.save(save_path, ret)
    return ret
