"""Per-iteration learning-rate schedule of the training loop (reference utils/lr_scheduler.py:30-67, called from
train.py:224 before every step).  The path trains with 'poly'; 'cos' and 'step' are the reference's two other modes.

The schedule is a pure function of the global iteration; `LR_Scheduler` is the callable the training loop holds (same
constructor arguments and call signature as the reference's class, so a train.py written against it runs unchanged) and it
feeds either a torch optimizer's param groups or the fused step's device-resident learning rate (`TrainStep.set_lr`)."""
import math


def _decay_poly(progress, epoch, cfg):
    return (1.0 - progress) ** 0.9


def _decay_cos(progress, epoch, cfg):
    floor = cfg['min_lr']
    return floor + (1.0 - floor) * 0.5 * (1.0 + math.cos(math.pi * progress))


def _decay_step(progress, epoch, cfg):
    return 0.1 ** (epoch // cfg['lr_step'])


DECAY = {'poly': _decay_poly, 'cos': _decay_cos, 'step': _decay_step}


def learning_rate(mode, base_lr, iteration, total_iters, epoch=0, lr_step=0, warmup_iters=0, min_lr=None):
    """Learning rate at global iteration `iteration` of `total_iters`."""
    if mode not in DECAY:
        raise NotImplementedError(mode)
    lr = base_lr * DECAY[mode](float(iteration) / total_iters, epoch, {'min_lr': min_lr, 'lr_step': lr_step})
    if min_lr is not None:
        lr = max(lr, min_lr)
    if iteration < warmup_iters:                  # linear ramp over the warm-up epochs
        lr *= float(iteration) / warmup_iters
    if lr < 0:
        raise ValueError('negative learning rate %r at iteration %d' % (lr, iteration))
    return lr


def apply_lr(optimizer, lr):
    """torch optimizer: group 0 gets lr, any further groups 10x (the reference's head/backbone split); fused step: set_lr."""
    if hasattr(optimizer, 'set_lr'):
        optimizer.set_lr(lr)
        return
    for k, group in enumerate(optimizer.param_groups):
        group['lr'] = lr if k == 0 else 10.0 * lr


class LR_Scheduler(object):
    def __init__(self, mode, base_lr, num_epochs, iters_per_epoch=0, lr_step=0, warmup_epochs=0, min_lr=None):
        if mode == 'step' and not lr_step:
            raise AssertionError("mode 'step' needs lr_step")
        self.mode, self.lr, self.lr_step, self.min_lr = mode, base_lr, lr_step, min_lr
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.epoch = -1

    def value(self, i, epoch):
        return learning_rate(self.mode, self.lr, epoch * self.iters_per_epoch + i, self.N, epoch, self.lr_step,
                             self.warmup_iters, self.min_lr)

    def __call__(self, optimizer, i, epoch, best_pred=0.0):
        lr = self.value(i, epoch)
        self.epoch = max(self.epoch, epoch)
        apply_lr(optimizer, lr)
        return lr
