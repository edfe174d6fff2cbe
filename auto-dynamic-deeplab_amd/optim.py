"""Learning-rate schedule of the reference (utils/lr_scheduler.py:30-67: 'poly', 'cos', 'step' + warm-up), set
every iteration (train.py:224).  Works with torch optimizers and with addk.train.TrainStep.set_lr."""
import math


class LR_Scheduler(object):
    def __init__(self, mode, base_lr, num_epochs, iters_per_epoch=0, lr_step=0, warmup_epochs=0, min_lr=None):
        self.mode = mode
        self.lr = base_lr
        if mode == 'step':
            assert lr_step
        self.lr_step = lr_step
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.epoch = -1
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.min_lr = min_lr

    def value(self, i, epoch):
        T = epoch * self.iters_per_epoch + i
        if self.mode == 'cos':
            lr = self.lr * ((1 - self.min_lr) * 0.5 * (1 + math.cos(1.0 * T / self.N * math.pi)) + self.min_lr)
        elif self.mode == 'poly':
            lr = self.lr * pow((1 - 1.0 * T / self.N), 0.9)
        elif self.mode == 'step':
            lr = self.lr * (0.1 ** (epoch // self.lr_step))
        else:
            raise NotImplementedError(self.mode)
        if self.min_lr is not None and lr < self.min_lr:
            lr = self.min_lr
        if self.warmup_iters > 0 and T < self.warmup_iters:
            lr = lr * 1.0 * T / self.warmup_iters
        assert lr >= 0
        return lr

    def __call__(self, optimizer, i, epoch, best_pred=0.0):
        lr = self.value(i, epoch)
        self.epoch = max(self.epoch, epoch)
        if hasattr(optimizer, 'set_lr'):
            optimizer.set_lr(lr)
        elif len(optimizer.param_groups) == 1:
            optimizer.param_groups[0]['lr'] = lr
        else:
            optimizer.param_groups[0]['lr'] = lr
            for k in range(1, len(optimizer.param_groups)):
                optimizer.param_groups[k]['lr'] = lr * 10
        return lr
