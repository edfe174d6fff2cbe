"""nn.CrossEntropyLoss(weight, ignore_index=255) as used at train.py:70,231, as one fused HIP pass over the NCHW
logits (softmax, NLL and the gradient in the same read).  Drop-in: `criterion(outputs[i], target)`."""
import torch
import torch.nn as nn

from . import _lib as L
from . import plan as _plan

import os
_CHECK_TARGETS = os.environ.get('ADDK_CHECK_TARGETS', '0') == '1'


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index):
        lib = L.load()
        _plan.require_device(logits)
        logits = logits.contiguous().float()
        N, Cc, H, W = logits.shape
        dev = logits.device
        # the kernel dereferences raw pointers: anything still on the host (a class-weight buffer never moved with
        # .cuda(), a CPU target) would be a GPU memory fault, so it is moved here like torch's own loss would refuse it
        target = target.to(dev, non_blocking=True).contiguous().long()
        if weight is not None:
            weight = weight.to(dev).float().contiguous()
            if weight.numel() != Cc:
                raise ValueError('CrossEntropyLoss: weight has %d entries for %d classes' % (weight.numel(), Cc))
        if tuple(target.shape) != (N, H, W):
            raise ValueError('CrossEntropyLoss: target shape %s does not match logits %s' % (tuple(target.shape), tuple(logits.shape)))
        if _CHECK_TARGETS:                   # ADDK_CHECK_TARGETS=1: torch raises on labels outside [0, C) that are not ignore_index
            bad = (target != ignore_index) & ((target < 0) | (target >= Cc))
            if bool(bad.any()):
                raise IndexError('CrossEntropyLoss: target %d is out of bounds' % int(target[bad][0]))
        loss = torch.zeros(1, device=dev)
        wsum = torch.zeros(1, device=dev)
        ws = torch.empty(int(lib.addk_ce_ws_floats(N, H * W)), device=dev)
        dl = torch.empty_like(logits) if logits.requires_grad or torch.is_grad_enabled() else None
        st = _plan.current_stream()
        wp = weight.data_ptr() if weight is not None else None
        L.check(lib.addk_ce_count(target.data_ptr(), N * H * W, wp, ignore_index, Cc, wsum.data_ptr(), ws.data_ptr(), st), 'ce_count')
        L.check(lib.addk_ce_fwd_bwd(logits.data_ptr(), target.data_ptr(), N, Cc, H * W, wp, ignore_index, wsum.data_ptr(), 1.0,
                                    loss.data_ptr(), dl.data_ptr() if dl is not None else None, ws.data_ptr(), st), 'ce_fwd_bwd')
        ctx.dl = dl
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        return ctx.dl * gout, None, None, None


class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=255):
        super().__init__()
        self.register_buffer('weight', weight.float().contiguous() if weight is not None else None)
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return _CEFn.apply(logits, target, self.weight, self.ignore_index)
