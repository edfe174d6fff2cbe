"""Evaluator of the reference (utils/metrics.py:4-52) with the confusion matrix accumulated on the GPU by an
integer-atomic kernel, and argmax over the NCHW logits fused in one pass (eval.py:183-185 does
`output.data.cpu().numpy(); np.argmax` on the host)."""
import torch

from . import _lib as L
from . import plan as _plan


def argmax_logits(logits):
    lib = L.load()
    _plan.require_device(logits)
    logits = logits.contiguous().float()
    N, Cc, H, W = logits.shape
    out = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
    L.check(lib.addk_argmax_nchw(logits.data_ptr(), N, Cc, H * W, out.data_ptr(), _plan.current_stream()), 'argmax')
    return out


class Evaluator(object):
    def __init__(self, num_class, device='cuda'):
        self.num_class = num_class
        self._cm = torch.zeros((num_class, num_class), dtype=torch.int64, device=device)

    @property
    def confusion_matrix(self):
        return self._cm.float()

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        lib = L.load()
        gt = gt_image.to(self._cm.device).long().contiguous()
        pr = pre_image.to(self._cm.device).long().contiguous()
        L.check(lib.addk_confusion(gt.data_ptr(), pr.data_ptr(), gt.numel(), self.num_class, self._cm.data_ptr(),
                                   _plan.current_stream()), 'confusion')

    def reset(self):
        self._cm.zero_()

    @staticmethod
    def torch_nanmean(x):
        ok = ~torch.isnan(x)
        return torch.where(ok, x, torch.zeros_like(x)).sum() / ok.sum()

    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        return torch.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        return self.torch_nanmean(torch.diag(cm) / cm.sum(dim=1))

    def Mean_Intersection_over_Union(self):
        cm = self.confusion_matrix
        d = torch.diag(cm)
        return self.torch_nanmean(d / (cm.sum(dim=1) + cm.sum(dim=0) - d)).item()

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        d = torch.diag(cm)
        freq = cm.sum(dim=1) / cm.sum()
        iu = d / (cm.sum(dim=1) + cm.sum(dim=0) - d)
        return (freq[freq > 0] * iu[freq > 0]).sum()
