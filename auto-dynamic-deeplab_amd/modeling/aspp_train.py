"""ASPP head (reference modeling/aspp_train.py:8-61) on the HIP path."""
import torch.nn as nn

from ..module import AddkModule, conv2d
from ..plan import Act


class ASPP_train(AddkModule):
    """ReLU -> {1x1, 3x3 d6m, d12m, d18m}(C->depth)+BN+ReLU  ||  GAP->1x1->BN->ReLU (broadcast)
    -> concat(5*depth) -> 1x1 -> BN.   The concat is virtual (conv1 reads the four branch tensors as K
    slices, each with its BN+ReLU applied in the prologue) and the image-pool branch — constant over
    the image — enters conv1 as a per-image bias computed by a tiny GEMM on the pooled vector."""

    def __init__(self, C, out, BatchNorm, depth=256, conv=nn.Conv2d, eps=1e-5, momentum=0.1, mult=1):
        super().__init__()
        self._C, self._depth = C, depth
        self.global_pooling = nn.AdaptiveAvgPool2d(1)
        self.relu = nn.ReLU(inplace=True)
        self.relu_non_inplace = nn.ReLU()
        self.aspp1 = conv2d(C, depth, 1, bias=False)
        self.aspp2 = conv2d(C, depth, 3, dilation=int(6 * mult), padding=int(6 * mult), bias=False)
        self.aspp3 = conv2d(C, depth, 3, dilation=int(12 * mult), padding=int(12 * mult), bias=False)
        self.aspp4 = conv2d(C, depth, 3, dilation=int(18 * mult), padding=int(18 * mult), bias=False)
        self.aspp5 = conv2d(C, depth, 1, bias=False)
        self.conv1 = conv2d(depth * 5, out, 1, bias=False)
        self.bn1 = BatchNorm(out, eps=eps, momentum=momentum)
        self.aspp1_bn = BatchNorm(depth, eps=eps, momentum=momentum)
        self.aspp2_bn = BatchNorm(depth, eps=eps, momentum=momentum)
        self.aspp3_bn = BatchNorm(depth, eps=eps, momentum=momentum)
        self.aspp4_bn = BatchNorm(depth, eps=eps, momentum=momentum)
        self.aspp5_bn = BatchNorm(depth, eps=eps, momentum=momentum)

    def emit(self, g, x):
        d = self._depth
        branches = []
        for i in range(1, 5):
            branches.append(g.conv_bn([x], getattr(self, 'aspp%d' % i), getattr(self, 'aspp%d_bn' % i),
                                      relu_in=True, post_relu=True))
        pooled = g.gap(x, relu_in=True)                                           # [N,1,1,C]
        x5 = g.conv_bn([pooled], self.aspp5, self.aspp5_bn, relu_in=False, post_relu=True)   # BN over N values (Q7)
        out_c = self.conv1.out_channels
        # image-pool branch folded into conv1: bias_n[n,co] = sum_c W1[co, 4d+c] * x5[n,c]
        bias_n = g.conv([x5], self.conv1.weight, out_c, 1, w_choff=4 * d, cin_total=5 * d)
        slab = rows = None
        if g.training and self.bn1.training:
            slab, rows = g.stats_slab(x.N * x.H * x.W, out_c)
        raw = g.conv(branches, self.conv1.weight, out_c, 1, bias_n=bias_n, stats=slab, cin_total=5 * d)
        return g.bn(raw, self.bn1, slab, rows or 0)
