"""Decoder head (reference modeling/decoder.py:6-30) on the HIP path."""
import torch.nn as nn

from ..module import AddkModule, conv2d
from ..plan import Act


class Decoder(AddkModule):
    """bilinear up to the low-level map -> virtual concat (256+48) -> ReLU,3x3,BN,ReLU,3x3,BN,ReLU,1x1(+bias)
    -> bilinear to the input size, written as NCHW logits."""

    def __init__(self, n_class, BatchNorm):
        super().__init__()
        eps, momentum = 1e-5, 0.1
        self._conv = nn.Sequential(
            nn.ReLU(inplace=True),
            conv2d(304, 256, 3, stride=1, padding=1, bias=False),
            BatchNorm(256, eps=eps, momentum=momentum),
            nn.ReLU(inplace=True),
            conv2d(256, 256, 3, stride=1, padding=1, bias=False),
            BatchNorm(256, eps=eps, momentum=momentum),
            nn.ReLU(inplace=True),
            conv2d(256, n_class, 1, stride=1, bias=True))

    def emit(self, g, x, low_level, size):
        if x.H != low_level.H:                      # decoder.py:24-25 compares H only
            x = g.resize(x, low_level.H, low_level.W)
        c = self._conv
        y = g.conv_bn([x, low_level], c[1], c[2], relu_in=True, post_relu=True)
        y = g.conv_bn([y], c[4], c[5], relu_in=False, post_relu=True)
        ncls = c[7].out_channels
        logits = g.conv([y], c[7].weight, ncls, 1, bias=c[7].bias)
        la = Act(logits, None, False, g.want_grad)
        return g.resize_to_nchw(la, size[0], size[1])

    def forward(self, x, low_level, size):
        size = (int(size[0]), int(size[1]))
        return self.run_plan(lambda g, a, b: self.emit(g, a, b, size), (x, low_level), tag=size)
