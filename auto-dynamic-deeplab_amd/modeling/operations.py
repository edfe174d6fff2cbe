"""Searched-cell operators on the HIP path — same registry surface as the reference's
modeling/operations.py (OPS[name](C, stride, BatchNorm, eps, momentum, affine) -> nn.Module whose parameters
live under `.op` / `.conv_1` / `.conv_2` / `.bn`, so reference checkpoints load unchanged).

Every module's forward() replays a plan of addk kernels (module.py); `emit` is the graph-builder
protocol used when the module is part of a larger plan (Cell, ADD)."""
import math

import torch
import torch.nn as nn

from .. import _lib as L
from ..module import AddkModule, conv2d
from ..plan import Act, Vec
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d  # noqa: F401  (re-exported like the reference)


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class ReLUConvBN(AddkModule):
    """reference operations.py:18-29.  One implicit-GEMM launch: ReLU (and the producer's BN) in the input
    prologue, BN statistics in the epilogue.  `emit` also accepts a list of Acts = virtual concat."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, BatchNorm, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.ReLU(inplace=False),
            conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine))

    def emit(self, g, x):
        return g.conv_bn(_as_list(x), self.op[1], self.op[2], relu_in=True)


class DilConv(AddkModule):
    """reference operations.py:32-43 — ReLU, DENSE dilated conv, BN."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, BatchNorm, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.ReLU(inplace=False),
            conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=False),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine))

    def emit(self, g, x):
        return g.conv_bn([x], self.op[1], self.op[2], relu_in=True)


class SepConv(AddkModule):
    """reference operations.py:46-62 — (ReLU, depthwise, pointwise, BN) x 2."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, BatchNorm, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.ReLU(inplace=False),
            conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, groups=C_in, bias=False),
            conv2d(C_out, C_out, 1, padding=0, bias=False),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine),
            nn.ReLU(inplace=False),
            conv2d(C_out, C_out, kernel_size, stride=1, padding=padding, groups=C_in, bias=False),
            conv2d(C_out, C_out, 1, padding=0, bias=False),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine))
        assert C_in == C_out, 'the path only instantiates depthwise SepConv with C_in == C_out (operations.py:12-13)'

    def emit(self, g, x, sum_terms=None, out=None):
        """Each half is one fused launch (plan.Graph.sep_half).  `sum_terms` / `out` (inference): this op closes a cell block —
        its second half adds the other branches and writes the block sum (ADD.py:108) itself."""
        y = g.sep_half(x, self.op[1], self.op[2], self.op[3])
        return g.sep_half(y, self.op[5], self.op[6], self.op[7], sum_terms=sum_terms, out=out)


class Identity(AddkModule):
    """reference operations.py:65-71."""

    def emit(self, g, x):
        return x


class Zero(AddkModule):
    """reference operations.py:74-83."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def emit(self, g, x):
        s = self.stride
        return g.zeros(x.N, (x.H + s - 1) // s, (x.W + s - 1) // s, x.C)


class _Pool3(AddkModule):
    """nn.MaxPool2d(3, stride, padding=1) / nn.AvgPool2d(3, stride, padding=1, count_include_pad=False)
    of the registry (reference operations.py:9-10); cold on every shipped genotype."""

    def __init__(self, stride, mode):
        super().__init__()
        self.stride, self.mode = stride, mode

    def emit(self, g, x):
        return g.pool3(x, self.stride, self.mode)


class FactorizedReduce(AddkModule):
    """reference operations.py:86-101 — ReLU, two stride-2 1x1 convs (the second on the input shifted by one
    pixel: expressed as pad=-1), channel concat (both write halves of one buffer), one BN."""
    _stride, _shift = 2, 1

    def __init__(self, C_in, C_out, BatchNorm, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        assert C_out % 2 == 0
        s = self._stride
        self.relu = nn.ReLU(inplace=False)
        self.conv_1 = conv2d(C_in, C_out // 2, 1, stride=s, padding=0, bias=False)
        self.conv_2 = conv2d(C_in, C_out // 2, 1, stride=s, padding=0, bias=False)
        self.bn = self._make_bn(BatchNorm, C_out, eps, momentum, affine)
        self.pad = nn.ConstantPad2d((0, self._shift, 0, self._shift), 0)

    @staticmethod
    def _make_bn(BatchNorm, C_out, eps, momentum, affine):
        return BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine)

    def emit(self, g, x):
        s, sh = self._stride, self._shift
        Cout = self.bn.num_features
        OH, OW = (x.H - 1) // s + 1, (x.W - 1) // s + 1
        raw = g.tensor(x.N, OH, OW, Cout)
        slab = rows = None
        if g.training and self.bn.training:
            slab, rows = g.stats_slab(x.N * OH * OW, Cout)
        half = Cout // 2
        for i, (conv, pad) in enumerate(((self.conv_1, 0), (self.conv_2, -sh))):
            st = None
            if slab is not None:
                st = Vec(slab, i * half * 4, 0)        # fp64 (sum, sumsq) pairs: 4 floats per channel
            g.conv([x], conv.weight, half, 1, s, pad, 1, relu_in=True, out=raw.chan(i * half, half), stats=st,
                   stats_ld=Cout, out_hw=(OH, OW))
        return g.bn(raw, self.bn, slab, rows or 0)


class DoubleFactorizedReduce(FactorizedReduce):
    """reference operations.py:104-119 (stride 4, shift 2; its BN is built with ctor defaults)."""
    _stride, _shift = 4, 2

    @staticmethod
    def _make_bn(BatchNorm, C_out, eps, momentum, affine):
        return BatchNorm(C_out, affine=affine)


OPS = {
    'none': lambda C, stride, BatchNorm, eps, momentum, affine: Zero(stride),
    'avg_pool_3x3': lambda C, stride, BatchNorm, eps, momentum, affine: _Pool3(stride, 1),
    'max_pool_3x3': lambda C, stride, BatchNorm, eps, momentum, affine: _Pool3(stride, 0),
    'skip_connect': lambda C, stride, BatchNorm, eps, momentum, affine: Identity(),
    'sep_conv_3x3': lambda C, stride, BatchNorm, eps, momentum, affine: SepConv(C, C, 3, stride, 1, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'sep_conv_5x5': lambda C, stride, BatchNorm, eps, momentum, affine: SepConv(C, C, 5, stride, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'dil_conv_3x3': lambda C, stride, BatchNorm, eps, momentum, affine: DilConv(C, C, 3, stride, 2, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'dil_conv_5x5': lambda C, stride, BatchNorm, eps, momentum, affine: DilConv(C, C, 5, stride, 4, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
}


def _scratch(x, n):
    return torch.empty(n, dtype=torch.float32, device=x.device)


def normalized_shannon_entropy(x, num_class=19):
    """reference operations.py:161-170: sum over batch and pixels of -sum_c p log p / log(num_class),
    divided by H*W; one fused kernel pass over the NCHW logits, then .item() (the gate's D2H sync)."""
    lib = L.load()
    x = x.detach().float().contiguous()
    N, Cc, H, W = x.shape
    out, ws = _scratch(x, 1), _scratch(x, 1024)
    L.check(lib.addk_entropy_sum(x.data_ptr(), N, Cc, H * W, out.data_ptr(), ws.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), 'entropy_sum')
    return out.item() / math.log(num_class) / (H * W)


def confidence_max(x, thresold, num_class=19):
    """reference operations.py:172-180 (host helper; unused by the working 'edm' gate)."""
    p = torch.softmax(x, dim=1).max(dim=1)[0]
    return int((p > thresold).sum()) / (x.shape[2] * x.shape[3])
