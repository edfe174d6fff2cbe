"""Baselin_Model (reference modeling/baseline_model.py:14-265): ADD without dense connections
(BASELINE config 1: searched_baseline network_path + genotype_2, exit=last)."""
import numpy as np
import torch
import torch.nn as nn

from ..module import AddkModule
from .ADD import (_aspp_size, _emit_stems, _init_weight, _make_heads, _make_stems, emit_blocks, scale_dimension)
from .decoder import Decoder
from .genotypes import PRIMITIVES
from .operations import OPS, FactorizedReduce, ReLUConvBN
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


class Cell_baseline(AddkModule):
    """reference baseline_model.py:14-90."""

    def __init__(self, BatchNorm, B, prev_prev_C, prev_C, cell_arch, network_arch, C_out, downup_sample):
        super().__init__()
        eps, momentum = 1e-5, 0.1
        self.cell_arch, self.downup_sample, self.B, self.C_out = cell_arch, downup_sample, B, C_out
        self.pre_preprocess = ReLUConvBN(prev_prev_C, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True)
        self.preprocess = ReLUConvBN(prev_C, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True)
        self._ops = nn.ModuleList()
        if downup_sample == -1:
            self.preprocess = FactorizedReduce(prev_C, C_out, BatchNorm, eps=eps, momentum=momentum)
        elif downup_sample == 1:
            self.scale = 2
        for x in np.asarray(self.cell_arch):
            self._ops.append(OPS[PRIMITIVES[int(x[1])]](C_out, 1, BatchNorm, eps=eps, momentum=momentum, affine=True))

    def emit(self, g, prev_prev_input, prev_input):
        s1 = prev_input
        if self.downup_sample == 1:
            s1 = g.resize(s1, scale_dimension(s1.H, 2), scale_dimension(s1.W, 2))
        s1 = self.preprocess.emit(g, s1)
        s0 = prev_prev_input
        if s0.H != s1.H:
            s0 = g.resize(s0, s1.H, s1.W)
        s0 = self.pre_preprocess.emit(g, s0)
        return prev_input, emit_blocks(g, self._ops, self.cell_arch, self.B, [s0, s1], self.C_out)


class Baselin_Model(AddkModule):
    """reference baseline_model.py:93-265 (class name spelled as in the reference)."""

    def __init__(self, network_arch, C_index, cell_arch, num_classes, args, low_level_layer):
        super().__init__()
        BatchNorm = SynchronizedBatchNorm2d if args.sync_bn == True else nn.BatchNorm2d   # noqa: E712
        F, B = args.F, args.B
        self.args = args
        self.cells = nn.ModuleList()
        self.cell_arch = torch.from_numpy(np.asarray(cell_arch))
        self._num_classes, self.low_level_layer = num_classes, low_level_layer
        self.decoder = Decoder(num_classes, BatchNorm)
        self.network_arch = list(int(v) for v in network_arch)
        self.num_net, self.C_index = len(self.network_arch), list(C_index)
        FB = F * B
        fm = {0: 1, 1: 2, 2: 4, 3: 8}
        _make_stems(self, BatchNorm)
        na = self.network_arch
        for i in range(self.num_net):
            level, prev_level, prev_prev_level = na[i], na[i - 1], na[i - 2]
            downup_sample = int(prev_level - level)
            if i == 0:
                ppc, pc, downup_sample = 64, 128, int(0 - level)
            elif i == 1:
                ppc, pc = 128, FB * fm[prev_level]
            else:
                ppc, pc = FB * fm[prev_prev_level], FB * fm[prev_level]
            self.cells += [Cell_baseline(BatchNorm, B, ppc, pc, self.cell_arch, level, F * fm[level], downup_sample)]
        _init_weight(self)
        self.pooling = nn.MaxPool2d(3, stride=2)
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.relu = nn.ReLU()
        _make_heads(self, na, self.C_index, F, B, num_classes, BatchNorm, low_level_layer)
        _init_weight(self)

    _init_weight = _init_weight

    def emit(self, g, x):
        """reference baseline_model.py:224-254."""
        size = (x.H, x.W)
        aspp_size = _aspp_size(size, self.network_arch[-1] + 2)
        two = list(_emit_stems(self, g, x))
        it, out, low = 0, [], None
        for i in range(self.num_net):
            two = list(self.cells[i].emit(g, two[0], two[1]))
            if i == self.low_level_layer:
                lc = self.low_level_conv
                low = g.conv_bn([two[1]], lc[1], lc[2], relu_in=True)
            if i in self.C_index or i == self.num_net - 1:
                y = two[1]
                if y.H < aspp_size[0] or y.W < aspp_size[1]:
                    y = g.resize(y, aspp_size[0], aspp_size[1])
                if self.network_arch[i] != self.network_arch[-1]:
                    y = self.conv_aspp[it].emit(g, y)
                    it += 1
                out.append(self.decoder.emit(g, self.aspp.emit(g, y), low, size))
        return out

    def forward(self, x):
        return self.run_plan(self.emit, (x,))
