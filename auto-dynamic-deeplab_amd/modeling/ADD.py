"""Multi-exit densely connected network ADD (reference modeling/ADD.py:14-525) on the HIP path.

Same constructor signatures, attribute names and state_dict keys as the reference; forward()
returns one [N,num_classes,H,W] logits tensor per exit.  The whole network is emitted into ONE static
plan per input shape (plan.py): ~2.4k kernel launches forward for config 2, no torch ops in between.
The Earlier-Decision-Maker gate stays on the host (north_star): dynamic_inference() replays the trunk
segment up to the gate, reads the scalar, and only then launches the chosen exit's kernels."""
import time

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..module import AddkModule, conv2d
from ..plan import Act
from .aspp_train import ASPP_train
from .decoder import Decoder
from .genotypes import PRIMITIVES
from .operations import (OPS, DoubleFactorizedReduce, FactorizedReduce, ReLUConvBN)
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


def scale_dimension(dim, scale):
    """reference ADD.py:65-66."""
    return int((float(dim) - 1.0) * scale + 1.0)


def emit_blocks(g, ops, cell_arch, B, states, C_out):
    """reference ADD.py:97-112.  ops[k] (built from genotype row k) is consumed by the k-th ACTIVE branch
    in ascending (block, branch) order (positional binding).  Each block's branch sum is
    written by one kernel straight into its channel slot of the cell's concat buffer."""
    active = set(int(v) for v in np.asarray(cell_arch)[:, 0])
    s1 = states[1]
    concat = g.tensor(s1.N, s1.H, s1.W, B * C_out)
    offset = used = 0
    n0 = len(states)
    from .operations import SepConv
    fuse_sum = not g.training and not g.want_grad        # inference: the closing SepConv of a block writes the branch sum itself
    for b in range(B):
        todo = []
        for j, h in enumerate(states):
            if offset + j in active:
                todo.append((ops[used], h))               # the binding is positional (Q1); the emission order inside a block is free
                if getattr(g, 'bindings', None) is not None:
                    g.bindings.append((used, offset + j))
                used += 1
        slot = concat.chan(b * C_out, C_out)
        closer = None
        if fuse_sum and len(todo) > 1:
            seps = [i for i, (op, _) in enumerate(todo) if isinstance(op, SepConv)]
            if seps:
                closer = todo.pop(seps[-1])
        terms = [op.emit(g, h) for op, h in todo]
        if closer is not None and not all(t.zero for t in terms):
            states.append(closer[0].emit(g, closer[1], sum_terms=terms, out=slot))
        else:
            if closer is not None:
                terms.append(closer[0].emit(g, closer[1]))
            if not terms or all(t.zero for t in terms):
                raise NotImplementedError('a cell block without any non-zero branch')
            states.append(g.affine_sum(terms, out=slot))
        offset += len(states) - 1
    assert len(states) - n0 == B
    return Act(concat, None, False, g.want_grad)


class Cell(AddkModule):
    """reference ADD.py:14-116."""

    def __init__(self, BatchNorm, B, prev_prev_C, prev_C, cell_arch, network_arch, C_out, downup_sample,
                 dense_in=False, dense_out=True):
        super().__init__()
        eps, momentum = 1e-5, 0.1
        self.cell_arch = cell_arch
        self.downup_sample = downup_sample
        self.B = B
        self.dense_in = dense_in
        self.dense_out = dense_out
        self.C_out = C_out
        self.preprocess = ReLUConvBN(prev_C, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True)
        self._ops = nn.ModuleList()
        if downup_sample == -1:
            self.preprocess = FactorizedReduce(prev_C, C_out, BatchNorm, eps=eps, momentum=momentum)
        elif downup_sample == 1:
            self.scale = 2
        if self.dense_in:
            self.pre_preprocess = nn.ModuleList()
            for c in prev_prev_C:
                self.pre_preprocess.append(ReLUConvBN(c, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True))
            self.pre_preprocess_1x1 = ReLUConvBN(len(prev_prev_C) * C_out, C_out, 1, 1, 0, BatchNorm, eps=eps,
                                                 momentum=momentum, affine=True)
        else:
            self.pre_preprocess = ReLUConvBN(prev_prev_C, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True)
        if self.dense_out:
            self.dense_process = ReLUConvBN(C_out * B, C_out, 1, 1, 0, BatchNorm, eps=eps, momentum=momentum, affine=True)
        for x in np.asarray(self.cell_arch):
            self._ops.append(OPS[PRIMITIVES[int(x[1])]](C_out, 1, BatchNorm, eps=eps, momentum=momentum, affine=True))

    scale_dimension = staticmethod(scale_dimension)

    def emit(self, g, prev_prev_input, prev_input):
        s1 = prev_input
        if self.downup_sample == 1:
            s1 = g.resize(s1, scale_dimension(s1.H, 2), scale_dimension(s1.W, 2))
        s1 = self.preprocess.emit(g, s1)

        def fit(t):                         # ADD.py:84-85,89-90 compare H only
            return g.resize(t, s1.H, s1.W) if t.H != s1.H else t

        if self.dense_in:
            parts = [self.pre_preprocess[i].emit(g, fit(t)) for i, t in enumerate(prev_prev_input)]
            s0 = self.pre_preprocess_1x1.emit(g, parts)     # virtual concat of the lazy parts
        else:
            s0 = self.pre_preprocess.emit(g, fit(prev_prev_input))
        concat = emit_blocks(g, self._ops, self.cell_arch, self.B, [s0, s1], self.C_out)
        if self.dense_out:
            return prev_input, concat, self.dense_process.emit(g, concat)
        return concat


def _make_stems(mod, BatchNorm):
    eps, momentum = 1e-5, 0.1
    mod.stem0 = nn.Sequential(conv2d(3, 64, 3, stride=2, padding=1, bias=False),
                              BatchNorm(64, eps=eps, momentum=momentum), nn.ReLU(inplace=True))
    mod.stem1 = nn.Sequential(conv2d(64, 64, 3, padding=1, bias=False), BatchNorm(64, eps=eps, momentum=momentum))
    mod.stem2 = nn.Sequential(nn.ReLU(inplace=True), conv2d(64, 128, 3, stride=2, padding=1, bias=False),
                              BatchNorm(128, eps=eps, momentum=momentum))


def _emit_stems(mod, g, x):
    """reference ADD.py:283-285.  Returns (stem0, stem1) as cell 0 receives them: stem2's in-place ReLU
    (ADD.py:166) has already been applied to the tensor handed on as prev_prev_input — here a
    pending-ReLU flag on the lazy activation."""
    s = g.conv_bn([x], mod.stem0[0], mod.stem0[1], relu_in=False, post_relu=True)
    s0 = g.conv_bn([s], mod.stem1[0], mod.stem1[1], relu_in=False)
    s1 = g.conv_bn([s0], mod.stem2[1], mod.stem2[2], relu_in=True)
    return Act(s0.raw, s0.bn, True, s0.needs_grad, rs=s0.rs), s1


def _make_heads(mod, network_arch, C_index, F, B, num_classes, BatchNorm, low_level_layer):
    eps, momentum = 1e-5, 0.1
    FB = F * B
    fm = {0: 1, 1: 2, 2: 4, 3: 8}
    last = network_arch[-1]
    mult = {1: 2, 2: 1, 3: 0.5}[last]
    mod.low_level_conv = nn.Sequential(nn.ReLU(), conv2d(F * B * 2 ** network_arch[low_level_layer], 48, 1, bias=False),
                                       BatchNorm(48, eps=eps, momentum=momentum))
    mod.aspp = ASPP_train(F * B * fm[last], 256, BatchNorm, mult=mult)
    mod.conv_aspp = nn.ModuleList()
    for c in C_index:
        d = network_arch[c] - last
        if d == -1:
            mod.conv_aspp.append(FactorizedReduce(FB * 2 ** network_arch[c], FB * 2 ** last, BatchNorm, eps=eps, momentum=momentum))
        elif d == -2:
            mod.conv_aspp.append(DoubleFactorizedReduce(FB * 2 ** network_arch[c], FB * 2 ** last, BatchNorm, eps=eps, momentum=momentum))
        elif d > 0:
            mod.conv_aspp.append(ReLUConvBN(FB * 2 ** network_arch[c], FB * 2 ** last, 1, 1, 0, BatchNorm, eps=eps,
                                            momentum=momentum, affine=True))


def _init_weight(model):
    """reference ADD.py:491-500."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):       # covers SynchronizedBatchNorm2d
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def _aspp_size(size, shift):
    return (int((float(size[0]) - 1.0) * (2 ** (-1 * shift)) + 1.0), int((float(size[1]) - 1.0) * (2 ** (-1 * shift)) + 1.0))


class ADD(AddkModule):
    """reference ADD.py:118-500."""

    def __init__(self, network_arch, C_index, cell_arch, num_classes, args, low_level_layer):
        super().__init__()
        BatchNorm = SynchronizedBatchNorm2d if args.sync_bn == True else nn.BatchNorm2d   # noqa: E712  (ADD.py:128)
        F, B = args.F, args.B
        self.args = args
        self.cells = nn.ModuleList()
        self.cell_arch = torch.from_numpy(np.asarray(cell_arch))
        self._num_classes = num_classes
        self.low_level_layer = low_level_layer
        self.decoder = Decoder(num_classes, BatchNorm)
        self.network_arch = list(int(v) for v in network_arch)
        self.num_net = len(self.network_arch)
        self.C_index = list(C_index)
        FB = F * B
        fm = {0: 1, 1: 2, 2: 4, 3: 8}
        _make_stems(self, BatchNorm)
        na = self.network_arch
        for i in range(self.num_net):
            level, prev_level, prev_prev_level = na[i], na[i - 1], na[i - 2]
            downup_sample = int(prev_level - level)
            if i == 0:
                cell = Cell(BatchNorm, B, 64, 128, self.cell_arch, level, F * fm[level], int(0 - level), dense_in=False, dense_out=True)
            elif i == 1:
                cell = Cell(BatchNorm, B, 128, FB * fm[prev_level], self.cell_arch, level, F * fm[level], downup_sample,
                            dense_in=False, dense_out=True)
            elif i == 2:
                cell = Cell(BatchNorm, B, FB * fm[prev_prev_level], FB * fm[prev_level], self.cell_arch, level, F * fm[level],
                            downup_sample, dense_in=False, dense_out=True)
            else:
                dense_channel_list = [F * fm[s] for s in na[:i - 1]]
                cell = Cell(BatchNorm, B, dense_channel_list, FB * fm[prev_level], self.cell_arch, level, F * fm[level],
                            downup_sample, dense_in=True, dense_out=(i < self.num_net - 2))
            self.cells += [cell]
        _init_weight(self)
        self.pooling = nn.MaxPool2d(3, stride=2)
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.relu = nn.ReLU()
        _make_heads(self, na, self.C_index, F, B, num_classes, BatchNorm, low_level_layer)
        _init_weight(self)

    _init_weight = _init_weight

    # ---- trunk shared by forward / get_feature / dynamic_inference (ADD.py:283-308) ----
    def _trunk(self, g, x):
        g.tag = 'stem'
        two = list(_emit_stems(self, g, x))
        dense, low, cur = [], None, None
        for i in range(self.num_net):
            g.tag = 'cell'
            if i < 3:
                two[0], two[1], fm = self.cells[i].emit(g, two[0], two[1])
                dense.append(fm)
                if i == 2:
                    cur = two[1]
            elif i < self.num_net - 2:
                _, cur, fm = self.cells[i].emit(g, list(dense[:-1]), cur)
                dense.append(fm)
            elif i == self.num_net - 1:
                cur = self.cells[i].emit(g, list(dense), cur)
            else:
                cur = self.cells[i].emit(g, list(dense[:-1]), cur)
            if i == self.low_level_layer:
                lc = self.low_level_conv
                g.tag = 'low'
                low = g.conv_bn([two[1]], lc[1], lc[2], relu_in=True)
            y = cur if i > 2 else two[1]
            g.tag = 'head'
            got = yield i, y, low
            if got is not None:                 # EDM's in-place ReLU mutated the feature (Q3)
                if i > 2:
                    cur = got
                else:
                    two[1] = got

    def _head(self, g, y, low, size, aspp_size, it, level, resize=True, adapt=True):
        g.tag = 'aspp'
        if resize and (y.H < aspp_size[0] or y.W < aspp_size[1]):
            y = g.resize(y, aspp_size[0], aspp_size[1])
        if adapt and level != self.network_arch[-1]:
            y = self.conv_aspp[it].emit(g, y)
        y = self.aspp.emit(g, y)
        g.tag = 'decoder'
        return self.decoder.emit(g, y, low, size)

    def emit(self, g, x):
        """reference ADD.py:277-325."""
        size = (x.H, x.W)
        aspp_size = _aspp_size(size, self.network_arch[-1] + 2)
        it, out = 0, []
        for i, y, low in self._trunk(g, x):
            if i in self.C_index or i == self.num_net - 1:
                lvl = self.network_arch[i]
                out.append(self._head(g, y, low, size, aspp_size, it, lvl))
                if lvl != self.network_arch[-1]:
                    it += 1
        return out

    def forward(self, x):
        return self.run_plan(self.emit, (x,))

    def _emit_get_feature(self, g, x):
        """reference ADD.py:327-377 — first exit only, aspp_size from 2^-last (Q5)."""
        size = (x.H, x.W)
        aspp_size = _aspp_size(size, self.network_arch[-1])
        for i, y, low in self._trunk(g, x):
            if i in self.C_index:
                return [self._head(g, y, low, size, aspp_size, 0, self.network_arch[i]), y]
        raise RuntimeError('no exit in C_index')

    def get_feature(self, x):
        out, feat = self.run_plan(self._emit_get_feature, (x,), tag='get_feature')
        return out, feat

    def dynamic_inference(self, x, threshold=1.0, confidence='edm', edm=False):
        """reference ADD.py:379-438 (working 'edm' gate only).  Three plan segments share one buffer
        set: trunk up to the gate + EDM, the early head, and the remaining cells + final head; the host reads the
        EDM scalar (one D2H sync, as in the reference's `if confidence_value > threshold`) and launches one."""
        if confidence != 'edm':
            raise NotImplementedError("only confidence='edm' is a working reference path (ADD.py:465-488 return features)")
        torch.cuda.synchronize()
        tic = time.perf_counter()
        plan = self._dynamic_plan(x, edm)
        y, earlier_exit, conf = plan.run(x, threshold)
        torch.cuda.synchronize()
        return y, earlier_exit, time.perf_counter() - tic, conf

    def _dynamic_plan(self, x, edm):
        from ..dynamic import DynamicPlan
        key = ('dyn', tuple(x.shape), id(edm), int(L.load().addk_get_conv_precision()))
        plans = self._plans()
        p = plans.get(key)
        if p is None or not p.check_params():
            p = plans[key] = DynamicPlan(self, edm, x)
        return p


class EDM(AddkModule):
    """reference ADD.py:502-525 — (in-place) ReLU, conv3x3 s2 400->128, ReLU, GAP, MLP 128-64-32-1.
    Linear layers run as 1x1 GEMMs on the [N,1,1,C] pooled vector."""

    def __init__(self):
        super().__init__()
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.relu = nn.ReLU(inplace=True)
        self.conv = conv2d(400, 128, 3, stride=2, padding=1, bias=False)
        self.edm = nn.Sequential(nn.Linear(128, 64), nn.ReLU(inplace=True), nn.Linear(64, 32), nn.ReLU(inplace=True),
                                 nn.Linear(32, 1))

    def emit(self, g, x, host_out=None):
        """Inference: ONE launch (plan.Graph.edm_head, csrc/edm.hip).  With gradients (train_edm.py) or shapes the fused kernel does
        not take: the generic launches below."""
        fused = g.edm_head(x, self.conv.weight, (self.edm[0], self.edm[2], self.edm[4]), host_out)
        if fused is not None:
            g.edm_fused = True
            return fused
        y = Act(g.conv([x], self.conv.weight, 128, 3, 2, 1, 1, relu_in=True), None, True, g.want_grad)
        v = g.gap(y)
        for i, lin in enumerate((self.edm[0], self.edm[2], self.edm[4])):
            raw = g.conv([v], lin.weight, lin.out_features, 1, bias=lin.bias)
            v = Act(raw, None, i < 2, g.want_grad)
        return v

    def forward(self, x):
        x = x.squeeze(1)                      # ADD.py:516: train_edm.py feeds [bs, 1, 400, h, w] (features stored per batch of 1)
        out = self.run_plan(self.emit, (x,))
        return out.reshape(out.shape[0], -1)
