"""PyTorch-CPU fp32 restatement of the reference hot path (TEST INFRASTRUCTURE).

Written from SURVEY.md §8(a) and a reading of /root/reference/modeling/*.py.
Module attribute names follow the reference because `state_dict` key names are
part of the drop-in boundary: the same state_dict loads into
the reference, this oracle and the HIP product.

Every arithmetic primitive here is a documented PyTorch op (cross-correlation
conv with zero padding, training-mode batch norm with biased variance for
normalisation and unbiased variance for running_var, half-pixel bilinear
interpolation with align_corners=False).
"""
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

# reference: modeling/genotypes.py:5-14 — index == genotype column 1
PRIMITIVES = ['none', 'max_pool_3x3', 'avg_pool_3x3', 'skip_connect',
              'sep_conv_3x3', 'sep_conv_5x5', 'dil_conv_3x3', 'dil_conv_5x5']

_EPS, _MOM = 1e-5, 0.1


def _conv(ci, co, k, stride=1, pad=0, dil=1, groups=1, bias=False):
    return nn.Conv2d(ci, co, k, stride=stride, padding=pad, dilation=dil,
                     groups=groups, bias=bias)


def _bilinear(x, size):
    """F.interpolate(mode='bilinear') as called at ADD.py:76,84,89,317 and
    decoder.py:24,28: align_corners=False, no antialias."""
    return F.interpolate(x, size=list(size), mode='bilinear', align_corners=False)


class ReLUConvBN(nn.Module):
    """operations.py:18-29 — ReLU -> Conv(k, bias=False) -> BN."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, BatchNorm,
                 eps=_EPS, momentum=_MOM, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.ReLU(),
            _conv(C_in, C_out, kernel_size, stride, padding),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine))

    def forward(self, x):
        return self.op(x)


class DilConv(nn.Module):
    """operations.py:32-43 — ReLU -> DENSE dilated conv (groups=1) -> BN."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation,
                 BatchNorm, eps=_EPS, momentum=_MOM, affine=True):
        super().__init__()
        self.op = nn.Sequential(
            nn.ReLU(),
            _conv(C_in, C_out, kernel_size, stride, padding, dilation),
            BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine))

    def forward(self, x):
        return self.op(x)


class SepConv(nn.Module):
    """operations.py:46-62 — (ReLU, depthwise k(stride), pointwise, BN) twice;
    the second depthwise has stride 1."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, BatchNorm,
                 eps=_EPS, momentum=_MOM, affine=True):
        super().__init__()
        layers = []
        for s, ci in ((stride, C_in), (1, C_out)):
            layers += [nn.ReLU(),
                       _conv(ci, C_out, kernel_size, s, padding, groups=C_in),
                       _conv(C_out, C_out, 1),
                       BatchNorm(C_out, eps=eps, momentum=momentum, affine=affine)]
        self.op = nn.Sequential(*layers)

    def forward(self, x):
        return self.op(x)


class Identity(nn.Module):
    """operations.py:65-71."""

    def forward(self, x):
        return x


class Zero(nn.Module):
    """operations.py:74-83 — x*0, strided slice first when stride > 1."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        if self.stride != 1:
            x = x[:, :, ::self.stride, ::self.stride]
        return x * 0.0


class _Reduce(nn.Module):
    """Shared body of FactorizedReduce (operations.py:86-101, stride 2, shift 1)
    and DoubleFactorizedReduce (:104-119, stride 4, shift 2): ReLU, then two
    strided 1x1 convs, the second on the input shifted by (shift, shift) with
    zero fill at the far edge, channel-concatenated, then BN."""

    def __init__(self, C_in, C_out, BatchNorm, stride, shift, bn_kwargs):
        super().__init__()
        assert C_out % 2 == 0
        self.conv_1 = _conv(C_in, C_out // 2, 1, stride)
        self.conv_2 = _conv(C_in, C_out // 2, 1, stride)
        self.bn = BatchNorm(C_out, **bn_kwargs)
        self._shift = shift

    def forward(self, x):
        x = F.relu(x)
        s = self._shift
        shifted = F.pad(x, (0, s, 0, s))[:, :, s:, s:]
        return self.bn(torch.cat([self.conv_1(x), self.conv_2(shifted)], dim=1))


class FactorizedReduce(_Reduce):
    def __init__(self, C_in, C_out, BatchNorm, eps=_EPS, momentum=_MOM, affine=True):
        super().__init__(C_in, C_out, BatchNorm, 2, 1,
                         dict(eps=eps, momentum=momentum, affine=affine))


class DoubleFactorizedReduce(_Reduce):
    def __init__(self, C_in, C_out, BatchNorm, eps=_EPS, momentum=_MOM, affine=True):
        # operations.py:111 builds this BN with constructor defaults (== 1e-5 / 0.1)
        super().__init__(C_in, C_out, BatchNorm, 4, 2, dict(affine=affine))


# reference: modeling/operations.py:7-16
OPS = {
    'none': lambda C, stride, BatchNorm, eps, momentum, affine: Zero(stride),
    'avg_pool_3x3': lambda C, stride, BatchNorm, eps, momentum, affine:
        nn.AvgPool2d(3, stride=stride, padding=1, count_include_pad=False),
    'max_pool_3x3': lambda C, stride, BatchNorm, eps, momentum, affine:
        nn.MaxPool2d(3, stride=stride, padding=1),
    'skip_connect': lambda C, stride, BatchNorm, eps, momentum, affine: Identity(),
    'sep_conv_3x3': lambda C, stride, BatchNorm, eps, momentum, affine:
        SepConv(C, C, 3, stride, 1, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'sep_conv_5x5': lambda C, stride, BatchNorm, eps, momentum, affine:
        SepConv(C, C, 5, stride, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'dil_conv_3x3': lambda C, stride, BatchNorm, eps, momentum, affine:
        DilConv(C, C, 3, stride, 2, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
    'dil_conv_5x5': lambda C, stride, BatchNorm, eps, momentum, affine:
        DilConv(C, C, 5, stride, 4, 2, BatchNorm, eps=eps, momentum=momentum, affine=affine),
}


class ASPP_train(nn.Module):
    """aspp_train.py:8-61 — ReLU, four conv branches (1x1, 3x3 d6m, d12m, d18m)
    + image-pool branch (GAP -> 1x1 -> BN -> ReLU -> broadcast: bilinear
    align_corners=True from 1x1 is a constant fill), concat 5*depth -> 1x1 -> BN
    (no trailing ReLU)."""

    def __init__(self, C, out, BatchNorm, depth=256, conv=nn.Conv2d, eps=_EPS,
                 momentum=_MOM, mult=1):
        super().__init__()
        self.aspp1 = _conv(C, depth, 1)
        for idx, rate in ((2, 6), (3, 12), (4, 18)):
            d = int(rate * mult)
            setattr(self, 'aspp%d' % idx, _conv(C, depth, 3, 1, d, d))
        self.aspp5 = _conv(C, depth, 1)
        self.conv1 = _conv(depth * 5, out, 1)
        self.bn1 = BatchNorm(out, eps=eps, momentum=momentum)
        for idx in range(1, 6):
            setattr(self, 'aspp%d_bn' % idx, BatchNorm(depth, eps=eps, momentum=momentum))

    def forward(self, x):
        x = F.relu(x)
        h, w = x.shape[2:]
        ys = [F.relu(getattr(self, 'aspp%d_bn' % i)(getattr(self, 'aspp%d' % i)(x)))
              for i in range(1, 5)]
        pooled = F.relu(self.aspp5_bn(self.aspp5(x.mean(dim=(2, 3), keepdim=True))))
        ys.append(pooled.expand(-1, -1, h, w))
        return self.bn1(self.conv1(torch.cat(ys, dim=1)))


class Decoder(nn.Module):
    """decoder.py:6-30 — resize ASPP output to the low-level map, concat
    (256+48=304), ReLU,3x3,BN,ReLU,3x3,BN,ReLU,1x1(+bias), resize to `size`."""

    def __init__(self, n_class, BatchNorm):
        super().__init__()
        self._conv = nn.Sequential(
            nn.ReLU(),
            _conv(304, 256, 3, 1, 1), BatchNorm(256, eps=_EPS, momentum=_MOM),
            nn.ReLU(),
            _conv(256, 256, 3, 1, 1), BatchNorm(256, eps=_EPS, momentum=_MOM),
            nn.ReLU(),
            _conv(256, n_class, 1, bias=True))

    def forward(self, x, low_level, size):
        if x.shape[2] != low_level.shape[2]:       # decoder.py:24-25 tests H only
            x = _bilinear(x, low_level.shape[2:])
        x = self._conv(torch.cat((x, low_level), dim=1))
        return _bilinear(x, size)


def _scale_dim(dim, scale):
    """ADD.py:65-66."""
    return int((float(dim) - 1.0) * scale + 1.0)


def _run_blocks(ops, cell_arch, B, states):
    """ADD.py:97-112 / baseline_model.py:74-89.  The k-th entry of `ops` (built
    from genotype row k) is consumed by the k-th ACTIVE branch in ascending
    (block, branch) order — positional binding; membership is tested
    against column 0 of the genotype (Q2)."""
    active = set(int(v) for v in cell_arch[:, 0])
    offset = used = 0
    for _ in range(B):
        acc = None
        for j, h in enumerate(states):
            if offset + j in active:
                y = ops[used](h)
                used += 1
                acc = y if acc is None else acc + y
        if acc is None:          # python sum([]) == 0 in the reference
            acc = 0
        offset += len(states)
        states.append(acc)
    return torch.cat(states[-B:], dim=1)


class Cell(nn.Module):
    """ADD.py:14-116."""

    def __init__(self, BatchNorm, B, prev_prev_C, prev_C, cell_arch, network_arch,
                 C_out, downup_sample, dense_in=False, dense_out=True):
        super().__init__()
        kw = dict(eps=_EPS, momentum=_MOM)
        self.cell_arch, self.B = cell_arch, B
        self.downup_sample, self.dense_in, self.dense_out = downup_sample, dense_in, dense_out
        if downup_sample == -1:
            self.preprocess = FactorizedReduce(prev_C, C_out, BatchNorm, **kw)
        else:
            self.preprocess = ReLUConvBN(prev_C, C_out, 1, 1, 0, BatchNorm, affine=True, **kw)
        self._ops = nn.ModuleList()
        if dense_in:
            self.pre_preprocess = nn.ModuleList(
                ReLUConvBN(c, C_out, 1, 1, 0, BatchNorm, affine=True, **kw) for c in prev_prev_C)
            self.pre_preprocess_1x1 = ReLUConvBN(len(prev_prev_C) * C_out, C_out, 1, 1, 0,
                                                 BatchNorm, affine=True, **kw)
        else:
            self.pre_preprocess = ReLUConvBN(prev_prev_C, C_out, 1, 1, 0, BatchNorm,
                                             affine=True, **kw)
        if dense_out:
            self.dense_process = ReLUConvBN(C_out * B, C_out, 1, 1, 0, BatchNorm,
                                            affine=True, **kw)
        for row in cell_arch:
            self._ops.append(OPS[PRIMITIVES[int(row[1])]](C_out, 1, BatchNorm, affine=True, **kw))

    def forward(self, prev_prev_input, prev_input):
        s1 = prev_input
        if self.downup_sample == 1:
            s1 = _bilinear(s1, (_scale_dim(s1.shape[2], 2), _scale_dim(s1.shape[3], 2)))
        s1 = self.preprocess(s1)
        hw = s1.shape[2:]

        def fit(t):                              # ADD.py:84-85,89-90 compare H only
            return _bilinear(t, hw) if t.shape[2] != hw[0] else t

        if self.dense_in:
            parts = [self.pre_preprocess[i](fit(t)) for i, t in enumerate(prev_prev_input)]
            s0 = self.pre_preprocess_1x1(torch.cat(parts, dim=1))
        else:
            s0 = self.pre_preprocess(fit(prev_prev_input))
        concat = _run_blocks(self._ops, self.cell_arch, self.B, [s0, s1])
        if self.dense_out:
            return prev_input, concat, self.dense_process(concat)
        return concat


def _kaiming_init(model):
    """ADD.py:491-500 (called twice in the reference ctor; fixtures carry
    explicit weights so RNG order never matters)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


_FM = {0: 1, 1: 2, 2: 4, 3: 8}


def _make_stems(mod, BatchNorm):
    """ADD.py:154-169 / baseline_model.py:131-146."""
    kw = dict(eps=_EPS, momentum=_MOM)
    mod.stem0 = nn.Sequential(_conv(3, 64, 3, 2, 1), BatchNorm(64, **kw), nn.ReLU())
    mod.stem1 = nn.Sequential(_conv(64, 64, 3, 1, 1), BatchNorm(64, **kw))
    mod.stem2 = nn.Sequential(nn.ReLU(), _conv(64, 128, 3, 2, 1), BatchNorm(128, **kw))


def _run_stems(mod, x):
    """Returns (stem0, stem1) as cell 0 sees them.  stem2's first layer is
    ReLU(inplace=True) in the reference (ADD.py:166), so the tensor cell 0
    receives as prev_prev_input is relu(stem1(...)) — SURVEY Q3."""
    pre = mod.stem1(mod.stem0(x))
    s0 = F.relu(pre)
    s1 = mod.stem2[2](mod.stem2[1](s0))
    return s0, s1


def _make_heads(mod, network_arch, C_index, F_, B, num_classes, BatchNorm, low_level_layer):
    """ADD.py:242-273 / baseline_model.py:188-220."""
    kw = dict(eps=_EPS, momentum=_MOM)
    FB = F_ * B
    last = network_arch[-1]
    mult = {1: 2, 2: 1, 3: 0.5}[last]
    mod.low_level_conv = nn.Sequential(
        nn.ReLU(), _conv(FB * 2 ** network_arch[low_level_layer], 48, 1), BatchNorm(48, **kw))
    mod.aspp = ASPP_train(FB * _FM[last], 256, BatchNorm, mult=mult)
    mod.conv_aspp = nn.ModuleList()
    for c in C_index:
        diff = network_arch[c] - last
        cin, cout = FB * 2 ** network_arch[c], FB * 2 ** last
        if diff == -1:
            mod.conv_aspp.append(FactorizedReduce(cin, cout, BatchNorm, **kw))
        elif diff == -2:
            mod.conv_aspp.append(DoubleFactorizedReduce(cin, cout, BatchNorm, **kw))
        elif diff > 0:
            mod.conv_aspp.append(ReLUConvBN(cin, cout, 1, 1, 0, BatchNorm, affine=True, **kw))


def _aspp_size(size, level_shift):
    return tuple(int((float(s) - 1.0) * (2 ** (-level_shift)) + 1.0) for s in size)


class ADD(nn.Module):
    """ADD.py:118-500 — multi-exit densely connected network."""

    def __init__(self, network_arch, C_index, cell_arch, num_classes, args, low_level_layer):
        super().__init__()
        BatchNorm = nn.BatchNorm2d     # reference SyncBN == F.batch_norm off DataParallel
        F_, B = args.F, args.B
        self.args = args
        self.cells = nn.ModuleList()
        self.cell_arch = torch.from_numpy(cell_arch)
        self._num_classes, self.low_level_layer = num_classes, low_level_layer
        self.decoder = Decoder(num_classes, BatchNorm)
        self.network_arch, self.num_net, self.C_index = network_arch, len(network_arch), C_index
        FB = F_ * B
        _make_stems(self, BatchNorm)
        for i, level in enumerate(network_arch):
            prev, pprev = network_arch[i - 1], network_arch[i - 2]
            down = int(prev - level)
            c_out = F_ * _FM[level]
            if i == 0:
                args_ = (64, 128, 0 - level, False, True)
            elif i == 1:
                args_ = (128, FB * _FM[prev], down, False, True)
            elif i == 2:
                args_ = (FB * _FM[pprev], FB * _FM[prev], down, False, True)
            else:
                dense = [F_ * _FM[l] for l in network_arch[:i - 1]]
                args_ = (dense, FB * _FM[prev], down, True, i < self.num_net - 2)
            ppc, pc, dus, din, dout = args_
            self.cells.append(Cell(BatchNorm, B, ppc, pc, self.cell_arch, level, c_out,
                                   int(dus), dense_in=din, dense_out=dout))
        _kaiming_init(self)
        _make_heads(self, network_arch, C_index, F_, B, num_classes, BatchNorm, low_level_layer)
        _kaiming_init(self)

    # -- trunk shared by forward/get_feature/dynamic_inference (ADD.py:283-308)
    def _trunk(self, x):
        """Generator over cells: yields (i, y, low_level) after each cell, where
        y is the tensor an exit head at cell i would consume."""
        two = list(_run_stems(self, x))
        dense, low = [], None
        cur = None
        for i in range(self.num_net):
            if i < 3:
                two[0], two[1], fm = self.cells[i](two[0], two[1])
                dense.append(fm)
                if i == 2:
                    cur = two[1]
            elif i < self.num_net - 2:
                _, cur, fm = self.cells[i](list(dense[:-1]), cur)
                dense.append(fm)
            elif i == self.num_net - 1:
                cur = self.cells[i](list(dense), cur)
            else:
                cur = self.cells[i](list(dense[:-1]), cur)
            if i == self.low_level_layer:
                low = self.low_level_conv(two[1])
            y = cur if i > 2 else two[1]
            got = yield i, y, low
            if got is not None:          # caller replaced the feature in place (EDM relu, Q3)
                if i > 2:
                    cur = got
                else:
                    two[1] = got

    def _head(self, y, low, size, aspp_size, conv_aspp_iter, level, resize=True, adapt=True):
        if resize and (y.shape[2] < aspp_size[0] or y.shape[3] < aspp_size[1]):
            y = _bilinear(y, aspp_size)
        if adapt and level != self.network_arch[-1]:
            y = self.conv_aspp[conv_aspp_iter](y)
        return self.decoder(self.aspp(y), low, size)

    def forward(self, x):
        """ADD.py:277-325."""
        size = tuple(x.shape[2:])
        aspp_size = _aspp_size(size, self.network_arch[-1] + 2)
        it, out = 0, []
        for i, y, low in self._trunk(x):
            if i in self.C_index or i == self.num_net - 1:
                lvl = self.network_arch[i]
                out.append(self._head(y, low, size, aspp_size, it, lvl))
                if lvl != self.network_arch[-1]:
                    it += 1
        return out

    def get_feature(self, x):
        """ADD.py:327-377 — first exit only; aspp_size uses 2^-last (Q5)."""
        size = tuple(x.shape[2:])
        aspp_size = _aspp_size(size, self.network_arch[-1])
        for i, y, low in self._trunk(x):
            if i in self.C_index:
                return self._head(y, low, size, aspp_size, 0, self.network_arch[i]), y
        return [], []

    def dynamic_inference(self, x, threshold=1.0, confidence='edm', edm=False):
        """ADD.py:379-438 (the 'edm' gate; the 'entropy'/'max' branches of the
        reference are broken — SURVEY Q6 — and are not restated)."""
        if confidence != 'edm':
            raise NotImplementedError("only confidence='edm' is a working reference path")
        tic = time.perf_counter()
        size = tuple(x.shape[2:])
        aspp_size = _aspp_size(size, self.network_arch[-1])
        earlier_exit, it, conf = 0, 0, None
        gen = self._trunk(x)
        send = None
        out = None
        while True:
            try:
                i, y, low = gen.send(send)
            except StopIteration:
                break
            send = None
            if i in self.C_index or i == self.num_net - 1:
                if i != self.num_net - 1:
                    conf = edm(y)
                    y = F.relu(y)       # EDM's in-place ReLU on a view of y (ADD.py:507,516,519)
                    send = y
                    if conf > threshold:
                        it += 1
                        continue
                    out = self._head(y, low, size, aspp_size, it, self.network_arch[i])
                    earlier_exit = 1
                    break
                out = self._head(y, low, size, aspp_size, it, self.network_arch[i],
                                 resize=False, adapt=False)      # ADD.py:433-435
        return out, earlier_exit, time.perf_counter() - tic, conf


class EDM(nn.Module):
    """ADD.py:502-525 — (in-place) ReLU, conv3x3 s2 400->128, ReLU, GAP, MLP 128-64-32-1."""

    def __init__(self):
        super().__init__()
        self.conv = _conv(400, 128, 3, 2, 1)
        self.edm = nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 32),
                                 nn.ReLU(), nn.Linear(32, 1))

    def forward(self, x):
        x = x.squeeze(1)                                  # ADD.py:516
        x = F.relu(self.conv(F.relu(x)))
        return self.edm(x.mean(dim=(2, 3)))


class Cell_baseline(nn.Module):
    """baseline_model.py:14-90 — Cell without dense connections."""

    def __init__(self, BatchNorm, B, prev_prev_C, prev_C, cell_arch, network_arch, C_out,
                 downup_sample):
        super().__init__()
        kw = dict(eps=_EPS, momentum=_MOM)
        self.cell_arch, self.B, self.downup_sample = cell_arch, B, downup_sample
        self.pre_preprocess = ReLUConvBN(prev_prev_C, C_out, 1, 1, 0, BatchNorm, affine=True, **kw)
        if downup_sample == -1:
            self.preprocess = FactorizedReduce(prev_C, C_out, BatchNorm, **kw)
        else:
            self.preprocess = ReLUConvBN(prev_C, C_out, 1, 1, 0, BatchNorm, affine=True, **kw)
        self._ops = nn.ModuleList(
            OPS[PRIMITIVES[int(r[1])]](C_out, 1, BatchNorm, affine=True, **kw) for r in cell_arch)

    def forward(self, prev_prev_input, prev_input):
        s1 = prev_input
        if self.downup_sample == 1:
            s1 = _bilinear(s1, (_scale_dim(s1.shape[2], 2), _scale_dim(s1.shape[3], 2)))
        s1 = self.preprocess(s1)
        s0 = prev_prev_input
        if s0.shape[2] != s1.shape[2]:
            s0 = _bilinear(s0, s1.shape[2:])
        s0 = self.pre_preprocess(s0)
        return prev_input, _run_blocks(self._ops, self.cell_arch, self.B, [s0, s1])


class Baselin_Model(nn.Module):
    """baseline_model.py:93-265 (class name spelled as in the reference)."""

    def __init__(self, network_arch, C_index, cell_arch, num_classes, args, low_level_layer):
        super().__init__()
        BatchNorm = nn.BatchNorm2d
        F_, B = args.F, args.B
        self.args = args
        self.cells = nn.ModuleList()
        self.cell_arch = torch.from_numpy(cell_arch)
        self._num_classes, self.low_level_layer = num_classes, low_level_layer
        self.decoder = Decoder(num_classes, BatchNorm)
        self.network_arch, self.num_net, self.C_index = network_arch, len(network_arch), C_index
        FB = F_ * B
        _make_stems(self, BatchNorm)
        for i, level in enumerate(network_arch):
            prev, pprev = network_arch[i - 1], network_arch[i - 2]
            down = int(prev - level)
            if i == 0:
                ppc, pc, down = 64, 128, 0 - level
            elif i == 1:
                ppc, pc = 128, FB * _FM[prev]
            else:
                ppc, pc = FB * _FM[pprev], FB * _FM[prev]
            self.cells.append(Cell_baseline(BatchNorm, B, ppc, pc, self.cell_arch, level,
                                            F_ * _FM[level], int(down)))
        _kaiming_init(self)
        _make_heads(self, network_arch, C_index, F_, B, num_classes, BatchNorm, low_level_layer)
        _kaiming_init(self)

    def forward(self, x):
        """baseline_model.py:224-254."""
        size = tuple(x.shape[2:])
        aspp_size = _aspp_size(size, self.network_arch[-1] + 2)
        two = list(_run_stems(self, x))
        it, out, low = 0, [], None
        for i in range(self.num_net):
            two = list(self.cells[i](two[0], two[1]))
            if i == self.low_level_layer:
                low = self.low_level_conv(two[1])
            if i in self.C_index or i == self.num_net - 1:
                y = two[1]
                if y.shape[2] < aspp_size[0] or y.shape[3] < aspp_size[1]:
                    y = _bilinear(y, aspp_size)
                if self.network_arch[i] != self.network_arch[-1]:
                    y = self.conv_aspp[it](y)
                    it += 1
                out.append(self.decoder(self.aspp(y), low, size))
        return out


def normalized_shannon_entropy(x, num_class=19):
    """operations.py:161-170 — mean over pixels (and SUM over batch) of
    -sum_c p log p / log(num_class)."""
    h, w = x.shape[2], x.shape[3]
    ent = -(F.softmax(x, dim=1) * F.log_softmax(x, dim=1)).sum(dim=1) / math.log(num_class)
    return (ent.sum() / (h * w)).item()


def confidence_max(x, thresold, num_class=19):
    """operations.py:172-180."""
    p = F.softmax(x, dim=1).max(dim=1)[0]
    return int((p > thresold).sum()) / (x.shape[2] * x.shape[3])


def global_batch_norm(shards, running_mean, running_var, weight, bias,
                      momentum=_MOM, eps=_EPS):
    """SyncBN parity definition: training-mode F.batch_norm over
    the concatenation of all ranks' shards; returns per-shard outputs."""
    full = torch.cat(shards, dim=0)
    y = F.batch_norm(full, running_mean, running_var, weight, bias, True, momentum, eps)
    return list(torch.split(y, [s.shape[0] for s in shards], dim=0))


def cross_entropy_mean_exits(outputs, target, weight=None, ignore_index=255):
    """train.py:70,229-233 — CrossEntropyLoss(weight, ignore_index=255) per exit,
    averaged over exits."""
    losses = [F.cross_entropy(o, target, weight=weight, ignore_index=ignore_index)
              for o in outputs]
    return sum(losses) / len(losses)


class Evaluator:
    """utils/metrics.py:4-52 — confusion matrix via bincount, mIoU = nanmean."""

    def __init__(self, num_class):
        self.num_class = num_class
        self.confusion_matrix = torch.zeros((num_class, num_class))

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        mask = (gt_image >= 0) & (gt_image < self.num_class)
        label = self.num_class * gt_image[mask].int() + pre_image[mask]
        count = torch.bincount(label, minlength=self.num_class ** 2)
        self.confusion_matrix += count.reshape(self.num_class, self.num_class)

    def reset(self):
        self.confusion_matrix = torch.zeros((self.num_class, self.num_class))

    @staticmethod
    def _nanmean(x):
        ok = ~torch.isnan(x)
        return torch.where(ok, x, torch.zeros_like(x)).sum() / ok.sum()

    def Pixel_Accuracy(self):
        return torch.diag(self.confusion_matrix).sum() / self.confusion_matrix.sum()

    def Pixel_Accuracy_Class(self):
        return self._nanmean(torch.diag(self.confusion_matrix) / self.confusion_matrix.sum(dim=1))

    def Mean_Intersection_over_Union(self):
        cm = self.confusion_matrix
        d = torch.diag(cm)
        return self._nanmean(d / (cm.sum(dim=1) + cm.sum(dim=0) - d)).item()

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        d = torch.diag(cm)
        freq = cm.sum(dim=1) / cm.sum()
        iu = d / (cm.sum(dim=1) + cm.sum(dim=0) - d)
        return (freq[freq > 0] * iu[freq > 0]).sum()


def class_weights_from_labels(label_batches, num_classes):
    """utils/calculate_weights.py:6-29 restated with numpy: labels outside [0, num_classes) are dropped, the class
    histogram z is accumulated over the batches, weight_c = 1 / ln(1.02 + z_c / sum(z))."""
    import numpy as np
    z = np.zeros((num_classes,))
    for y in label_batches:
        y = np.asarray(y)
        mask = (y >= 0) & (y < num_classes)
        z += np.bincount(y[mask].astype(np.uint8), minlength=num_classes)
    return np.array([1 / np.log(1.02 + f / z.sum()) for f in z])
