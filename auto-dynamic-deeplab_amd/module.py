"""nn.Module front end of the plan runtime: any addk module's forward(x) builds (once per input
signature) and replays a static plan of HIP kernel launches; autograd sees one Function."""
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import plan as _plan
from .plan import Act, Graph, NbtCounter, OutRef

_world = None          # set by addk.parallel.init_sync_bn()


def set_world(w):
    global _world
    _world = w


def conv2d(ci, co, k, stride=1, padding=0, dilation=1, groups=1, bias=False):
    """nn.Conv2d used purely as a parameter holder (same state_dict keys as the reference).  Dense
    k>1 weights are kept in channels_last memory ([O][KH][KW][I]) — the layout the kernels read."""
    m = nn.Conv2d(ci, co, k, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias)
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


def ensure_layout(p):
    if p.dim() == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
        p.data = p.data.contiguous(memory_format=torch.channels_last)
    elif p.dim() != 4 and not p.data.is_contiguous():
        p.data = p.data.contiguous()


def _flatten(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
        return ('t',)
    if isinstance(x, (list, tuple)):
        return ('l', [_flatten(v, out) for v in x])
    raise TypeError('addk modules take tensors or (nested) lists of tensors, got %r' % type(x))


def _unflatten(spec, it):
    if spec[0] == 't':
        return next(it)
    return [_unflatten(s, it) for s in spec[1]]


class Plan:
    def __init__(self, module, emit, inputs, spec, training, want_grad):
        dev = inputs[0].device
        for p in module.parameters():
            ensure_layout(p)
        self.g = g = Graph(dev, training, want_grad, _world)
        self.inrefs, acts = [], []
        for x in inputs:
            a, r = g.input_nchw(x, requires_grad=x.requires_grad)
            acts.append(a)
            self.inrefs.append(r)
        self.in_tensors = None
        res = emit(g, *_unflatten(spec, iter(acts))) if spec[0] == 'l' else emit(g, acts[0])
        self.single = not isinstance(res, (list, tuple))
        res = [res] if self.single else list(res)
        self.outs = []
        for r in res:
            if isinstance(r, Act):
                idx = [i for i, a in enumerate(acts) if a is r]
                self.outs.append(('in', idx[0]) if idx else ('out', g.output_nchw(r)))
            elif isinstance(r, OutRef):
                self.outs.append(('out', r))
            else:
                raise TypeError('emit returned %r' % type(r))
        # whole lists are always replayed front to back here: level-ordered, batched and on two streams like the fused train step
        g.reorder = True
        import os
        g.finalize(int(os.environ.get('ADDK_STREAMS', '2')))
        self.params = list(g.params)
        self.param_ptrs = [p.data_ptr() for p in self.params]
        self.serial = 0
        self.nbt = NbtCounter(g.nbt)
        # inference plans (no autograd) are replayed as one hipGraph launch from the 3rd call on: at ~900 launches per
        # forward the Python/ctypes launch loop (~15 us per command) costs more than the kernels themselves
        self.graph = self.bgraph = None
        self.calls = 0
        self.static_in = self.static_gy = None

    def check_params(self):
        for p, ptr in zip(self.params, self.param_ptrs):
            if p.data_ptr() != ptr:
                return False
        return True

    _COLLECTIVES = ('allreduce', 'allreduce_packed', 'grad_allreduce')

    def _graph_allowed(self, inputs):
        """Inference plans always (ADDK_GRAPH_INFER=0 disables); training / autograd plans — the reference's own call pattern
        model(x); loss.backward() — as TWO hipGraphs, forward list and backward list (ADDK_GRAPH_MODULE=0 disables), unless the
        lists hold RCCL calls (SyncBN at world > 1: those steps belong to train.TrainStep, which decides capture collectively)."""
        if not inputs[0].is_cuda:
            return False
        if not (self.g.want_grad or self.g.training):
            return os.environ.get('ADDK_GRAPH_INFER', '1') == '1'
        if os.environ.get('ADDK_GRAPH_MODULE', '1') != '1':
            return False
        if getattr(self, '_has_coll', None) is None:
            self._has_coll = any(c.name in self._COLLECTIVES for c in list(self.g.fwd) + list(self.g.bwd))
        return not self._has_coll

    def _forward_graphed(self, inputs):
        if getattr(self, '_graph_refused', False) or not self._graph_allowed(inputs):
            return False
        self.calls += 1
        if self.calls < 3:
            return False
        if self.graph is None:
            self.static_in = [torch.empty_like(x, memory_format=torch.contiguous_format) for x in inputs]
            for r, xs, x in zip(self.inrefs, self.static_in, inputs):
                xs.copy_(x)
                r.bind(xs)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    self.g.run_parallel(self.g.fwd, None)
            except Exception as e:      # a refused capture costs nothing but this call's graph: the eager launch loop stays the path of this plan
                import warnings
                warnings.warn('addk: hipGraph capture of the forward list refused (%s); this plan keeps the eager launch loop' % e)
                self._graph_refused = True
                return False
            self.graph = graph
        for xs, x in zip(self.static_in, inputs):
            xs.copy_(x)
        self.graph.replay()
        return True

    def _backward_graphed(self, gouts):
        """Replay of the backward list as one hipGraph: the incoming gradients are copied into plan-owned buffers (fixed addresses).
        Captured on the first backward that follows a graphed forward (the lists have run eagerly at least twice by then)."""
        if self.graph is None or not self.g.want_grad:
            return False
        if self.static_gy is None:
            self.static_gy = []
            for kind, o in self.outs:
                if kind == 'in':
                    self.static_gy.append(None)
                    continue
                buf = torch.zeros_like(o.y, memory_format=torch.contiguous_format)
                o.set_grad(buf)
                self.static_gy.append(buf)
        for (kind, o), gy, buf in zip(self.outs, gouts, self.static_gy):
            if buf is None:
                continue
            if gy is None:
                buf.zero_()
            else:
                buf.copy_(gy)
        if getattr(self, '_bgraph_refused', False):       # capture was refused once: the list runs eagerly on the plan-owned gradient buffers
            self.g.run_parallel(self.g.bwd, None)
            return True
        if self.bgraph is None:
            torch.cuda.synchronize()
            bgraph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(bgraph, capture_error_mode='thread_local'):
                    self.g.run_parallel(self.g.bwd, None)
            except Exception as e:      # the incoming gradients already sit in the plan-owned buffers the commands point at: run the list eagerly
                import warnings
                warnings.warn('addk: hipGraph capture of the backward list refused (%s); this plan keeps the eager launch loop' % e)
                self._bgraph_refused = True
                self.g.run_parallel(self.g.bwd, None)
                return True
            self.bgraph = bgraph
        self.bgraph.replay()
        return True

    def forward(self, inputs):
        self.in_tensors = inputs
        if not self._forward_graphed(inputs):
            for r, x in zip(self.inrefs, inputs):
                r.bind(x)
            self.g.run_parallel(self.g.fwd, None)
        self.nbt.bump()
        self.serial += 1
        # outputs are COPIES of the plan-owned buffers: a caller written for the reference may keep the result of one forward()
        # across the next one (ADDK_OUTPUT_VIEWS=1 hands out views and saves the 0.1 ms copy per full-resolution logits tensor)
        views = os.environ.get('ADDK_OUTPUT_VIEWS', '0') == '1'
        outs = []
        for kind, o in self.outs:
            outs.append(inputs[o] if kind == 'in' else (o.y.view_as(o.y) if views else o.y.clone()))
        return outs

    def backward(self, gouts):
        if not self._backward_graphed(gouts):
            hold = []
            for (kind, o), gy in zip(self.outs, gouts):
                if kind == 'in':
                    continue
                if gy is None:
                    gy = torch.zeros_like(o.y)
                hold.append(o.set_grad(gy))
            self.g.run_parallel(self.g.bwd, None)
        gin = [r.grad for r in self.inrefs]
        # passthrough outputs route their gradient straight back to the input
        for (kind, o), gy in zip(self.outs, gouts):
            if kind == 'in' and gy is not None and self.in_tensors[o].requires_grad:
                gin[o] = gy if gin[o] is None else gin[o] + gy
        gp = [self.g.pgrad.get(p) if p in self.g.pginit else None for p in self.params]
        return gin, gp


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, n_in, *tensors):
        ctx.plan, ctx.n_in = plan, n_in
        outs = plan.forward(list(tensors[:n_in]))
        ctx.serial = plan.serial
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if ctx.serial != plan.serial:
            raise RuntimeError('addk: backward() after a newer forward() of the same plan — activations were overwritten')
        gin, gp = plan.backward(gouts)
        gin = [g if (g is not None and x.requires_grad) else None for g, x in zip(gin, plan.in_tensors)]
        return (None, None) + tuple(gin) + tuple(gp)


class AddkModule(nn.Module):
    """Base of every module on the path.  Subclasses implement emit(g, *acts)."""

    def _plans(self):
        d = self.__dict__.get('_addk_plans')
        if d is None:
            d = self.__dict__['_addk_plans'] = {}
        return d

    def run_plan(self, emit, inputs, tag=''):
        flat = []
        spec = _flatten(list(inputs) if len(inputs) != 1 else inputs[0], flat)
        if not flat:
            raise TypeError('addk module called without tensors')
        for x in flat:
            _plan.require_device(x)
        L.load()
        flat = [x if x.dtype == torch.float32 else x.float() for x in flat]
        want_grad = torch.is_grad_enabled() and (any(x.requires_grad for x in flat) or
                                                 any(p.requires_grad for p in self.parameters()))
        key = (tag, tuple(tuple(x.shape) for x in flat), tuple(bool(x.requires_grad) for x in flat), self.training,
               want_grad, id(_world), int(L.load().addk_get_conv_precision()))     # a plan belongs to the arithmetic mode it was built in
        plans = self._plans()
        plan = plans.get(key)
        if plan is not None and not plan.check_params():
            plan = None          # parameters were re-allocated (.cuda()/.to()): rebuild
        if plan is None:
            plan = plans[key] = Plan(self, emit, flat, spec, self.training, want_grad)
        if want_grad:
            outs = _PlanFn.apply(plan, len(flat), *flat, *plan.params)
        else:
            with torch.no_grad():
                outs = plan.forward(flat)
        outs = list(outs)
        return outs[0] if plan.single else outs

    def forward(self, *inputs):
        return self.run_plan(self.emit, inputs)

    def emit(self, g, *acts):
        raise NotImplementedError

    def train(self, mode=True):
        return super().train(mode)
