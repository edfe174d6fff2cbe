"""EDM-gated dynamic inference (reference ADD.dynamic_inference, ADD.py:379-438; eval.py:195-230).

The gate stays on the host and only selects which exit's kernels fire (north_star): the trunk is emitted
once into a single launch list that is cut into segments — [stems+cells up to gate k + EDM] and, per gate,
[early head k]; the last segment is [remaining cells + final head].  All segments share one buffer set."""
import torch

from .module import ensure_layout
from . import plan as _plan
from .plan import Act, Graph


class DynamicPlan:
    def __init__(self, model, edm, x):
        from .modeling.ADD import _aspp_size
        for p in list(model.parameters()) + list(edm.parameters()):
            ensure_layout(p)
        if x.shape[0] != 1:
            # ADD.py:421 `if confidence_value > threshold` on a [bs, 1] tensor raises for bs > 1 ("Boolean value of Tensor with more
            # than one value is ambiguous"): the gate is a per-image decision (eval.py:195-230 runs bs = 1); same error class here
            raise RuntimeError('dynamic_inference gates one image at a time (got batch size %d): the reference\'s '
                               '`if confidence_value > threshold` is ambiguous for more than one value' % x.shape[0])
        self.g = g = Graph(x.device, False, False, None)
        # the gate scalar travels through pinned host memory: the fused EDM head (csrc/edm.hip) writes it there itself; on the generic path an
        # asynchronous 4-byte copy does.  Either way the host waits on ONE event, not on the whole device (ADD.py:421 does
        # `if confidence_value > threshold`, a blocking read)
        self._conf_host = torch.zeros(1, dtype=torch.float32)
        self._conf_evt = None
        if x.is_cuda:                                   # (dry-run planning on CPU in tests/test_plan_dryrun.py builds plans without a device)
            self._conf_host = self._conf_host.pin_memory()
            self._conf_evt = torch.cuda.Event()
        self.conf_fused = []
        a, self.inref = g.input_nchw(x)
        size = (a.H, a.W)
        aspp_size = _aspp_size(size, model.network_arch[-1])          # 2^-last
        self.trunk_end, self.head_rng, self.conf, self.heads = [], [], [], []
        gen = model._trunk(g, a)
        send, it = None, 0
        self.final = None
        while True:
            try:
                i, y, low = gen.send(send)
            except StopIteration:
                break
            send = None
            if i in model.C_index or i == model.num_net - 1:
                if i != model.num_net - 1:
                    g.edm_fused = False
                    ca = edm.emit(g, y, host_out=self._conf_host if x.is_cuda else None)
                    self.conf_fused.append(bool(g.edm_fused))
                    if g.edm_fused:                                   # the head wrote [N,1,1,1] itself (and the pinned word): no layout launch
                        from .plan import OutRef
                        conf = OutRef(ca.raw.view().permute(0, 3, 1, 2))
                    else:
                        conf = g.output_nchw(ca)                      # EDM applies ReLU to y in place (Q3) ...
                    y = Act(y.raw, y.bn, True, False, rs=y.rs)              # ... so everything downstream sees relu(y)
                    send = y
                    self.trunk_end.append(len(g.fwd))
                    self.conf.append(conf)
                    h0 = len(g.fwd)
                    self.heads.append(model._head(g, y, low, size, aspp_size, it, model.network_arch[i]))
                    self.head_rng.append((h0, len(g.fwd)))
                    it += 1                                           # ADD.py:422 increments on every passed gate
                else:
                    self.final = model._head(g, y, low, size, aspp_size, it, model.network_arch[i], resize=False, adapt=False)
        g.finalize()
        self.params = list(g.params)
        self.ptrs = [p.data_ptr() for p in self.params]
        self.x_static = torch.empty_like(x, dtype=torch.float32, memory_format=torch.contiguous_format)
        self.inref.bind(self.x_static)
        self.graphs = {}          # (begin, end) -> hipGraph of that launch-list segment
        self.segs = {}            # (begin, end) -> that segment's launch list, level-ordered / batched / scheduled on two streams
        import os
        g.nstreams = int(os.environ.get('ADDK_STREAMS', '2'))
        self.calls = 0

    def check_params(self):
        return all(p.data_ptr() == q for p, q in zip(self.params, self.ptrs))

    def _seg(self, i0, i1):
        """Run launch-list segment [i0, i1): eagerly the first two calls, then as a captured hipGraph (each segment —
        trunk up to a gate, an exit head, the remainder — is its own graph; the host gate picks which ones replay)."""
        import os
        g = self.g
        if i1 < 0:
            i1 = len(g.fwd)
        cmds = self.segs.get((i0, i1))
        if cmds is None:          # every command belongs to exactly one segment: re-order and schedule the slice on its own
            cmds = list(g.fwd[i0:i1])
            if os.environ.get('ADDK_LEVEL_BATCH', '1') == '1':
                g._level_batch(cmds)
            _plan.schedule(cmds, g.nstreams)
            for c in cmds:
                if c.event:
                    c.event = torch.cuda.Event()
            self.segs[(i0, i1)] = cmds
        if self.calls < 3 or os.environ.get('ADDK_GRAPH_INFER', '1') != '1':
            g.run_parallel(cmds, None)
            return
        gr = self.graphs.get((i0, i1))
        if gr is None:
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                g.run_parallel(cmds, None)
            self.graphs[(i0, i1)] = gr
        gr.replay()

    def run(self, x, threshold):
        self.calls += 1
        self.x_static.copy_(x)
        pos, conf = 0, None
        with torch.no_grad():
            for k, end in enumerate(self.trunk_end):
                self._seg(pos, end)
                conf = self.conf[k].y.reshape(x.shape[0], -1)
                assert conf.numel() == 1
                h0, h1 = self.head_rng[k]
                if not self.conf_fused[k]:
                    self._conf_host[:1].copy_(conf.reshape(-1), non_blocking=True)
                if self._conf_evt is not None:
                    self._conf_evt.record()
                    self._conf_evt.synchronize()                      # the host waits for this one scalar only
                if bool(self._conf_host[0] > threshold):              # the gate (ADD.py:421); bs = 1 as in eval.py:195-230
                    pos = h1
                    continue
                self._seg(h0, h1)
                return self.heads[k].y, 1, conf
            self._seg(pos, -1)
        return self.final.y, 0, conf
