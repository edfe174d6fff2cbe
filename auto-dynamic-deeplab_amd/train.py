"""Fused training step of the path (reference train.py:216-242): forward of all exits, CrossEntropyLoss per exit
averaged over exits, backward, SGD(momentum, weight_decay, nesterov) — emitted as ONE static plan over flat
parameter/gradient buffers and replayed as a single hipGraph launch (eager launch list when RCCL collectives
sit inside the step)."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import module as _module
from . import plan as _plan
from .module import ensure_layout
from .plan import Graph, NbtCounter


def _flat_views(params, device):
    """Re-point every parameter into one flat fp32 buffer (16-byte aligned slots, strides preserved) and build
    matching views into a flat gradient buffer."""
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    flat_p = torch.zeros(total, dtype=torch.float32, device=device)
    flat_g = torch.zeros(total, dtype=torch.float32, device=device)
    gviews = {}
    for p, o in zip(params, offs):
        pv = torch.as_strided(flat_p, p.shape, p.stride(), o)
        with torch.no_grad():
            pv.copy_(p.data)
        p.data = pv
        gviews[p] = torch.as_strided(flat_g, p.shape, p.stride(), o)
    return flat_p, flat_g, gviews


class TrainStep:
    def __init__(self, model, batch_shape, lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, class_weight=None,
                 ignore_index=255, use_graph=None, sync_comm=None, nstreams=None):
        lib = self.lib = L.load()
        dev = next(model.parameters()).device
        _plan.require_device(next(model.parameters()))
        self.model, self.dev = model, dev
        model.train()
        for p in model.parameters():
            ensure_layout(p)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat_p, self.flat_g, gviews = _flat_views(self.params, dev)
        self.mom_buf = torch.zeros_like(self.flat_p)
        self.lr_dev = torch.tensor([lr], dtype=torch.float32, device=dev)
        self.hyper = (momentum, weight_decay, int(nesterov))
        N, Cin, H, W = batch_shape
        self.x = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.target = torch.zeros((N, H, W), dtype=torch.int64, device=dev)
        # data-parallel world: from torch.distributed itself, NOT from the SyncBN communicator — the reference's DDP
        # averages gradients with or without --sync-bn (train.py:173-175)
        import torch.distributed as dist
        self.group = sync_comm.group if sync_comm is not None else None
        self.world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        self.comm = sync_comm
        g = self.g = Graph(dev, True, True, sync_comm)
        g.pgrad_views = gviews
        # the step never hands out full-resolution logits: up-sampling + loss + their backward run as one launch per exit
        g.fuse_ce = os.environ.get('ADDK_FUSE_CE', '1') == '1'
        a, self.inref = g.input_nchw(self.x)
        self.inref.bind(self.x)
        outs = model.emit(g, a)
        self.outs = outs
        nex = len(outs)
        ncls = outs[0].shape[1] if outs[0].fused_ce else outs[0].y.shape[1]
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.wsum = torch.zeros(1, dtype=torch.float32, device=dev)
        ws = torch.zeros(int(lib.addk_ce_ws_floats(N, H * W)), dtype=torch.float32, device=dev)
        cw = class_weight.to(dev).float().contiguous() if class_weight is not None else None
        self._keep = [ws, cw]
        cwp = cw.data_ptr() if cw is not None else None
        g._add(g.fwd, 'loss_zero', lib.addk_fill, self.loss.data_ptr(), 1, 0.0, wr=[self.loss])
        g._add(g.fwd, 'ce_count', lib.addk_ce_count, self.target.data_ptr(), N * H * W, cwp, ignore_index, ncls,
               self.wsum.data_ptr(), ws.data_ptr(), rd=[self.target], wr=[self.wsum, ws])
        self.dlogits = []
        for o in outs:
            if o.fused_ce:
                o.ce = dict(target=self.target, class_w=cwp, ignore_index=ignore_index, wsum=self.wsum, scale=1.0 / nex, loss=self.loss)
                self.dlogits.append(None)
                continue
            d = torch.empty_like(o.y)
            self.dlogits.append(d)
            g._add(g.fwd, 'ce_fwd_bwd', lib.addk_ce_fwd_bwd, o.y.data_ptr(), self.target.data_ptr(), N, ncls, H * W, cwp,
                   ignore_index, self.wsum.data_ptr(), 1.0 / nex, self.loss.data_ptr(), d.data_ptr(), ws.data_ptr(),
                   rd=[o.y, self.target, self.wsum], wr=[self.loss, d, ws])
            o.dy_ptr, o.dynamic = d.data_ptr(), False
        has_coll = self.world > 1 or (sync_comm is not None and sync_comm.force)
        self.gsync = None
        if has_coll:
            from .parallel import GradSync
            self.gsync = g.grad_sync = GradSync(self.flat_g, gviews, self.group, int(os.environ.get('ADDK_GRAD_BUCKETS', '4')),
                                                log=getattr(sync_comm, 'log', None))
        if nstreams is None:
            # two HIP streams: independent branches of the cell DAG overlap (-4 ms of 82 at config 2, eager or captured);
            # 3, 4 and 6 streams measure the same or slightly worse (68.5 / 68.7 / 69.8 / 69.7 ms).  The SyncBN path
            # keeps its collectives on one stream.
            nstreams = int(os.environ.get('ADDK_STREAMS', '2'))
        g.finalize(nstreams)
        self.nbt = NbtCounter(g.nbt)
        self.nbt.bump(); self.nbt.flat.sub_(self.nbt.inc)      # flatten now (pointers must be fixed before graph capture)
        self.n_active = self.flat_p.numel()
        self.nbytes = g.nbytes
        if use_graph is None:
            # A step WITHOUT collectives is captured whole.  A step with RCCL collectives: whole when the world is this one
            # process (the forced-exchange rehearsal, tests/test_gpu_train.py) or ADDK_GRAPH_DDP=1 asks for it; with world > 1 the
            # default is the eager launch list until a multi-rank capture has been seen on hardware (no multi-GPU box was available
            # to the builder: DESIGN.md §7; bench.py times the eager list first and then TRIES the capture under a watchdog).
            # (Round 3's third form — hipGraph segments between eager collectives — measured slower than the eager list twice
            # and was removed in round 4.)
            ddp = os.environ.get('ADDK_GRAPH_DDP')
            if ddp is None:
                ddp = '1' if self.world == 1 else '0'
            graphs_on = os.environ.get('ADDK_GRAPH', '1') == '1'
            use_graph = graphs_on and (not has_coll or ddp == '1')
        self.has_coll = has_coll
        self.graph = None
        self.use_graph = use_graph
        self.steps = 0

    # one eager pass of the whole step on the current stream
    def _run(self):
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        g = self.g
        g.run_parallel(g.fwd, main)
        g.run_parallel(g.bwd, main)          # the bucketed gradient all-reduces are commands of this list (parallel.GradSync)
        if self.gsync is not None:
            self.gsync.wait()
        self._sgd(st)
        self.nbt.flat.add_(self.nbt.inc)

    def _sgd(self, st):
        mom, wd, nest = self.hyper
        L.check(self.lib.addk_sgd_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.mom_buf.data_ptr(), self.n_active,
                                       self.lr_dev.data_ptr(), mom, wd, nest, 0, 1.0 / self.world, st), 'sgd_step')

    def forward_backward_only(self):
        """Forward + loss + backward without the optimizer update (parity tests)."""
        main = torch.cuda.current_stream()
        self.g.run_parallel(self.g.fwd, main)
        self.g.run_parallel(self.g.bwd, main)
        if self.gsync is not None:
            self.gsync.wait()

    def _capture(self):
        """Capture one step into a hipGraph.  Returns the graph, or None when the step has collectives and ANY rank's runtime
        refused the capture (then every rank stays on the eager launch list: the decision is made collectively, so no rank
        replays a graph while another issues eager collectives).  Nothing is executed here: the caller's eager step stands."""
        graph, err = torch.cuda.CUDAGraph(), None
        try:
            # A step with RCCL exchanges is captured in THREAD-LOCAL error mode: ProcessGroupNCCL's watchdog thread keeps
            # querying the events of the eager step's collectives, and in the default global mode such a query from another
            # thread while this one captures is an illegal call.
            with torch.cuda.graph(graph, capture_error_mode='thread_local' if self.has_coll else 'global'):
                self._run()
        except Exception as e:
            if not self.has_coll:
                raise
            err = e
        if self.has_coll:
            ok = torch.tensor([0.0 if err is not None else 1.0], device=self.dev)
            import torch.distributed as dist
            if self.world > 1 and dist.is_available() and dist.is_initialized():
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok.item()) < 1.0:
                import sys
                why = str(err).splitlines()[0] if err is not None else 'another rank failed'
                sys.stderr.write('[addk] hipGraph capture of the data-parallel step failed (%s): eager replay on every rank\n' % why)
                if self.gsync is not None:
                    del self.gsync.works[:]
                torch.cuda.synchronize()
                return None
        return graph

    def enable_capture(self):
        """Switch a step that runs on the eager list to whole-step capture; the next step() runs eagerly once more and captures
        (a refused capture keeps every rank on the eager list).  Used by bench.py at world > 1 after the eager timing."""
        self.use_graph, self.graph = True, None

    def load_batch(self, images, targets):
        self.x.copy_(images, non_blocking=True)
        self.target.copy_(targets, non_blocking=True)

    def set_lr(self, lr):
        self.lr_dev.fill_(float(lr))

    def step(self, images=None, targets=None, lr=None):
        """Runs one optimisation step on the resident batch (or on `images`/`targets` if given) and returns the
        loss as a device tensor (no host sync; the reference's per-step loss.item() is the caller's choice)."""
        if images is not None:
            self.load_batch(images, targets)
        if lr is not None:
            self.set_lr(lr)
        if self.use_graph:
            if self.graph is None:
                # this call's step runs eagerly (it also warms RCCL); the capture that follows executes nothing
                self._run()
                torch.cuda.synchronize()
                if self.has_coll:
                    # ProcessGroupNCCL's watchdog thread polls the events of the eager step's collectives (every 100 ms) until it has seen them
                    # complete; a query from that thread while this one captures is the one cross-thread HIP call left in the process.  Give
                    # it time to retire them first.  Inference, not a shown cause: an abort inside the capture of a step with collectives was
                    # seen in rounds 2 and 5 (once each, not reproducible) — tests/conftest.py now leaves the native stack
                    import time
                    time.sleep(0.3)
                self.graph = self._capture()
                if self.graph is None:
                    self.use_graph = False
            else:
                self.graph.replay()
        else:
            self._run()
        self.steps += 1
        return self.loss

    def close(self):
        """Release everything that refers to the process group BEFORE `dist.destroy_process_group()`: the captured hipGraph (its
        nodes are RCCL kernels of that communicator), the outstanding Work handles of the gradient buckets and the side streams'
        pending launches.  Callers that own a process group call this, then `torch.cuda.synchronize()`, then destroy the group
        (train.py:49-53 is the lifecycle the drop-in has to survive: a communicator torn down under a live graph or a polling
        watchdog was the one ordering fault found in bench.py's and the tests' teardown paths)."""
        if self.gsync is not None:
            try:
                self.gsync.wait()
            except Exception:
                del self.gsync.works[:]
        if self.dev.type == 'cuda':
            torch.cuda.synchronize()
        self.graph = None
        self.use_graph = False

    def grads(self):
        return {p: self.g.pgrad.get(p) for p in self.params}
