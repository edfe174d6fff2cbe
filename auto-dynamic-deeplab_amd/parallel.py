"""Data-parallel pieces (one process per GPU, torch.distributed backend "nccl" == RCCL over xGMI):

 * SyncBNComm — the cross-rank reduction of BatchNorm statistics: per BN call the fp64 (sum, sumsq) vector (2C doubles)
   in forward and (dmean, dvar) (2C floats) in backward; replaces the reference's master/slave queue protocol
   (modeling/sync_batchnorm/comm.py:56-129, batchnorm.py:95-108).  The vectors of all BatchNorms of one dependency level
   of the launch list live back to back in one arena (plan.Graph._bind_late), so a level is ONE plain all_reduce on a
   contiguous tensor: 322 collectives per training step at config 2 instead of 624 (the floor is the network's depth in
   BatchNorms, ~150 per direction: an exchange on the critical path waits for the one before it).
 * GradSync — gradient averaging across ranks (the reference's DistributedDataParallel, train.py:173-175): the step's
   FLAT gradient buffer is cut into a few contiguous buckets; each bucket's all-reduce is issued asynchronously right
   after the last backward launch that writes into it, so it runs on RCCL's stream underneath the rest of the backward
   pass; the SGD kernel waits for all of them.  Scaling by 1/world happens inside the SGD kernel.

gloo works for CPU tests of the host logic (tests/test_parallel_gloo.py)."""
import torch
import torch.distributed as dist

from . import module as _module


class SyncBNComm:
    def __init__(self, group=None, force=False):
        assert dist.is_initialized(), 'init_process_group first'
        self.group = group
        self.force = force          # run the exchange even at world_size 1 (exercises the N>1 kernels on one GPU)
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.calls = 0
        self.log = None             # tests: list collecting (numel, dtype) of every collective, in issue order

    def _allreduce(self, t, stream):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)      # enqueued on torch's current stream
        self.calls += 1
        if self.log is not None:
            self.log.append(('stats', t.numel(), str(t.dtype)))
        return 0

    def _allreduce_vec(self, vec, stream):
        return self._allreduce(vec.view(), stream)

    def emit_allreduce(self, g, lst, vec):
        """Append the all-reduce of one statistics vector (plan.LateVec) to command list `lst`.  When the list is level-
        ordered, the exchanges of one level are merged into one all-reduce of their shared arena (plan.Graph._level_batch)."""
        c = g._add(lst, 'allreduce', self._allreduce_vec, vec, rd=[vec], wr=[vec], pin=True)     # RCCL call: main stream only
        c.payload = vec
        return c


class GradSync:
    """Bucketed, overlapped gradient all-reduce of a fused train step (train.TrainStep).  `insert` is called by
    plan.Graph.finalize once the backward list has its final order: for every parameter it finds the last launch that
    writes its gradient, cuts the flat gradient buffer into ~`nbuckets` contiguous buckets of similar size (address order)
    and places one asynchronous all-reduce per bucket directly behind the bucket's last producer.  `wait()` joins them in
    front of the optimizer.  Sum only: the 1/world factor is applied by the SGD kernel."""

    def __init__(self, flat_g, views, group=None, nbuckets=4, log=None):
        self.flat_g, self.views, self.group = flat_g, views, group
        self.nbuckets = max(1, int(nbuckets))
        self.works, self.buckets, self.log = [], [], log

    def _issue(self, t, stream):
        self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.log is not None:
            self.log.append(('grad', t.numel(), str(t.dtype)))
        return 0

    def wait(self):
        for w in self.works:
            w.wait()              # NCCL/RCCL: the current stream waits for the collective's stream (no host block)
        del self.works[:]

    def insert(self, g, bwd):
        from .plan import Cmd, _overlap, _region
        esz = self.flat_g.element_size()
        base = self.flat_g.untyped_storage().data_ptr()
        spans = []                                    # (lo, hi) float offsets of every parameter's gradient, address order
        for p, v in self.views.items():
            lo = v.storage_offset()
            spans.append((lo, lo + (v.numel() + 3) // 4 * 4))
        spans.sort()
        total = self.flat_g.numel()
        # last writer of every 16-byte-aligned span
        last = {}
        for i, c in enumerate(bwd):
            for r in getattr(c, 'wr', ()):
                if r[0] == base:
                    last.setdefault((r[1], r[2]), i)
                    last[(r[1], r[2])] = i
        def ready(lo, hi):
            k = (lo * esz, hi * esz)
            best = -1
            for (a, b), i in last.items():
                if a < k[1] and k[0] < b:
                    best = max(best, i)
            return best
        target = -(-total // self.nbuckets)
        buckets, cur_lo, cur_rdy = [], 0, -1
        for lo, hi in spans:
            cur_rdy = max(cur_rdy, ready(lo, hi))
            if hi - cur_lo >= target:
                buckets.append((cur_lo, hi, cur_rdy))
                cur_lo, cur_rdy = hi, -1
        if cur_lo < total:
            buckets.append((cur_lo, total, cur_rdy))
        self.buckets = buckets
        # insert from the back so earlier indices stay valid; a bucket nobody writes (frozen parameters) goes to the end
        for lo, hi, rdy in sorted(buckets, key=lambda b: -b[2]):
            t = self.flat_g[lo:hi]
            c = Cmd('grad_allreduce', self._issue, (t,), rd=[t], wr=[t], pin=True)
            c.tag = 'comm'
            bwd.insert(rdy + 1 if rdy >= 0 else len(bwd), c)


def init_sync_bn(group=None, force=False):
    """Enable cross-rank statistics for every SynchronizedBatchNorm2d emitted from now on."""
    comm = SyncBNComm(group, force)
    _module.set_world(comm)
    return comm


def disable_sync_bn():
    _module.set_world(None)


def allreduce_grads(flat_grad, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)


def broadcast_params(model, src=0, group=None):
    """One-time parameter/buffer broadcast from rank 0 (DDP's constructor behaviour, train.py:173)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)


def shard_indices(n, rank, world, shuffle=True, seed=0, epoch=0):
    """The index list `torch.utils.data.DistributedSampler(dataset, num_replicas=world, rank=rank)` yields — what the
    reference builds (dataloaders/__init__.py:33, default shuffle=True, seed 0; it never calls set_epoch, so every epoch
    replays the epoch-0 permutation): randperm(n) from a generator seeded seed+epoch, padded by wrap-around to a multiple
    of `world`, then strided by rank.  tests/test_parallel_gloo.py checks equality against torch's sampler."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = (n + world - 1) // world * world
    pad = total - n
    if pad:
        idx += (idx * ((pad + n - 1) // n))[:pad]
    return idx[rank:total:world]
