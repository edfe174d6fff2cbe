"""Data-parallel pieces (one process per GPU, torch.distributed backend "nccl" == RCCL over xGMI):

 * SyncBNComm — the cross-rank reduction of BatchNorm statistics: per BN call ONE all-reduce of the fp64
   (sum, sumsq) vector (2C doubles) in forward and one of (dmean, dvar) (2C floats) in backward; replaces the
   reference's master/slave queue protocol (modeling/sync_batchnorm/comm.py:56-129, batchnorm.py:95-108).
 * allreduce_grads — gradient averaging across ranks on the step's FLAT gradient buffer: a single large RCCL
   all-reduce (45 MB at F=20) instead of DDP's per-bucket copies (train.py:173-175).

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

    def _allreduce(self, t, stream):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)      # enqueued on torch's current stream
        self.calls += 1
        return 0

    def _allreduce_multi(self, ts, stream):
        """The exchanges of several independent BatchNorms (one dependency level of the launch list) as ONE grouped RCCL
        call (ncclGroupStart/End through torch's coalescing manager): one launch and one latency instead of len(ts)."""
        cm = getattr(dist, '_coalescing_manager', None)          # private API: fall back to single calls if it is not there
        if cm is not None and dist.get_backend(self.group) == 'nccl' and len(ts) > 1:
            with cm(self.group, device=ts[0].device, async_ops=False):
                for t in ts:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for t in ts:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        return 0

    def emit_allreduce(self, g, lst, vec):
        """Append an all-reduce of `vec` (Vec) to command list `lst`.  Forward statistics are fp64 pairs."""
        t = vec.view()
        if lst is g.fwd:
            t = t.view(torch.float64)
        c = g._add(lst, 'allreduce', self._allreduce, t, rd=[vec], wr=[vec], pin=True)     # RCCL call: main stream only
        c.payload = t


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
