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


class HipMailbox:
    """Transport of SmallComm on the GPU: libaddk's addk_comm_* entry points (csrc/comm.hip) — a fine-grained mailbox per rank, mapped by the
    peers through hipIpc, one single-workgroup launch per exchange."""

    def __init__(self):
        import ctypes as C
        from . import _lib as L
        self.C, self.L, self.lib = C, L, L.load()
        self.box, self.comm = C.c_void_p(), C.c_void_p()

    def alloc(self, world, max_bytes):
        h = (self.C.c_uint8 * 64)()
        self.L.check(self.lib.addk_comm_alloc(world, max_bytes, self.C.byref(self.box), h), 'comm_alloc')
        return bytes(h)

    def open(self, rank, world, max_bytes, handles):
        blob = (self.C.c_uint8 * (64 * world)).from_buffer_copy(b''.join(handles))
        self.L.check(self.lib.addk_comm_open(rank, world, max_bytes, self.box, blob, self.C.byref(self.comm)), 'comm_open')

    def allreduce(self, t, stream):
        return self.lib.addk_comm_allreduce(self.comm, t.data_ptr(), t.numel(), 1 if t.dtype == torch.float64 else 0, stream)

    def status(self):
        seq, err = self.C.c_int64(), self.C.c_int64()
        self.L.check(self.lib.addk_comm_status(self.comm, self.C.byref(seq), self.C.byref(err)), 'comm_status')
        return int(seq.value), int(err.value) & 0xFFFFFFFFFFFFFFFF

    def close(self):
        if self.comm or self.box:
            self.lib.addk_comm_close(self.comm, self.box)
        self.box, self.comm = self.C.c_void_p(), self.C.c_void_p()


class SmallComm:
    """The us-class exchange of the SyncBN statistics inside one node (`comm_allreduce_small`; the reference's master / slave
    pipes: modeling/sync_batchnorm/comm.py:56-129, batchnorm.py:95-108).  Host side of the protocol csrc/comm.hip documents:

      1. every rank allocates its mailbox and exports a 64-byte handle;  2. the handles (and host names: hipIpc is a one-node transport)
      travel through the process group;  3. every rank maps its peers;  4. a SELF-TEST exchanges a probe vector of each dtype and compares it
      with `dist.all_reduce` — the stock collective is the checker and stays the fallback;  5. every step so far is agreed collectively
      (all_reduce MIN of a success flag): either every rank uses the mailboxes or none does.

    `SmallComm.create` returns None when any rank failed any step (another node, no hipIpc, a failed probe)."""

    def __init__(self, transport, group, rank, world, max_bytes):
        self.t, self.group, self.rank, self.world, self.max_bytes = transport, group, rank, world, max_bytes
        self.calls = 0

    @staticmethod
    def _agree(ok, group, device):
        f = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        return float(f.item()) >= 1.0

    @classmethod
    def create(cls, group=None, max_bytes=1 << 16, device=None, transport=None, stream_of=None, ctl_device=None):
        import socket
        import sys
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        ctl = ctl_device if ctl_device is not None else device       # where the process group's own collectives take their tensors (gloo: CPU)
        t, err, handle = transport, None, b'\0' * 64
        try:
            if t is None:
                t = HipMailbox()
            handle = t.alloc(world, max_bytes)
        except Exception as e:          # noqa: BLE001  (any failure of the optional path means: fall back, on every rank)
            err = e
        peers = [None] * world
        dist.all_gather_object(peers, (socket.gethostname(), handle, err is None), group=group)
        ok = err is None and all(p[2] for p in peers) and len({p[0] for p in peers}) == 1
        if ok:
            try:
                t.open(rank, world, max_bytes, [p[1] for p in peers])
            except Exception as e:      # noqa: BLE001
                ok, err = False, e
        ok = cls._agree(ok, group, ctl)
        self = cls(t, group, rank, world, max_bytes) if ok else None
        if ok:
            # self-test against the stock collective: rank-dependent probe values, both dtypes, twice (both flag sets)
            try:
                stream = stream_of() if stream_of is not None else (torch.cuda.current_stream().cuda_stream if device.type == 'cuda' else 0)
                for rep in range(2):
                    for dt in (torch.float64, torch.float32):
                        probe = (torch.arange(64, dtype=dt, device=device) * 0.37 + 1.0) * (rank + 1 + rep)
                        ref = probe.to(ctl, copy=True)
                        rc = t.allreduce(probe, stream)
                        dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=group)
                        same = rc == 0 and bool(torch.allclose(probe.to(ctl), ref, rtol=1e-6 if dt == torch.float32 else 1e-14, atol=0))
                        ok = ok and same
                seq, e = t.status()
                ok = ok and e == 0 and seq == 4
            except Exception as e:      # noqa: BLE001
                ok, err = False, e
            ok = cls._agree(ok, group, ctl)
        if not ok:
            if rank == 0:
                sys.stderr.write('[addk] small-message SyncBN exchange unavailable (%s): stock all_reduce on every rank\n' %
                                 (str(err).splitlines()[0] if err is not None else 'a rank failed, ranks on several hosts, or the self-test differed'))
            try:
                if t is not None:
                    t.close()
            except Exception:           # noqa: BLE001
                pass
            return None
        return self

    def allreduce(self, t, stream):
        self.calls += 1
        return self.t.allreduce(t, stream)

    def fits(self, t):
        nbytes = t.numel() * t.element_size()
        return t.is_contiguous() and nbytes <= self.max_bytes and nbytes % 8 == 0 and t.data_ptr() % 8 == 0 and t.dtype in (torch.float32, torch.float64)

    def check(self):
        """Raise if an exchange timed out (a flag that never arrived: csrc/comm.hip sets the error word and returns instead of spinning)."""
        from ._lib import AddkError
        seq, err = self.t.status()
        if err:
            raise AddkError('small-message exchange %d timed out waiting for rank %d (after %d completed exchanges)' % ((err >> 8) & ((1 << 55) - 1), err & 0xFF, seq))
        return seq

    def close(self):
        self.t.close()


class SyncBNComm:
    def __init__(self, group=None, force=False, small=None):
        assert dist.is_initialized(), 'init_process_group first'
        import os
        self.group = group
        self.force = force          # run the exchange even at world_size 1 (exercises the N>1 kernels on one GPU)
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.calls = 0
        self.log = None             # tests: list collecting (numel, dtype) of every collective, in issue order
        # the small-message path: created lazily by the first exchange of a CUDA tensor (collectively: every rank reaches that call in the
        # same place of its launch list), or handed in (tests).  ADDK_COMM_SMALL=0: stock all_reduce only
        self.small = small
        self._small_tried = small is not None or os.environ.get('ADDK_COMM_SMALL', '1') == '0'

    def _allreduce(self, t, stream):
        if not self._small_tried and t.is_cuda and not torch.cuda.is_current_stream_capturing():
            self._small_tried = True
            self.small = SmallComm.create(self.group, device=t.device)
        if self.small is not None and self.small.fits(t):
            rc = self.small.allreduce(t, stream)
            self.calls += 1
            if self.log is not None:
                self.log.append(('stats', t.numel(), str(t.dtype)))
            return rc
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)      # enqueued on torch's current stream
        self.calls += 1
        if self.log is not None:
            self.log.append(('stats', t.numel(), str(t.dtype)))
        return 0

    def check(self):
        if self.small is not None:
            self.small.check()

    def close(self):
        if self.small is not None:
            self.small.close()
            self.small = None

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


def init_sync_bn(group=None, force=False, small=None):
    """Enable cross-rank statistics for every SynchronizedBatchNorm2d emitted from now on."""
    comm = SyncBNComm(group, force, small)
    _module.set_world(comm)
    return comm


def disable_sync_bn():
    """Back to per-replica BatchNorm; releases the small-message mailboxes of the communicator that was active (call it after the device has
    drained and before `dist.destroy_process_group()`)."""
    w = getattr(_module, '_world', None)
    if w is not None and hasattr(w, 'close'):
        w.close()
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
