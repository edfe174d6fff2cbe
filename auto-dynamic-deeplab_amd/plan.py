"""Static execution plans for the ADD hot path.

A module tree (modeling/*.py) is *emitted* once per (input shape, mode) into a `Graph`:
a flat list of C-ABI kernel launches over pre-allocated NHWC buffers plus the matching
backward list (built in reverse with first-touch/accumulate flags decided at build
time, so no memsets and no separate add kernels).  Replaying a plan is a tight loop of
ctypes calls on one HIP stream — or a single hipGraph launch once captured.

Activations are *lazy*: `Act = (raw, bn, relu)` means relu?(a*raw+b) with (a,b) owned by
the producing BatchNorm; consumers apply it while staging their tiles, so BatchNorm/ReLU
never cost a pass over HBM.
"""
import collections
import os
import ctypes as C

import torch

from . import _lib as L


TRACE_BN = None        # diagnostics (tests/tools): a list that receives (BatchNorm module, raw TRef) of every Graph.bn call


class Buf:
    """Flat fp32 device buffer."""
    __slots__ = ('t', 'ptr', 'n', 'gbuf', 'ginit')

    def __init__(self, n, device, zero=False):
        self.n = int(n)
        self.t = (torch.zeros if zero else torch.empty)(max(self.n, 4), dtype=torch.float32, device=device)
        self.ptr = self.t.data_ptr()
        self.gbuf = None       # gradient twin
        self.ginit = []        # initialised channel intervals of this (gradient) buffer


class TRef:
    """NHWC view (n,h,w,c) -> buf[off + ((n*H+h)*W+w)*ld + c]."""
    __slots__ = ('buf', 'off', 'N', 'H', 'W', 'C', 'ld')

    def __init__(self, buf, off, N, H, W, C, ld):
        self.buf, self.off, self.N, self.H, self.W, self.C, self.ld = buf, off, N, H, W, C, ld

    @property
    def ptr(self):
        return self.buf.ptr + 4 * self.off

    @property
    def P(self):
        return self.N * self.H * self.W

    def chan(self, c0, C):
        assert 0 <= c0 and c0 + C <= self.C
        return TRef(self.buf, self.off + c0, self.N, self.H, self.W, C, self.ld)

    def view(self):
        """torch view [N,H,W,C] (strided)."""
        return self.buf.t.as_strided((self.N, self.H, self.W, self.C),
                                     (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), self.off)


class Vec:
    __slots__ = ('buf', 'off', 'n')

    def __init__(self, buf, off, n):
        self.buf, self.off, self.n = buf, off, n

    @property
    def ptr(self):
        return self.buf.ptr + 4 * self.off

    def view(self):
        return self.buf.t[self.off:self.off + self.n]


class LateVec:
    """A small device vector whose storage is assigned only after the launch list has been ordered (Graph._bind_late):
    the statistics vectors exchanged by the SyncBN all-reduces of ONE dependency level are laid out back to back in one
    arena, so a level costs a single plain `all_reduce` on a contiguous tensor.  `n` counts 4-byte slots; `f64` marks
    fp64 pairs (forward statistics).  Until bound, dependency tracking uses the object itself as its region."""
    __slots__ = ('n', 'f64', 'buf', 'off', 'binders')

    def __init__(self, n, f64=False):
        self.n, self.f64, self.buf, self.off, self.binders = int(n), f64, None, 0, []

    @property
    def ptr(self):
        if self.buf is None:
            raise RuntimeError('LateVec used before Graph.finalize() bound it')
        return self.buf.ptr + 4 * self.off

    def bind(self, buf, off):
        self.buf, self.off = buf, off
        for b in self.binders:
            b(self.ptr)

    def view(self):
        t = self.buf.t[self.off:self.off + self.n]
        return t.view(torch.float64) if self.f64 else t


class BNState:
    """Per-call state of one BatchNorm application: lazy affine + saved statistics + the
    partial (dA,dB) slabs its consumers produce in backward."""

    def __init__(self, mod, C, a, b, training, count):
        self.mod, self.C, self.a, self.b = mod, C, a, b
        self.training, self.count = training, count
        self.mean = self.invstd = None
        self.slabs = []          # [(ptr, rows)]
        self.c1 = self.c2 = None


class Act:
    """Lazy activation: value = relu?(a*raw+b).  `rs` (a Resample, shared by every Act derived from one Graph.resize call): `raw` is
    the bilinear interpolation of another map and has NOT been written yet — a 1x1 convolution that consumes it samples the source
    map itself (addk_src.rs_hw); any other consumer makes Graph.src / Graph.lz emit the stand-alone resize launch first."""
    __slots__ = ('raw', 'bn', 'relu', 'needs_grad', 'zero', 'rs')

    def __init__(self, raw, bn=None, relu=False, needs_grad=False, zero=False, rs=None):
        self.raw, self.bn, self.relu, self.needs_grad, self.zero, self.rs = raw, bn, relu, needs_grad, zero, rs

    N = property(lambda s: s.raw.N)
    H = property(lambda s: s.raw.H)
    W = property(lambda s: s.raw.W)
    C = property(lambda s: s.raw.C)


class Resample:
    """Deferred bilinear resize (Graph.resize): `src` is the TRef of the map to sample, `emit()` appends the stand-alone launch that
    writes the resized tensor, `done` says that some launch writes it (that launch, or a folding consumer's copy-out in training)."""
    __slots__ = ('src', 'emit', 'done')

    def __init__(self, src, emit):
        self.src, self.emit, self.done = src, emit, False


class Cmd:
    """One kernel launch of a plan: fn(*args, stream).  `rd`/`wr` are the memory regions it reads / writes
    (keys from `_region`), used to schedule independent launches on parallel HIP streams."""
    __slots__ = ('name', 'fn', 'args', 'rd', 'wr', 'stream', 'waits', 'event', 'pin', 'tag', 'tags', 'payload', 'bkey', 'arena', 'members')

    def __init__(self, name, fn, args, rd=(), wr=(), pin=False):
        self.name, self.fn, self.args = name, fn, list(args)
        self.rd, self.wr = [k for k in (_region(x) for x in rd) if k], [k for k in (_region(x) for x in wr) if k]
        self.stream, self.waits, self.event, self.pin, self.tag = 0, (), None, pin, ''
        self.tags = None
        self.payload = None            # argument struct of a launch that has a table-driven batched form (level_batch)
        self.bkey = 0                  # kernel-variant key: only launches with equal keys share a batch
        self.arena, self.members = None, 1

    def __iter__(self):                # unpacks like the (name, fn, args) triple it replaces
        return iter((self.name, self.fn, self.args))

    def __getitem__(self, i):
        return (self.name, self.fn, self.args)[i]


def _region(x):
    """(buffer id, lo, hi): channel interval inside the pixel row for NHWC views, whole buffer otherwise."""
    if x is None:
        return None
    if isinstance(x, TRef):
        lo = x.off % x.ld if x.ld else 0
        return (id(x.buf), lo, lo + x.C)
    if isinstance(x, Vec):
        return (id(x.buf), 0, 1 << 30)
    if isinstance(x, LateVec):
        return (id(x), 0, 1 << 30)
    if isinstance(x, Buf):
        return (id(x), 0, 1 << 30)
    if isinstance(x, torch.Tensor):
        # byte interval inside the storage: views of one flat parameter / gradient buffer (train.TrainStep) are distinct regions
        lo = x.storage_offset() * x.element_size()
        return (x.untyped_storage().data_ptr(), lo, lo + max(1, x.numel()) * x.element_size())
    if isinstance(x, tuple):
        return x
    raise TypeError(type(x))


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def schedule(cmds, nstreams):
    """Assign every command a stream and the cross-stream waits it needs.  A command follows the stream of its most
    recent dependency when that dependency is still the tail of its stream (a chain stays on one stream, no event);
    otherwise it opens on the least recently used stream and waits on events.  Program order inside a stream plus the
    recorded waits preserve every RAW/WAW/WAR relation of the sequential list."""
    writers, readers = {}, {}            # buffer id -> [(region, idx)]
    tail = [-1] * nstreams               # index of the last command on each stream
    # vector clocks: clock[i][t] = latest command of stream t known to have completed before command i starts.  A wait
    # that a previous wait already implies TRANSITIVELY (A -> B -> C and A -> C) is dropped (fewer event edges in the graph).
    clock = []
    for i, c in enumerate(cmds):
        deps = set()
        for k in c.rd:
            deps.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
        for k in c.wr:
            deps.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
            deps.update(j for r, j in readers.get(k[0], ()) if _overlap(r, k))
        if c.pin or nstreams == 1:
            st = 0
        elif deps:
            last = max(deps)
            ls = cmds[last].stream
            st = ls if tail[ls] == last else min(range(nstreams), key=lambda t: tail[t])
        else:
            st = min(range(nstreams), key=lambda t: tail[t])
        # hipStreamEndCapture (ROCm 7.2) crashes when two SIDE streams wait on each other (measured with
        # scripts/capture_probe.py: 'mutual12' dumps core, 'ordered' does not).  Side stream s therefore only ever waits
        # on side streams t > s (and on the main stream 0, which may wait on anybody); a command that would need the
        # other direction runs on the main stream instead.
        if st != 0:
            vc0 = clock[tail[st]] if tail[st] >= 0 else [-1] * nstreams
            if any(0 < cmds[j].stream < st and vc0[cmds[j].stream] < j for j in deps):
                st = 0
        vc = list(clock[tail[st]]) if tail[st] >= 0 else [-1] * nstreams
        waits = []
        for j in sorted(deps, reverse=True):
            t = cmds[j].stream
            if t != st and vc[t] < j:
                waits.append(j)
                vc = [max(a, b) for a, b in zip(vc, clock[j])]
        vc[st] = i
        clock.append(vc)
        c.stream, c.waits = st, tuple(waits)
        tail[st] = i
        for k in c.wr:
            writers[k[0]] = [(r, j) for r, j in writers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])] + [(k, i)]
            readers[k[0]] = [(r, j) for r, j in readers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])]   # only readers this write fully covers are implied by it
        for k in c.rd:
            readers.setdefault(k[0], []).append((k, i))
    need = set(j for c in cmds for j in c.waits)
    for j in need:
        cmds[j].event = True
    return cmds


class Graph:
    def __init__(self, device, training, want_grad, world=None):
        self.lib = L.load()
        self.device = device
        self.training = training
        self.want_grad = want_grad
        self.world = world            # SyncBN communicator (parallel.SyncBNComm) or None
        self.fwd, self.bwd = [], []   # [(name, fn, args)]
        self._bwd_emitters = []       # closures emitting backward commands, run in reverse
        self.keep = []                # keep ctypes structs / tensors alive
        self.pgrad = {}               # param -> grad tensor
        self.pginit = set()
        self._pcols = {}
        self.params = []              # ordered unique params touched
        self.nbt = {}                 # BatchNorm module -> num_batches_tracked increments per forward
        self._wgrads = []             # wgrad arg structs sharing one scratch buffer (launches are stream-ordered)
        self.nbytes = 0
        self._bufs = []               # owns every device buffer: kernels only see raw pointers
        self.nstreams = 1             # set by finalize(nstreams=k)
        self.tag = ''                 # segment label stamped on emitted commands (stem / cell / low / aspp / decoder)
        self._evalbn, self._evalbn_cmd = [], None
        self._dwreds = []             # deferred depthwise weight-gradient reductions (item, workspace, grad tensor)
        self._packs, self._pack_cmd = [], None   # hoisted weight packs of the halo-patch conv launches (descriptor bytes)
        self.meta = []                # per-launch algorithmic work of the dense convs (bench / roofline)
        self.grad_sync = None         # parallel.GradSync of a data-parallel train step (set before finalize)
        # the arithmetic mode the packed weights and kernel choices of this plan were built for: replaying the launch list under
        # another mode would hand bf16-plane packs to the fp32 kernel — run() / run_parallel() refuse it
        self.precision = int(self.lib.addk_get_conv_precision())

    # ---------------- memory ----------------
    def buf(self, n, zero=False):
        b = Buf(n, self.device, zero)
        self._bufs.append(b)
        self.nbytes += 4 * b.n
        return b

    def tensor(self, N, H, W, Cc, ld=None):
        ld = ld or (Cc + 3) // 4 * 4
        return TRef(self.buf(N * H * W * ld, zero=(ld != Cc)), 0, N, H, W, Cc, ld)

    def vec(self, n, zero=False):
        return Vec(self.buf(n, zero), 0, n)

    def grad(self, tref):
        """Gradient twin of a TRef (same geometry in the twin buffer)."""
        b = tref.buf
        if b.gbuf is None:
            b.gbuf = self.buf(b.n)
        return TRef(b.gbuf, tref.off, tref.N, tref.H, tref.W, tref.C, tref.ld)

    def acc_flag(self, gref):
        """0 the first time a channel interval of a gradient buffer is written in backward order, 1 afterwards."""
        iv = gref.buf.ginit
        c0 = gref.off % gref.ld if gref.ld else 0
        c1 = c0 + gref.C
        covered = sum(max(0, min(c1, b) - max(c0, a)) for a, b in iv)
        if covered == 0:
            iv.append((c0, c1))
            return 0
        if covered == c1 - c0:
            return 1
        raise NotImplementedError('partially initialised gradient interval [%d,%d) vs %s' % (c0, c1, iv))

    def grad_ready(self, tref):
        b = tref.buf.gbuf
        if b is None:
            return False
        c0 = tref.off % tref.ld if tref.ld else 0
        c1 = c0 + tref.C
        return sum(max(0, min(c1, y) - max(c0, x)) for x, y in b.ginit) == c1 - c0

    def param(self, p):
        if p not in self.pgrad:          # Tensor hashes by identity
            self.pgrad[p] = None
            self.params.append(p)
        return p.data_ptr()

    def param_grad(self, p, cols=None):
        """(ptr, accumulate) of the plan-owned gradient of parameter p.  `cols` = (c0, c1) is the input-channel
        interval a launch writes (virtual-concat sources and the ASPP image-pool fold write disjoint column
        ranges of one weight): first touch of an interval overwrites, later touches accumulate."""
        self.param(p)
        if self.pgrad[p] is None:
            views = getattr(self, 'pgrad_views', None)
            self.pgrad[p] = views[p] if views is not None and p in views else torch.empty_like(p)
            self._pcols[p] = []
        iv = self._pcols[p]
        c0, c1 = cols if cols is not None else (0, 1 << 30)
        covered = sum(max(0, min(c1, b) - max(c0, a)) for a, b in iv)
        if covered == 0:
            iv.append((c0, c1))
            acc = 0
        elif covered == c1 - c0 or (cols is None and iv == [(0, 1 << 30)]):
            acc = 1
        else:
            raise NotImplementedError('partially initialised weight-gradient columns')
        self.pginit.add(p)
        return self.pgrad[p].data_ptr(), acc

    def finalize(self, nstreams=1):
        """Emit the backward list (reverse op order), schedule both lists on `nstreams` streams and allocate one wgrad
        scratch per stream (launches of one stream are ordered, so they can share it)."""
        for em in reversed(self._bwd_emitters):
            em()
        self._bwd_emitters = []
        self.nstreams = max(1, int(nstreams))
        if self._evalbn:
            n = len(self._evalbn)
            arr = (L.BnEvalEntry * n)(*self._evalbn)
            host = torch.frombuffer(arr, dtype=torch.uint8)
            tab = host.to(self.device) if self.device.type == 'cuda' else host.clone()
            self.keep += [arr, tab]
            self._evalbn_cmd.args[0], self._evalbn_cmd.args[1] = tab.data_ptr(), n
        self._emit_batched_dw_reductions()
        self._emit_batched_wgrads()
        if self._packs:
            host = bytearray(b''.join(self._packs))
            tab = torch.frombuffer(host, dtype=torch.uint8)
            tab = tab.to(self.device) if self.device.type == 'cuda' else tab.clone()
            self.keep.append(tab)
            self._pack_cmd.args[0], self._pack_cmd.args[1] = tab.data_ptr(), len(self._packs)
        batch = ((self.training and self.want_grad) or getattr(self, 'reorder', False)) and os.environ.get('ADDK_LEVEL_BATCH', '1') == '1'
        for lst in (self.fwd, self.bwd):
            self._bind_late(lst, by_level=batch)          # storage of the exchanged statistics vectors: one arena per level
            if batch:
                self._level_batch(lst)
        if self.grad_sync is not None:
            self.grad_sync.insert(self, self.bwd)          # bucketed gradient all-reduces, each right after its last producer
        for lst in (self.fwd, self.bwd):
            for i, c in enumerate(lst):
                if not isinstance(c, Cmd):           # commands appended as plain triples (e.g. collectives): pinned to the main stream
                    lst[i] = Cmd(c[0], c[1], c[2], pin=True)
            schedule(lst, self.nstreams)
            for c in lst:
                if c.event:
                    c.event = torch.cuda.Event()

    _BATCHED = {'bn_finalize': ('addk_bn_finalize_batch', lambda p: p.C), 'bn_bwd': ('addk_bn_bwd_batch', lambda p: p.C),
                'bn_bwd_apply': ('addk_bn_bwd_apply_batch', lambda p: p.P),
                'slab_reduce': ('addk_slab_reduce_batch', lambda p: p.C), 'bn_bwd_coeffs': ('addk_bn_bwd_coeffs_batch', lambda p: p.C),
                'conv_fwd': ('addk_conv_fwd_batch_prepare', 'addk_conv_batch_run'), 'conv_dgrad': ('addk_conv_dgrad_batch_prepare', 'addk_conv_batch_run'),
                'dw_fwd': ('addk_dw_fwd_batch_prepare', 'addk_dw_batch_run'), 'dw_bwd': ('addk_dw_bwd_batch_prepare', 'addk_dw_batch_run'),
                'sep_fwd': ('addk_sep_fwd_batch_prepare', 'addk_sep_batch_run'),
                'sep_bwd': ('addk_sep_bwd_batch_prepare', 'addk_sep_bwd_batch_run'),
                'resize_bwd': ('addk_resize_bwd_batch_prepare', 'addk_resize_bwd_batch_run'),
                'allreduce': (None, None)}

    @staticmethod
    def _levels(lst):
        """Dependency level (longest path from the inputs) of every command of a launch list."""
        writers, readers, level = {}, {}, []
        for i, c in enumerate(lst):
            deps = set()
            for k in c.rd:
                deps.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
            for k in c.wr:
                deps.update(j for r, j in writers.get(k[0], ()) if _overlap(r, k))
                deps.update(j for r, j in readers.get(k[0], ()) if _overlap(r, k))
            level.append(1 + max(level[j] for j in deps) if deps else 0)
            for k in c.wr:
                writers[k[0]] = [(r, j) for r, j in writers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])] + [(k, i)]
                readers[k[0]] = [(r, j) for r, j in readers.get(k[0], ()) if not (r[1] >= k[1] and r[2] <= k[2])]   # only readers this write fully covers are implied by it
            for k in c.rd:
                readers.setdefault(k[0], []).append((k, i))
        return level

    def _bind_late(self, lst, by_level):
        """Give every exchanged statistics vector (LateVec payload of an 'allreduce' command) its storage.  With level
        batching the vectors of one dependency level share one arena, in list order, and `_level_batch` replaces their
        commands by ONE all-reduce of the arena; otherwise each gets its own buffer.  Runs before any table-driven batch is
        built, because the producers (slab_reduce / bn_bwd) and consumers (bn_finalize / bn_bwd_coeffs) of a vector sit on
        other levels and carry its address in their argument structs."""
        ars = [(i, c) for i, c in enumerate(lst) if isinstance(c, Cmd) and c.name == 'allreduce' and isinstance(c.payload, LateVec)]
        if not ars:
            return
        level = self._levels(lst) if by_level else list(range(len(lst)))
        groups = collections.defaultdict(list)
        for i, c in ars:
            groups[(level[i], c.payload.f64)].append(c)
        for (_, f64), cs in sorted(groups.items()):
            unit = 4                                   # slots of 16 bytes: every vector starts 16-byte aligned
            total = sum((c.payload.n + unit - 1) // unit * unit for c in cs)
            arena = self.buf(total, zero=True)
            off = 0
            for c in cs:
                c.payload.bind(arena, off)
                off += (c.payload.n + unit - 1) // unit * unit
            for c in cs:
                c.arena = (arena, total, f64)

    def _level_batch(self, lst):
        """Reorder a launch list by dependency LEVEL (longest path from the inputs) and merge the mutually independent
        BatchNorm vector launches of one level into one table-driven launch each.  The cell DAG offers ~10 independent
        branches per level, so the 312 finalize / 312 backward / 312 apply launches of a step — each a ~10 us latency
        chain on the critical path — shrink to one launch per level.  Commands of one level never depend on each other,
        so any order inside a level is valid; program order of the original list is kept."""
        level = Graph._levels(lst)
        buckets = collections.defaultdict(list)
        for i, c in enumerate(lst):
            buckets[level[i]].append(c)
        # [r4] kinds left as single launches.  In the TRAINING step the fused SepConv halves run better one by one (33.6-33.7 vs 33.95-34.0 ms per
        # step, ABAB on one box): a merged launch waits for its latest input and holds every consumer back, and these launches are long enough
        # (15-70 us) for the two streams to overlap them; every other kind measured slower or equal un-merged (bn_bwd +1.3 ms, bn_bwd_apply +0.9,
        # conv_dgrad +0.4), and inference (exit latency, forward segment) is equal or better with the halves merged.  ADDK_NO_BATCH=kind,... overrides.
        env = os.environ.get('ADDK_NO_BATCH')
        single = set(filter(None, env.split(','))) if env is not None else ({'sep_fwd', 'sep_bwd'} if (getattr(self, 'training', False) and getattr(self, 'want_grad', False)) else set())
        out = []
        for lv in sorted(buckets):
            groups = collections.defaultdict(list)
            for c in buckets[lv]:
                if c.name in self._BATCHED and c.payload is not None and c.name not in single:
                    # the segment tag is part of the key: a merged launch never mixes network segments (stem / cell / low / aspp / decoder), so the
                    # per-segment figures bench.segment_roofline reports time exactly that segment's work; all-reduces stay per level
                    groups[(c.name, c.bkey, '' if c.name == 'allreduce' else c.tag)].append(c)
                else:
                    out.append(c)
            for (name, _, _), cs in groups.items():
                if len(cs) == 1 and name != 'allreduce':
                    out.append(cs[0])
                    continue
                fname, size_of = self._BATCHED[name]
                n = len(cs)
                if name == 'allreduce':        # SyncBN exchanges of one level: their vectors are contiguous in one arena (_bind_late) -> ONE plain all-reduce
                    arena, total, f64 = cs[0].arena
                    assert all(c.arena[0] is arena for c in cs)
                    t = arena.t[:total]
                    m = Cmd('allreduce_packed', self.world._allreduce, (t.view(torch.float64) if f64 else t,), pin=True)
                    m.rd = [r for c in cs for r in c.rd]
                    m.wr = [r for c in cs for r in c.wr]
                    m.tag = cs[0].tag
                    m.tags = {c.tag for c in cs}
                    m.members = len(cs)
                    out.append(m)
                    continue
                arr = (type(cs[0].payload) * n)(*[c.payload for c in cs])
                if isinstance(size_of, str):   # pointwise / depthwise convs: the library turns the argument structs into kernel descriptors
                    prep = getattr(self.lib, fname)
                    meta = (C.c_int64 * 8)()
                    size = prep(arr, n, None, 0, meta)
                    if size < 0:
                        L.check(int(size), fname)
                    blob = (C.c_uint8 * size)()
                    rc = prep(arr, n, blob, size, meta)
                    if rc < 0:
                        L.check(int(rc), fname)
                    host = torch.frombuffer(bytearray(bytes(blob)), dtype=torch.uint8)
                    tab = host.to(self.device) if self.device.type == 'cuda' else host.clone()
                    self.keep += [arr, tab, meta]
                    m = Cmd(name + '_batch', getattr(self.lib, size_of), (tab.data_ptr(), meta))
                else:
                    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                    tab = host.to(self.device) if self.device.type == 'cuda' else host.clone()
                    self.keep += [arr, tab]
                    m = Cmd(name + '_batch', getattr(self.lib, fname), (tab.data_ptr(), n, max(int(size_of(c.payload)) for c in cs)))
                m.rd = [r for c in cs for r in c.rd]
                m.wr = [r for c in cs for r in c.wr]
                m.tag = cs[0].tag
                m.tags = {c.tag for c in cs}           # a merged launch may hold members of several network segments (bench.segment_roofline checks)
                out.append(m)
        lst[:] = out

    def _hoist_pack(self, desc_fn, args, weight, wpk, create):
        """Move the weight pack of a halo-patch conv launch out of the launch: all packs of the plan (forward and data
        gradient: the weights only change in the optimizer) run as ONE launch placed before the first such conv."""
        if self._pack_cmd is None:
            if not create:
                return
            self._pack_cmd = self._add(self.fwd, 'conv_pack_batch', self.lib.addk_conv_pack_batch, None, 0)
        nb = int(self.lib.addk_conv_pack_desc_bytes())
        buf = (C.c_uint8 * nb)()
        L.check(desc_fn(C.byref(args), buf), 'conv_pack_desc')
        self._packs.append(bytes(buf))
        args.wpack_ready = 1
        self._pack_cmd.rd.append(_region(weight))
        self._pack_cmd.wr.append(_region(wpk))

    def _emit_batched_dw_reductions(self):
        """One launch reduces the weight-gradient partials of every depthwise conv of the backward pass; two items that
        write the same weight (a module used twice) go to consecutive launches, in emission order."""
        waves = []
        for it, ws, grad in self._dwreds:
            if not waves or it.dw in waves[-1][1]:
                waves.append(([], set()))
            waves[-1][0].append((it, ws, grad))
            waves[-1][1].add(it.dw)
        nchunk = int(os.environ.get('ADDK_WGRAD_CHUNKS', '3'))
        parts = []
        for items, _ in waves:           # cut in emission (= backward) order so the early parts overlap with the backward pass
            k = nchunk if len(items) >= 8 * nchunk else 1
            per = -(-len(items) // k)
            parts += [items[c * per:(c + 1) * per] for c in range(k) if items[c * per:(c + 1) * per]]
        for items in parts:
            n = len(items)
            arr = (L.DwWreduceItem * n)(*[it for it, _, _ in items])
            host = bytes(arr)
            tab = torch.frombuffer(bytearray(host), dtype=torch.uint8)
            tab = tab.to(self.device) if self.device.type == 'cuda' else tab.clone()
            self.keep += [arr, tab]
            self._add(self.bwd, 'dw_wreduce_batch', self.lib.addk_dw_wreduce_batch, tab.data_ptr(), n,
                      rd=[ws for _, ws, _ in items], wr=[g for _, _, g in items])

    def _emit_batched_wgrads(self):
        """All weight gradients of the backward pass are deferred to its end and launched in a few batches: one launch
        per tile configuration covers every conv that uses it (their dy / activation buffers are never reused, so
        deferral is safe).  Two gradients that accumulate into the same weight columns (the ASPP/decoder head is
        shared by the exits) must not sit in the same batch: the second goes to a later wave."""
        if not self._wgrads:
            return
        lib = self.lib
        waves = []          # [ {cfg: [(wa, rd, grad)]} , set(regions) ]
        for wa, rd, grad, region in self._wgrads:
            cfg = (C.c_int32 * 4)()
            L.check(lib.addk_conv_wgrad_config(C.byref(wa), cfg), 'conv_wgrad_config')
            key = (cfg[0], cfg[1], cfg[2])
            for groups, used in waves:
                if not any(r[0] == region[0] and r[1] < region[2] and region[1] < r[2] for r in used):
                    break
            else:
                groups, used = {}, []
                waves.append((groups, used))
            used.append(region)
            groups.setdefault(key, []).append((wa, rd, grad))
        nchunk = int(os.environ.get('ADDK_WGRAD_CHUNKS', '3'))
        for groups, _ in waves:
            split = {}
            for key, items in groups.items():
                # big groups (the cells' 1x1 / dilated convs: ~100-300 ops) are cut into `nchunk` batches in emission order
                # (= backward order): the early batches only need the gradients of the late cells, so the level pass can
                # place them in the middle of the backward pass where they overlap with it on the second stream
                k = nchunk if len(items) >= 32 * nchunk else 1
                per = -(-len(items) // k)
                for c in range(k):
                    if items[c * per:(c + 1) * per]:
                        split[(key, c)] = items[c * per:(c + 1) * per]
            for (key, _c), items in split.items():
                n = len(items)
                arr = (L.ConvWgradArgs * n)()
                for i, (wa, _, _) in enumerate(items):
                    ws = self.buf(wa.ws_floats)
                    wa.ws = ws.ptr
                    arr[i] = wa
                meta = (C.c_int64 * 8)()
                size = lib.addk_conv_wgrad_batch_prepare(arr, n, None, 0, meta)
                if size < 0:
                    L.check(int(size), 'conv_wgrad_batch_prepare')
                host = (C.c_uint8 * size)()
                rc = lib.addk_conv_wgrad_batch_prepare(arr, n, host, size, meta)
                if rc < 0:
                    L.check(int(rc), 'conv_wgrad_batch_prepare')
                if self.device.type == 'cuda':
                    blob = torch.frombuffer(host, dtype=torch.uint8).to(self.device)
                else:                       # dry-run planning on CPU (tests)
                    blob = torch.frombuffer(host, dtype=torch.uint8).clone()
                self.keep += [arr, meta, blob]
                self.nbytes += int(size)
                if os.environ.get('ADDK_DEBUG_WGRAD') == '1':      # algorithmic work of this batch (tuning aid)
                    fl = sum(2.0 * w.N * w.OH * w.OW * w.Cout * w.KH * w.KW * w.src.C for w, _, _ in items)
                    by = sum(4.0 * (w.N * w.OH * w.OW * w.Cout + w.N * w.H * w.W * w.src.C) for w, _, _ in items)
                    shapes = collections.Counter((w.OH, w.OW, w.Cout, w.src.C, w.KH, w.stride) for w, _, _ in items)
                    print('wgrad batch kind/cty/ctz=%s n=%d blocks=%d  %.1f GF  %.1f MB min traffic  %s' % (
                        key, n, meta[5], fl / 1e9, by / 1e6, dict(shapes)), flush=True)
                self._add(self.bwd, 'conv_wgrad_batch', lib.addk_conv_wgrad_batch_run, blob.data_ptr(), meta,
                          rd=[r for _, rd, _ in items for r in rd], wr=[g for _, _, g in items])

    # ---------------- command helpers ----------------
    def _add(self, lst, name, fn, *args, rd=(), wr=(), pin=False):
        c = Cmd(name, fn, args, rd, wr, pin)
        c.tag = self.tag
        lst.append(c)
        return c

    def _check_mode(self):
        if int(self.lib.addk_get_conv_precision()) != self.precision:
            raise L.AddkError('this plan was built under arithmetic mode %d and cannot be replayed under mode %d: its packed weights and '
                              'kernel choices belong to the mode it was built in (rebuild the plan after addk.set_precision)'
                              % (self.precision, int(self.lib.addk_get_conv_precision())))

    def run(self, cmds, stream):
        """Sequential replay on one stream."""
        self._check_mode()
        for name, fn, args in cmds:
            rc = fn(*args, stream)
            if rc:
                L.check(rc, name)

    def run_parallel(self, cmds, main):
        """Replay a scheduled list on `nstreams` HIP streams (stream 0 = `main`, a torch.cuda.Stream): independent
        branches of the cell DAG overlap, which is what fills 256 CUs when single launches are small.  Works eagerly
        and under hipGraph capture (fork/join through events)."""
        ns = self.nstreams
        if self.device.type != 'cuda':          # dry-run planning on CPU (tests): launches are stubbed
            return self.run(cmds, current_stream())
        self._check_mode()
        if main is None:
            main = torch.cuda.current_stream()
        if ns == 1:
            return self.run(cmds, main.cuda_stream)
        streams = [main] + self._side_streams()
        fork = torch.cuda.Event()
        fork.record(main)
        for s in streams[1:]:
            s.wait_event(fork)
        ptrs = [s.cuda_stream for s in streams]
        for c in cmds:
            s = streams[c.stream]
            for j in c.waits:
                s.wait_event(cmds[j].event)
            rc = c.fn(*c.args, ptrs[c.stream])
            if rc:
                L.check(rc, c.name)
            if c.event is not None:
                c.event.record(s)
        for s in streams[1:]:
            e = torch.cuda.Event()
            e.record(s)
            main.wait_event(e)

    def _side_streams(self):
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.nstreams - 1)]
        return self._streams

    def _pending(self, act):
        return act.rs is not None and not act.rs.done

    def src(self, act, relu_in=False, fold=False):
        """Source descriptor of a lazy activation.  fold=True (a consumer that samples a deferred resize itself): the descriptor
        points at the map to interpolate; otherwise a still-deferred resize is emitted as its own launch now."""
        s = L.Src()
        if fold and self._pending(act):
            r = act.rs.src
            s.x, s.ld, s.rs_hw = r.ptr, r.ld, (r.H << 16) | r.W
        else:
            if self._pending(act):
                act.rs.emit()
            s.x, s.ld = act.raw.ptr, act.raw.ld
        if act.bn is not None:
            s.a, s.b = act.bn.a.ptr, act.bn.b.ptr
        s.C, s.relu = act.raw.C, int(bool(act.relu or relu_in))
        return s

    def lz(self, act, fold=False):
        """Memory regions a consumer of lazy activation `act` reads."""
        if fold and self._pending(act):
            first = act.rs.src
        else:
            if self._pending(act):
                act.rs.emit()
            first = act.raw
        return [first] + ([act.bn.a, act.bn.b] if act.bn is not None else [])

    def _dab(self, act, rows):
        """Allocate a (dA,dB) partial slab for a lazy source whose BN is in training mode."""
        if act.bn is None or not act.bn.training or not self.want_grad:
            return None
        return self.buf(rows * act.C * 4)      # fp64 [rows][C][2]

    # ---------------- ops ----------------
    def conv(self, srcs, weight, Cout, k, stride=1, pad=0, dil=1, relu_in=False, bias=None, bias_n=None,
             out=None, stats=None, stats_ld=0, out_hw=None, w_choff=0, cin_total=None, fwd=True):
        """Dense conv over virtually concatenated `srcs` (list of Act).  Returns the raw output TRef.  fwd=False registers only
        the backward pass (the forward launch is part of a fused command, sep_half)."""
        lib = self.lib
        a0 = srcs[0]
        N, H, W = a0.N, a0.H, a0.W
        if out_hw is None:
            OH = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
            OW = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        else:
            OH, OW = out_hw
        csum = sum(s.C for s in srcs)
        cin_total = cin_total or csum
        if out is None:
            out = self.tensor(N, OH, OW, Cout)
        assert (out.N, out.H, out.W, out.C) == (N, OH, OW, Cout), 'conv output geometry'
        assert len(srcs) <= L.MAX_SRC
        wptr = self.param(weight)
        ldw = k * k * cin_total
        assert weight.numel() == Cout * ldw, 'weight %s does not match Cout=%d k=%d cin=%d' % (tuple(weight.shape), Cout, k, cin_total)
        if fwd:
            ar = L.ConvArgs()
            # a deferred bilinear resize in front of a 1x1 convolution (ADD.py:76-77,84-90) is sampled by the convolution itself where
            # the pointwise kernels cover the shape (addk_conv_fwd_resample_ok): no resize launch, no resized tensor at inference; in
            # training the launch also writes the resized tensor, which the backward pass reads (ADDK_FOLD_RESIZE=0: never)
            fold = (k == 1 and stride == 1 and pad == 0 and out_hw is None and any(self._pending(s) for s in srcs)
                    and os.environ.get('ADDK_FOLD_RESIZE', '1') == '1')
            ar.nsrc = len(srcs)
            ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, OH, OW
            ar.KH = ar.KW = k
            ar.stride, ar.pad, ar.dil, ar.Cout = stride, pad, dil, Cout
            ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = ldw, cin_total, w_choff, out.ld
            ar.w, ar.y = wptr, out.ptr
            ar.bias = self.param(bias) if bias is not None else None
            ar.bias_n = bias_n.ptr if bias_n is not None else None
            ar.stats = stats.ptr if stats is not None else None
            ar.stats_ld = stats_ld
            rs_out = None
            if fold:
                for i, s in enumerate(srcs):
                    assert (s.N, s.H, s.W) == (N, H, W), 'virtual concat needs equal spatial size'
                    ar.src[i] = self.src(s, relu_in, fold=True)
                if self.want_grad:                    # one copy-out slot: only src[0] may be a sampled map when a backward pass follows
                    if any(self._pending(s) for s in srcs[1:]):
                        fold = False
                    elif self._pending(srcs[0]):
                        ar.rs_y, ar.rs_ldy = srcs[0].raw.ptr, srcs[0].raw.ld
                fold = fold and int(lib.addk_conv_fwd_resample_ok(C.byref(ar))) == 1
                if not fold:
                    ar.rs_y, ar.rs_ldy = None, 0
            if not fold:
                for i, s in enumerate(srcs):
                    assert (s.N, s.H, s.W) == (N, H, W), 'virtual concat needs equal spatial size'
                    ar.src[i] = self.src(s, relu_in)  # (a still-deferred resize is emitted as its own launch here)
            rd_src = [r for s_ in srcs for r in self.lz(s_, fold=fold)]
            if fold and self.want_grad and self._pending(srcs[0]):
                rs_out = srcs[0].raw
                srcs[0].rs.done = True                # this launch writes the resized tensor
            wpk = None
            npk = 0 if fold else int(lib.addk_conv_fwd_pack_floats(C.byref(ar)))       # > 0: wide 3x3 stride-1 conv, halo-patch kernel (conv3.hip)
            if npk > 0:
                wpk = self.buf(npk)
                ar.wpack, ar.wpack_floats = wpk.ptr, npk
                self._hoist_pack(lib.addk_conv_fwd_pack_desc, ar, weight, wpk, create=True)
            self.keep.append(ar)
            cf = self._add(self.fwd, 'conv_fwd', lib.addk_conv_fwd, C.byref(ar),
                           rd=rd_src + [weight, bias, bias_n], wr=[out, stats, wpk, rs_out])
            bk = int(lib.addk_conv_fwd_batch_key(C.byref(ar)))
            if bk >= 0:
                cf.payload, cf.bkey = ar, bk
            self.meta.append(dict(kind='conv_fwd', cmd=cf, flops=2.0 * N * OH * OW * Cout * k * k * csum,
                                  bytes=4.0 * (N * H * W * csum + N * OH * OW * Cout + Cout * k * k * csum),
                                  shape=(N, H, W, csum, Cout, k, stride, dil), halo=npk > 0))
    
        if self.want_grad:
            srcs_l = list(srcs)

            def emit_bwd():
                if not self.grad_ready(out):
                    return
                dy = self.grad(out)
                P = N * OH * OW
                if bias is not None:
                    self._colsum(dy, bias, per_image=False)
                if bias_n is not None:
                    gbn = self.grad(bias_n)
                    self._colsum_n(dy, gbn)
                choff = w_choff
                for s in srcs_l:
                    # weight gradient
                    wa = L.ConvWgradArgs()
                    wa.dy, wa.lddy, wa.Cout = dy.ptr, dy.ld, Cout
                    wa.N, wa.H, wa.W, wa.OH, wa.OW, wa.KH, wa.KW, wa.stride, wa.pad, wa.dil = N, H, W, OH, OW, k, k, stride, pad, dil
                    wa.src = self.src(s, relu_in)
                    gp, acc = self.param_grad(weight, (choff, choff + s.C))
                    wa.dw, wa.ldw, wa.cin_total, wa.w_choff, wa.accumulate = gp, ldw, cin_total, choff, acc
                    wa.ws_floats = lib.addk_conv_wgrad_ws(P, Cout, s.C, k * k)
                    self._wgrads.append((wa, [dy] + self.lz(s), self.pgrad[weight], (id(weight), choff, choff + s.C)))
                    # data gradient
                    if s.needs_grad:
                        da = L.ConvDgradArgs()
                        da.dy, da.lddy, da.Cout = dy.ptr, dy.ld, Cout
                        da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, OH, OW, k, k, stride, pad, dil
                        da.w, da.ldw, da.cin_total, da.w_choff = wptr, ldw, cin_total, choff
                        da.dst = self.src(s, relu_in)
                        gs = self.grad(s.raw)
                        da.g, da.ldg, da.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                        rows = lib.addk_conv_rows(N * H * W, s.C)
                        slab = self._dab(s, rows)
                        if slab is not None:
                            da.dab = slab.ptr
                            s.bn.slabs.append((slab, rows))
                        dpk = None
                        npk = int(lib.addk_conv_dgrad_pack_floats(C.byref(da)))
                        if npk > 0:
                            dpk = self.buf(npk)
                            da.wpack, da.wpack_floats = dpk.ptr, npk
                            self._hoist_pack(lib.addk_conv_dgrad_pack_desc, da, weight, dpk, create=False)
                        self.keep.append(da)
                        cd = self._add(self.bwd, 'conv_dgrad', lib.addk_conv_dgrad, C.byref(da), rd=[dy, weight] + self.lz(s),
                                       wr=[gs, slab, dpk])
                        bk = int(lib.addk_conv_dgrad_batch_key(C.byref(da)))
                        if bk >= 0:
                            cd.payload, cd.bkey = da, bk
                    choff += s.C
            self._bwd_emitters.append(emit_bwd)
        return out

    def _colsum(self, dy, bias, per_image):
        """bias gradient: db[c] = sum_p dy[p,c]  (via the GAP kernel over all N*H*W pixels as one 'image')."""
        lib = self.lib
        s = L.Src(); s.x, s.ld, s.C, s.relu = dy.ptr, dy.ld, dy.C, 0
        tmp = self.vec(dy.C)
        rows = lib.addk_ew_rows(dy.P, dy.C)
        ws = self.buf(rows * dy.C)
        self.keep.append(s)
        self._add(self.bwd, 'bias_grad', lib.addk_gap_fwd, C.byref(s), 1, dy.P, tmp.ptr, dy.C, ws.ptr, 0, rd=[dy], wr=[tmp, ws])
        gp, acc = self.param_grad(bias)
        one = L.AffineSumArgs()
        t = L.Src(); t.x, t.ld, t.C = tmp.ptr, dy.C, dy.C
        one.term[0] = t; one.nterm = 1; one.P = 1; one.C = dy.C
        one.out, one.ldo, one.relu_out, one.accumulate = gp, dy.C, 0, acc
        self.keep.append(one)
        self._add(self.bwd, 'bias_grad_acc', lib.addk_affine_sum_fwd, C.byref(one), rd=[tmp], wr=[self.pgrad[bias]])

    def _colsum_n(self, dy, gbn):
        """per-image bias gradient: g[n,c] = sum_{p in image n} dy[p,c]."""
        lib = self.lib
        s = L.Src(); s.x, s.ld, s.C, s.relu = dy.ptr, dy.ld, dy.C, 0
        rows = lib.addk_ew_rows(dy.H * dy.W, dy.C)
        ws = self.buf(dy.N * rows * dy.C)
        assert self.acc_flag(gbn) == 0
        self.keep.append(s)
        self._add(self.bwd, 'bias_n_grad', lib.addk_gap_fwd, C.byref(s), dy.N, dy.H * dy.W, gbn.ptr, gbn.ld, ws.ptr, 0,
                  rd=[dy], wr=[gbn, ws])

    def stats_slab(self, P, Cc):
        rows = self.lib.addk_conv_rows(P, Cc)
        return self.buf(rows * Cc * 4), rows      # fp64 [rows][C][2]

    def bn(self, raw, mod, slab=None, rows=0, post_relu=False, needs_grad=True, fuse=None):
        """Apply BatchNorm module `mod` lazily to `raw`.  Training: statistics come from `slab`.  `fuse` = (args, cmd) of the
        producing launch when its kernel can finalize the statistics itself (last workgroup, csrc/bnfin.h): the finalize
        arguments go into `args.fin` and no bn_finalize launch is emitted (local BatchNorm only: the SyncBN exchange sits
        between the slab and the finalize)."""
        lib = self.lib
        Cc = raw.C
        a, b = self.vec(Cc), self.vec(Cc)
        training = self.training and mod.training
        count = float(raw.P)
        if TRACE_BN is not None:
            TRACE_BN.append((mod, raw))
        st = BNState(mod, Cc, a, b, training, count)
        gam = self.param(mod.weight) if mod.weight is not None else None
        bet = self.param(mod.bias) if mod.bias is not None else None
        if training:
            st.mean, st.invstd = self.vec(Cc), self.vec(Cc)
            sync = self.world is not None and getattr(mod, 'sync', False) and (self.world.size > 1 or self.world.force)
            fa = L.BnFinalizeArgs()
            if sync:
                red = LateVec(4 * Cc, f64=True)      # fp64 [C][2], summed over ranks
                sri = L.SlabReduceItem()
                sri.partial, sri.rows, sri.C = slab.ptr, rows, Cc
                csr = self._add(self.fwd, 'slab_reduce', lib.addk_slab_reduce, slab.ptr, rows, Cc, None, rd=[slab], wr=[red])
                csr.payload = sri
                red.binders += [lambda p, sri=sri: setattr(sri, 'out', p), lambda p, csr=csr: csr.args.__setitem__(3, p),
                                lambda p: setattr(fa, 'partial', p)]
                self.world.emit_allreduce(self, self.fwd, red)    # every rank has the same per-rank count
                fa.rows = 1
                st.count = count * self.world.size
            else:
                fa.partial, fa.rows = slab.ptr, rows
            fa.C, fa.count = Cc, st.count
            fa.gamma, fa.beta = gam, bet
            if mod.track_running_stats and mod.running_mean is not None:
                fa.running_mean, fa.running_var = mod.running_mean.data_ptr(), mod.running_var.data_ptr()
                if mod.num_batches_tracked is not None:
                    self.nbt[mod] = self.nbt.get(mod, 0) + 1
            fa.momentum = 0.1 if mod.momentum is None else mod.momentum
            fa.eps = mod.eps
            fa.a, fa.b, fa.mean, fa.invstd = a.ptr, b.ptr, st.mean.ptr, st.invstd.ptr
            self.keep.append(fa)
            stat_wr = [a, b, st.mean, st.invstd] + ([mod.running_mean, mod.running_var] if fa.running_mean else [])
            if fuse is not None and not sync and os.environ.get('ADDK_FUSE_FINALIZE', '0') == '1':
                args, cmd, fuse_blocks = fuse
                for f_ in ('count', 'gamma', 'beta', 'running_mean', 'running_var', 'momentum', 'eps', 'a', 'b', 'mean', 'invstd'):
                    setattr(args.fin, f_, getattr(fa, f_))
                nblk, sld = fuse_blocks(args)
                ctr = self.buf((int(lib.addk_bn_fin_ws_bytes(nblk, sld)) + 3) // 4, zero=True)   # ticket counters + group rows of this BatchNorm call
                args.fin_counter = ctr.ptr
                cmd.rd += [r for r in (_region(mod.weight), _region(mod.bias)) if r]
                cmd.wr += [r for r in (_region(x) for x in stat_wr + [ctr]) if r]
            else:
                cfin = self._add(self.fwd, 'bn_finalize', lib.addk_bn_finalize, C.byref(fa),
                                 rd=[red if sync else slab, mod.weight, mod.bias], wr=stat_wr)
                cfin.payload = fa
        else:
            # inference: (a, b) of ALL BatchNorms come from one batched launch placed where the first one is emitted
            e = L.BnEvalEntry()
            e.gamma, e.beta, e.rm, e.rv = gam, bet, mod.running_mean.data_ptr(), mod.running_var.data_ptr()
            e.a, e.b, e.C, e.eps = a.ptr, b.ptr, Cc, mod.eps
            if not self._evalbn:
                self._evalbn_cmd = self._add(self.fwd, 'bn_eval_affine_batch', lib.addk_bn_eval_affine_batch, None, 0)
            self._evalbn.append(e)
            self._evalbn_cmd.rd += [_region(t) for t in (mod.weight, mod.bias, mod.running_mean, mod.running_var) if t is not None]
            self._evalbn_cmd.wr += [_region(a), _region(b)]
        act = Act(raw, st, post_relu, needs_grad and self.want_grad)

        if self.want_grad and training:
            def emit_bwd():
                if not self.grad_ready(raw) and not st.slabs:
                    return
                if not self.grad_ready(raw):
                    raise RuntimeError('BN output received (dA,dB) but no direct gradient')
                g = self.grad(raw)
                ba = L.BnBwdArgs()
                assert len(st.slabs) <= L.MAX_SLAB, 'too many consumers of one BatchNorm output (%d)' % len(st.slabs)
                for i, (p, r) in enumerate(st.slabs):
                    ba.slab[i], ba.rows[i] = p.ptr, r
                ba.nslab, ba.C, ba.count = len(st.slabs), Cc, st.count
                # dy_raw = G + c1 + c2*(x - mean): the mean is subtracted first, as the reference does (batchnorm.py:51-53), instead
                # of folding -c2*mean into c1 where it cancels against c2*x (ADDK_BN_CENTERED=0 restores the folded form)
                centered = os.environ.get('ADDK_BN_CENTERED', '1') == '1'
                ba.centered = int(centered)
                mu = st.mean if centered else None
                ba.gamma, ba.mean, ba.invstd, ba.a = gam, st.mean.ptr, st.invstd.ptr, a.ptr
                if mod.weight is not None:
                    gp, acc = self.param_grad(mod.weight)
                    bp, acc2 = self.param_grad(mod.bias)
                    assert acc == acc2
                    ba.dgamma, ba.dbeta, ba.accumulate = gp, bp, acc
                    pg = [self.pgrad[mod.weight], self.pgrad[mod.bias]]
                else:
                    pg = []
                rd_bn = [sl for sl, _ in st.slabs] + [mod.weight, st.mean, st.invstd, a]
                c1, c2 = self.vec(Cc), self.vec(Cc)
                sync = self.world is not None and getattr(mod, 'sync', False) and (self.world.size > 1 or self.world.force)
                self.keep.append(ba)
                if sync:
                    dmv = LateVec(2 * Cc)              # (dmean, dvar) contributions of this rank, summed over ranks
                    self._add(self.bwd, 'bn_bwd', lib.addk_bn_bwd, C.byref(ba), rd=rd_bn, wr=pg + [dmv]).payload = ba
                    self.world.emit_allreduce(self, self.bwd, dmv)
                    cit = L.BnCoeffsItem()
                    cit.c1, cit.c2, cit.count, cit.C = c1.ptr, c2.ptr, st.count, Cc
                    cco = self._add(self.bwd, 'bn_bwd_coeffs', lib.addk_bn_bwd_coeffs_from_dmv, None, Cc, st.count, c1.ptr, c2.ptr,
                                    rd=[dmv], wr=[c1, c2])
                    cco.payload = cit
                    dmv.binders += [lambda p, ba=ba: setattr(ba, 'dmv', p), lambda p, cit=cit: setattr(cit, 'dmv', p),
                                    lambda p, cco=cco: cco.args.__setitem__(0, p)]
                else:
                    ba.c1, ba.c2 = c1.ptr, c2.ptr
                    self._add(self.bwd, 'bn_bwd', lib.addk_bn_bwd, C.byref(ba), rd=rd_bn, wr=pg + [c1, c2]).payload = ba
                # dy_raw = G + c1 + c2*x, in place on the accumulated gradient
                cap = self._add(self.bwd, 'bn_bwd_apply', lib.addk_bn_bwd_apply, g.ptr, g.ld, raw.ptr, raw.ld, mu.ptr if mu else None,
                                c1.ptr, c2.ptr, raw.P, Cc, g.ptr, g.ld, rd=[g, raw, c1, c2, mu], wr=[g])
                if Cc % 4 == 0 and Cc <= 1024 and g.ld % 4 == 0 and raw.ld % 4 == 0 and g.ptr % 16 == 0 and raw.ptr % 16 == 0:
                    it = L.BnApplyItem()
                    it.g, it.x, it.c1, it.c2, it.out, it.P = g.ptr, raw.ptr, c1.ptr, c2.ptr, g.ptr, raw.P
                    it.mean = mu.ptr if mu else None
                    it.ldg, it.ldx, it.ldo, it.C = g.ld, raw.ld, g.ld, Cc
                    cap.payload = it
            self._bwd_emitters.append(emit_bwd)
        return act

    def conv_bn(self, srcs, conv_mod, bn_mod, relu_in, post_relu=False, out=None, **kw):
        """ReLU? -> Conv -> BN (lazy).  conv_mod/bn_mod are nn.Conv2d / nn.BatchNorm2d parameter holders."""
        k = conv_mod.kernel_size[0]
        a0 = srcs[0]
        stride, pad, dil = conv_mod.stride[0], conv_mod.padding[0], conv_mod.dilation[0]
        OH = (a0.H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        OW = (a0.W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Cout = conv_mod.out_channels
        slab = rows = None
        training = self.training and bn_mod.training
        if training:
            slab, rows = self.stats_slab(a0.N * OH * OW, Cout)
        raw = self.conv(srcs, conv_mod.weight, Cout, k, stride, pad, dil, relu_in, out=out, stats=slab, **kw)
        return self.bn(raw, bn_mod, slab, rows or 0, post_relu)

    def dwconv(self, src, conv_mod, relu_in, fwd=True):
        """Depthwise conv; returns a materialised Act.  fwd=False: only the output buffer and the backward pass (sep_half)."""
        lib = self.lib
        k = conv_mod.kernel_size[0]
        stride, pad, dil = conv_mod.stride[0], conv_mod.padding[0], conv_mod.dilation[0]
        N, H, W, Cc = src.N, src.H, src.W, src.C
        OH = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = self.tensor(N, OH, OW, Cc)
        wptr = self.param(conv_mod.weight)
        ar = L.DwArgs()
        ar.src = self.src(src, relu_in)
        ar.N, ar.H, ar.W, ar.OH, ar.OW, ar.KH, ar.KW, ar.stride, ar.pad, ar.dil = N, H, W, OH, OW, k, k, stride, pad, dil
        ar.w, ar.y, ar.ldy = wptr, out.ptr, out.ld
        self.keep.append(ar)
        if fwd:
            cdw = self._add(self.fwd, 'dw_fwd', lib.addk_dw_fwd, C.byref(ar), rd=self.lz(src) + [conv_mod.weight], wr=[out])
            bk = int(lib.addk_dw_fwd_batch_key(C.byref(ar)))
            if bk >= 0:
                cdw.payload, cdw.bkey = ar, bk
        act = Act(out, None, False, self.want_grad)
        if self.want_grad:
            def emit_bwd():
                if not self.grad_ready(out):
                    return
                dy = self.grad(out)
                ba = L.DwBwdArgs()
                ba.dy, ba.lddy = dy.ptr, dy.ld
                ba.N, ba.H, ba.W, ba.OH, ba.OW, ba.KH, ba.KW, ba.stride, ba.pad, ba.dil = N, H, W, OH, OW, k, k, stride, pad, dil
                ba.src = self.src(src, relu_in)
                ba.w = wptr
                rows = lib.addk_dw_rows(N * H * W, Cc)
                if src.needs_grad:
                    gs = self.grad(src.raw)
                    ba.g, ba.ldg, ba.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                    slab = self._dab(src, rows)
                    if slab is not None:
                        ba.dab = slab.ptr
                        src.bn.slabs.append((slab, rows))
                else:
                    gs = slab = None
                gp, acc = self.param_grad(conv_mod.weight)
                ba.dw, ba.dw_accumulate = gp, acc
                ws = self.buf(rows * Cc * k * k)
                ba.ws = ws.ptr
                ba.defer_wreduce = 1          # the [rows][C][k*k] partials are reduced at the end of the backward pass, all convs in one launch
                it = L.DwWreduceItem()
                it.ws, it.dw, it.rows, it.n, it.accumulate = ws.ptr, gp, rows, Cc * k * k, acc
                self._dwreds.append((it, ws, self.pgrad[conv_mod.weight]))
                self.keep.append(ba)
                cdb = self._add(self.bwd, 'dw_bwd', lib.addk_dw_bwd, C.byref(ba), rd=[dy, conv_mod.weight] + self.lz(src),
                                wr=[gs, slab, ws])
                bk = int(lib.addk_dw_bwd_batch_key(C.byref(ba)))
                if bk >= 0:
                    cdb.payload, cdb.bkey = ba, bk
            self._bwd_emitters.append(emit_bwd)
        return act

    def _sep_bwd(self, src, dw_mod, pw_mod, raw, t=None, probe=False):
        """Backward of one SepConv half as ONE launch (addk_sep_bwd, csrc/sepb.hip): data gradient of the pointwise conv and the
        whole depthwise backward, the gradient between them staying on chip; the pointwise WEIGHT gradient joins the deferred
        weight-gradient batches.  probe=True only asks whether the kernel covers the shape."""
        lib = self.lib
        k = dw_mod.kernel_size[0]
        N, H, W, Cc = src.N, src.H, src.W, src.C
        Cout = pw_mod.out_channels

        def fill(ba, dy_ptr, dy_ld):
            ba.dy, ba.lddy = dy_ptr, dy_ld
            ba.N, ba.H, ba.W, ba.K = N, H, W, k
            ba.src = self.src(src, True)
            ba.Cout, ba.ldw = Cout, Cc
            ba.dw_w, ba.pw_w = self.param(dw_mod.weight), self.param(pw_mod.weight)
        def fill_w(wa, dy_ptr, dy_ld, t_src, gp, acc):
            wa.dy, wa.lddy, wa.Cout = dy_ptr, dy_ld, Cout
            wa.N, wa.H, wa.W, wa.OH, wa.OW, wa.KH, wa.KW, wa.stride, wa.pad, wa.dil = N, H, W, H, W, 1, 1, 1, 0, 1
            wa.src = t_src
            wa.dw, wa.ldw, wa.cin_total, wa.w_choff, wa.accumulate = gp, Cc, Cc, 0, acc
        if probe:
            ba = L.SepBwdArgs()
            fill(ba, raw.ptr, raw.ld)
            ok = int(lib.addk_sep_bwd_rows(C.byref(ba))) > 0
            return ok

        def emit_bwd():
            if not self.grad_ready(raw):
                return
            dy = self.grad(raw)
            # pointwise weight gradient: dW[co][ci] = sum_p dy[p][co] t[p][ci]
            wa = L.ConvWgradArgs()
            gp, acc = self.param_grad(pw_mod.weight, (0, Cc))
            fill_w(wa, dy.ptr, dy.ld, self.src(t), gp, acc)
            wa.ws_floats = lib.addk_conv_wgrad_ws(N * H * W, Cout, Cc, 1)
            self._wgrads.append((wa, [dy] + self.lz(t), self.pgrad[pw_mod.weight], (id(pw_mod.weight), 0, Cc)))
            # fused data gradient + depthwise backward
            ba = L.SepBwdArgs()
            fill(ba, dy.ptr, dy.ld)
            rows = int(lib.addk_sep_bwd_rows(C.byref(ba)))
            gs = slab = None
            if src.needs_grad:
                gs = self.grad(src.raw)
                ba.g, ba.ldg, ba.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                slab = self._dab(src, rows)
                if slab is not None:
                    ba.dab = slab.ptr
                    src.bn.slabs.append((slab, rows))
            gpw, accw = self.param_grad(dw_mod.weight)
            ws = self.buf(rows * Cc * k * k)
            ba.ws = ws.ptr
            it = L.DwWreduceItem()
            it.ws, it.dw, it.rows, it.n, it.accumulate = ws.ptr, gpw, rows, Cc * k * k, accw
            self._dwreds.append((it, ws, self.pgrad[dw_mod.weight]))
            self.keep.append(ba)
            cb = self._add(self.bwd, 'sep_bwd', lib.addk_sep_bwd, C.byref(ba), rd=[dy, dw_mod.weight, pw_mod.weight] + self.lz(src),
                           wr=[gs, slab, ws])
            bk = int(lib.addk_sep_bwd_batch_key(C.byref(ba)))
            if bk >= 0:
                cb.payload, cb.bkey = ba, bk
        self._bwd_emitters.append(emit_bwd)
        return True

    def sep_half(self, src, dw_mod, pw_mod, bn_mod, sum_terms=None, out=None):
        """One half of SepConv (operations.py:51-54 / 55-58): ReLU -> depthwise k x k -> pointwise 1x1 -> BN (lazy), as ONE
        launch where the fused kernel covers the shape (addk_sep_fwd, csrc/sepf.hip: the depthwise output stays on chip; in
        training it is also written out because the backward pass reads it, and the launch's last workgroup finalizes the
        BatchNorm statistics itself).  Inference only: `sum_terms` (other branches of the cell block) makes the epilogue apply
        this op's frozen BatchNorm and write the block sum (ADD.py:108) into `out`."""
        lib = self.lib
        k = dw_mod.kernel_size[0]
        N, H, W, Cc = src.N, src.H, src.W, src.C
        Cout = pw_mod.out_channels
        training = self.training and bn_mod.training
        ar = L.SepArgs()
        ar.src = self.src(src, True)
        ar.N, ar.H, ar.W, ar.K, ar.Cout, ar.ldw = N, H, W, k, Cout, Cc
        ar.dw_w, ar.pw_w = self.param(dw_mod.weight), self.param(pw_mod.weight)
        fused = (os.environ.get('ADDK_FUSE_SEP', '1') == '1' and dw_mod.stride[0] == 1 and dw_mod.dilation[0] == 1
                 and dw_mod.padding[0] == k // 2 and pw_mod.kernel_size[0] == 1 and (sum_terms is None or not (training or self.want_grad)))
        if fused:
            raw = out if (out is not None and sum_terms is not None) else self.tensor(N, H, W, Cout)
            ar.y, ar.ldy = raw.ptr, raw.ld
            fused = bool(lib.addk_sep_fwd_supported(C.byref(ar)))
        if not fused:
            t = self.dwconv(src, dw_mod, relu_in=True)
            act = self.conv_bn([t], pw_mod, bn_mod, relu_in=False)
            return act if sum_terms is None else self.affine_sum(list(sum_terms) + [act], out=out)
        t = None
        slab = rows = None
        if training:
            rows = max(int(lib.addk_sep_rows(C.byref(ar))), int(lib.addk_conv_rows(N * H * W, Cout)))
            slab = self.buf(rows * Cout * 4)
            ar.stats_rows = rows
        if self.want_grad:
            if os.environ.get('ADDK_FUSE_SEP_BWD', '1') == '1' and self._sep_bwd(src, dw_mod, pw_mod, raw, probe=True):
                t = Act(self.tensor(N, H, W, Cc), None, False, True)           # depthwise output: written by the fused forward, read by the pointwise weight gradient
                self._sep_bwd(src, dw_mod, pw_mod, raw, t=t)
            else:
                t = self.dwconv(src, dw_mod, relu_in=True, fwd=False)
                self.conv([t], pw_mod.weight, Cout, 1, out=raw, stats=slab, fwd=False)       # backward of the pointwise half
            ar.t, ar.ldt = t.raw.ptr, t.raw.ld
        ar.stats = slab.ptr if slab is not None else None
        ar.stats_ld = 0
        rd = self.lz(src) + [dw_mod.weight, pw_mod.weight]
        if sum_terms is not None:
            # inference: y = a*acc + b (this op's frozen BatchNorm) + the other branches, written straight into the block's slot
            st = self.bn(raw, bn_mod, None, 0).bn
            terms = [tm for tm in sum_terms if not tm.zero]
            assert len(terms) <= L.MAX_TERMS and out is not None
            ar.ea, ar.eb, ar.nterm = st.a.ptr, st.b.ptr, len(terms)
            for i, tm in enumerate(terms):
                assert (tm.N, tm.H, tm.W, tm.C) == (N, H, W, Cout), 'branch shapes differ'
                ar.term[i] = self.src(tm)
                rd += self.lz(tm)
            rd += [st.a, st.b]
        self.keep.append(ar)
        c = self._add(self.fwd, 'sep_fwd', lib.addk_sep_fwd, C.byref(ar), rd=rd, wr=[raw, slab, t.raw if t is not None else None])
        if sum_terms is not None:
            bk = int(lib.addk_sep_fwd_batch_key(C.byref(ar)))
            if bk >= 0:
                c.payload, c.bkey = ar, bk
            return Act(raw, None, False, False)
        act = self.bn(raw, bn_mod, slab, rows or 0, fuse=(ar, c, lambda a_: (int(lib.addk_sep_rows(C.byref(a_))), Cout)))
        bk = int(lib.addk_sep_fwd_batch_key(C.byref(ar)))         # after bn(): the fused finalize is part of the launch's validity
        if bk >= 0:
            c.payload, c.bkey = ar, bk
        return act

    def affine_sum(self, terms, out=None, relu_out=False):
        """Materialise sum_i relu_i?(a_i*x_i+b_i) (optionally ReLU'd) into `out`."""
        lib = self.lib
        terms = [t for t in terms if not t.zero]
        t0 = terms[0]
        if out is None:
            out = self.tensor(t0.N, t0.H, t0.W, t0.C)
        assert len(terms) <= L.MAX_TERMS
        ar = L.AffineSumArgs()
        for i, t in enumerate(terms):
            assert (t.N, t.H, t.W, t.C) == (out.N, out.H, out.W, out.C), 'branch shapes differ'
            ar.term[i] = self.src(t)
        ar.nterm, ar.P, ar.C = len(terms), out.P, out.C
        ar.out, ar.ldo, ar.relu_out, ar.accumulate = out.ptr, out.ld, int(relu_out), 0
        self.keep.append(ar)
        self._add(self.fwd, 'affine_sum', lib.addk_affine_sum_fwd, C.byref(ar), rd=[r for t in terms for r in self.lz(t)], wr=[out])
        ng = self.want_grad and any(t.needs_grad for t in terms)
        act = Act(out, None, False, ng)
        if ng:
            def emit_bwd():
                if not self.grad_ready(out):
                    return
                do = self.grad(out)
                ba = L.AffineSumBwdArgs()
                rows = lib.addk_ew_rows(out.P, out.C)
                wr_ = []
                for i, t in enumerate(terms):
                    ba.term[i] = self.src(t)
                    if t.needs_grad:
                        gt = self.grad(t.raw)
                        ba.g[i], ba.ldg[i], ba.accumulate[i] = gt.ptr, gt.ld, self.acc_flag(gt)
                        slab = self._dab(t, rows)
                        if slab is not None:
                            ba.dab[i] = slab.ptr
                            t.bn.slabs.append((slab, rows))
                        wr_ += [gt, slab]
                ba.nterm, ba.P, ba.C = len(terms), out.P, out.C
                ba.dout, ba.lddo = do.ptr, do.ld
                ba.out, ba.ldo, ba.relu_out = out.ptr, out.ld, int(relu_out)
                self.keep.append(ba)
                self._add(self.bwd, 'affine_sum_bwd', lib.addk_affine_sum_bwd, C.byref(ba),
                          rd=[do, out] + [r for t in terms for r in self.lz(t)], wr=wr_)
            self._bwd_emitters.append(emit_bwd)
        return act

    def materialize(self, act, relu_in=False):
        if act.bn is None and not (act.relu or relu_in):
            return act
        a = Act(act.raw, act.bn, act.relu or relu_in, act.needs_grad, rs=act.rs)
        return self.affine_sum([a])

    def resize(self, src, OH, OW, relu_in=False):
        """Bilinear resize.  A lazy BN without pending ReLU passes through (affine commutes with
        interpolation); a pending ReLU forces materialisation."""
        lib = self.lib
        relu = bool(src.relu or relu_in)
        N, H, W, Cc = src.N, src.H, src.W, src.C
        out = self.tensor(N, OH, OW, Cc)
        ar = L.ResizeArgs()
        carrier = src if relu else Act(src.raw, None, False, src.needs_grad, rs=src.rs)
        ar.src = self.src(carrier, relu_in)
        ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, OH, OW
        ar.y, ar.ldy, ar.nchw_out = out.ptr, out.ld, 0
        self.keep.append(ar)
        tag = self.tag

        def emit_fwd():
            c = self._add(self.fwd, 'resize_fwd', lib.addk_resize_fwd, C.byref(ar), rd=self.lz(carrier), wr=[out])
            c.tag = tag
            if rs is not None:
                rs.done = True
        rs = None
        if relu or self._pending(src) or src.raw.H >= (1 << 15) or src.raw.W >= (1 << 16):
            emit_fwd()                    # the interpolated values are relu(a x + b): not an affine of the interpolated raw map
        else:
            rs = Resample(src.raw, emit_fwd)      # deferred: a 1x1 consumer samples src.raw itself (Graph.conv), anyone else emits it
        res = Act(out, None if relu else src.bn, False, src.needs_grad, rs=rs)
        if self.want_grad and src.needs_grad:
            def emit_bwd():
                if not self.grad_ready(out):
                    return
                dy = self.grad(out)
                ba = L.ResizeBwdArgs()
                ba.dy, ba.lddy, ba.nchw_in = dy.ptr, dy.ld, 0
                ba.src = self.src(carrier, relu_in)
                ba.N, ba.H, ba.W, ba.OH, ba.OW = N, H, W, OH, OW
                gs = self.grad(src.raw)
                ba.g, ba.ldg, ba.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                slab = None
                if relu:
                    rows = lib.addk_ew_rows(N * H * W, Cc)
                    slab = self._dab(src, rows)
                    if slab is not None:
                        ba.dab = slab.ptr
                        src.bn.slabs.append((slab, rows))
                self.keep.append(ba)
                crb = self._add(self.bwd, 'resize_bwd', lib.addk_resize_bwd, C.byref(ba), rd=[dy] + self.lz(src), wr=[gs, slab])
                bk = int(lib.addk_resize_bwd_batch_key(C.byref(ba))) if os.environ.get('ADDK_BATCH_RESIZE_BWD', '1') == '1' else -1
                if bk >= 0:
                    crb.payload, crb.bkey = ba, bk
            self._bwd_emitters.append(emit_bwd)
        return res

    def resize_to_nchw(self, src, OH, OW):
        """Final logits resize (decoder.py:28) into a contiguous [N,C,OH,OW] tensor.  Returns an OutRef.
        In a fused training step (`self.fuse_ce`, set by train.TrainStep) nobody reads the full-resolution logits: the
        resize is not emitted at all and the OutRef carries what the fused up-sampling + cross-entropy launch needs
        (`addk_ce_upsample_fwd_bwd`, placed at the head of the backward list with the loss parameters TrainStep binds)."""
        lib = self.lib
        assert src.bn is None and not src.relu
        N, H, W, Cc = src.N, src.H, src.W, src.C
        if (getattr(self, 'fuse_ce', False) and self.want_grad and src.needs_grad
                and lib.addk_ce_upsample_supported(N, H, W, OH, OW, Cc) == 1):
            out = OutRef(None)
            out.fused_ce, out.shape, out.ce = True, (N, Cc, OH, OW), None

            def emit_ce():
                assert out.ce is not None, 'fused logits output without a loss binding'
                a = L.CeUpsampleArgs()
                s = self.src(src)
                a.logits, a.ld = s.x, s.ld
                a.N, a.H, a.W, a.C, a.OH, a.OW = N, H, W, Cc, OH, OW
                ce = out.ce
                a.target, a.class_w, a.ignore_index = ce['target'].data_ptr(), ce['class_w'], ce['ignore_index']
                a.wsum, a.scale, a.loss_out = ce['wsum'].data_ptr(), ce['scale'], ce['loss'].data_ptr()
                gs = self.grad(src.raw)
                a.g, a.ldg, a.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                ws = torch.zeros(int(lib.addk_ce_upsample_ws_floats(N, H, W)), dtype=torch.float32, device=self.device)
                a.ws = ws.data_ptr()
                self.keep += [a, ws]
                self._add(self.bwd, 'ce_upsample', lib.addk_ce_upsample_fwd_bwd, C.byref(a),
                          rd=self.lz(src) + [ce['target'], ce['wsum']], wr=[gs, ce['loss'], ws])
            self._bwd_emitters.append(emit_ce)
            return out
        y = torch.empty((N, Cc, OH, OW), dtype=torch.float32, device=self.device)
        self.nbytes += y.numel() * 4
        ar = L.ResizeArgs()
        ar.src = self.src(src)
        ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, OH, OW
        ar.y, ar.ldy, ar.nchw_out = y.data_ptr(), 0, 1
        self.keep.append(ar)
        self._add(self.fwd, 'resize_nchw', lib.addk_resize_fwd, C.byref(ar), rd=self.lz(src), wr=[y])
        out = OutRef(y)
        if self.want_grad and src.needs_grad:
            def emit_bwd():
                if out.dy_ptr is None and not out.dynamic:
                    return
                ba = L.ResizeBwdArgs()
                ba.dy = out.dy_ptr
                ba.lddy, ba.nchw_in = 0, 1
                ba.dy_scale = out.dy_scale
                ba.src = self.src(src)
                ba.N, ba.H, ba.W, ba.OH, ba.OW = N, H, W, OH, OW
                gs = self.grad(src.raw)
                ba.g, ba.ldg, ba.accumulate = gs.ptr, gs.ld, self.acc_flag(gs)
                self.keep.append(ba)
                out.bwd_args = ba
                self._add(self.bwd, 'resize_nchw_bwd', lib.addk_resize_bwd, C.byref(ba), rd=self.lz(src), wr=[gs])
            self._bwd_emitters.append(emit_bwd)
        return out

    def gap(self, src, relu_in=False):
        lib = self.lib
        N, H, W, Cc = src.N, src.H, src.W, src.C
        out = self.tensor(N, 1, 1, Cc)
        s = self.src(src, relu_in)
        rows = lib.addk_ew_rows(H * W, Cc)
        ws = self.buf(N * rows * Cc)
        self.keep.append(s)
        self._add(self.fwd, 'gap_fwd', lib.addk_gap_fwd, C.byref(s), N, H * W, out.ptr, out.ld, ws.ptr, 1, rd=self.lz(src), wr=[out, ws])
        act = Act(out, None, False, self.want_grad and src.needs_grad)
        if act.needs_grad:
            def emit_bwd():
                if not self.grad_ready(out):
                    return
                dy = self.grad(out)
                gs = self.grad(src.raw)
                acc = self.acc_flag(gs)
                rows2 = lib.addk_ew_rows(N * H * W, Cc)
                slab = self._dab(src, rows2)
                if slab is not None:
                    src.bn.slabs.append((slab, rows2))
                self._add(self.bwd, 'gap_bwd', lib.addk_gap_bwd, C.byref(s), N, H * W, dy.ptr, dy.ld, gs.ptr, gs.ld, acc,
                          slab.ptr if slab is not None else None, rd=[dy] + self.lz(src), wr=[gs, slab])
            self._bwd_emitters.append(emit_bwd)
        return act

    def edm_head(self, src, conv_w, lins, host_out=None):
        """The whole Earlier-Decision-Maker head (ADD.py:502-525) as ONE launch (csrc/edm.hip): in-place ReLU, conv 3x3 stride 2 -> 128,
        ReLU, global average pool, the three Linear layers.  Inference only.  `host_out`: pinned host tensor the confidence is also
        written to (the gate of dynamic_inference reads it there: no device-to-host copy).  Returns the [N,1,1,1] Act, or None when the
        kernel does not cover the arguments (the caller then emits the generic launches)."""
        lib = self.lib
        if self.want_grad or os.environ.get('ADDK_FUSE_EDM', '1') != '1':
            return None
        N, H, W = src.N, src.H, src.W
        ar = L.EdmArgs()
        ar.src = self.src(src, True)
        ar.N, ar.H, ar.W = N, H, W
        ar.conv_w = self.param(conv_w)
        (l1, l2, l3) = lins
        ar.w1, ar.b1, ar.w2, ar.b2, ar.w3, ar.b3 = (self.param(l1.weight), self.param(l1.bias), self.param(l2.weight), self.param(l2.bias),
                                                    self.param(l3.weight), self.param(l3.bias))
        out = self.tensor(N, 1, 1, 1)
        ws = self.buf((int(lib.addk_edm_head_ws_bytes(N, H, W)) + 3) // 4, zero=True)
        ar.out, ar.ldo, ar.ws = out.ptr, out.ld, ws.ptr
        ar.out_host = host_out.data_ptr() if host_out is not None else None
        shapes_ok = (tuple(conv_w.shape[:1]) == (128,) and conv_w.numel() == 128 * 9 * src.C and tuple(l1.weight.shape) == (64, 128)
                     and tuple(l2.weight.shape) == (32, 64) and tuple(l3.weight.shape) == (1, 32))
        if not shapes_ok or int(lib.addk_edm_head_supported(C.byref(ar))) != 1:
            return None
        self.keep += [ar, host_out]
        self._add(self.fwd, 'edm_head', lib.addk_edm_head, C.byref(ar),
                  rd=self.lz(src) + [conv_w, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias], wr=[out, ws])
        return Act(out, None, False, False)

    def pool3(self, src, stride, mode):
        lib = self.lib
        N, H, W, Cc = src.N, src.H, src.W, src.C
        OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = self.tensor(N, OH, OW, Cc)
        s = self.src(src)
        self.keep.append(s)
        self._add(self.fwd, 'pool3_fwd', lib.addk_pool3_fwd, C.byref(s), N, H, W, OH, OW, stride, mode, out.ptr, out.ld,
                  rd=self.lz(src), wr=[out])
        act = Act(out, None, False, self.want_grad and src.needs_grad)
        if act.needs_grad:
            assert src.bn is None or not src.bn.training, 'pool3 backward through a lazy training BN is not built (cold path)'

            def emit_bwd():
                if not self.grad_ready(out):
                    return
                dy = self.grad(out)
                gs = self.grad(src.raw)
                self._add(self.bwd, 'pool3_bwd', lib.addk_pool3_bwd, C.byref(s), N, H, W, OH, OW, stride, mode, dy.ptr, dy.ld,
                          gs.ptr, gs.ld, self.acc_flag(gs), rd=[dy] + self.lz(src), wr=[gs])
            self._bwd_emitters.append(emit_bwd)
        return act

    def zeros(self, N, H, W, Cc):
        t = TRef(self.buf(N * H * W * Cc, zero=True), 0, N, H, W, Cc, Cc)
        return Act(t, None, False, False, zero=True)

    def input_nchw(self, x, requires_grad=False, ld=None):
        """Stage an NCHW torch tensor into the plan.  Returns (Act, InRef)."""
        lib = self.lib
        N, Cc, H, W = x.shape
        t = self.tensor(N, H, W, Cc, ld)
        inref = InRef(tuple(x.shape))
        self._add(self.fwd, 'nchw_to_nhwc', _in_stage, lib, inref, N, Cc, H * W, t.ptr, t.ld, wr=[t])
        act = Act(t, None, False, requires_grad and self.want_grad)
        if act.needs_grad:
            gx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=self.device)
            inref.grad = gx

            def emit_bwd():
                if not self.grad_ready(t):
                    inref.grad = None
                    return
                g = self.grad(t)
                s = L.Src(); s.x, s.ld, s.C = g.ptr, g.ld, g.C
                self.keep.append(s)
                self._add(self.bwd, 'in_grad_to_nchw', lib.addk_nhwc_to_nchw, C.byref(s), N, H * W, gx.data_ptr(), rd=[g], wr=[gx])
            self._bwd_emitters.append(emit_bwd)
        return act, inref

    def output_nchw(self, act):
        """Materialise `act` (incl. its lazy BN/ReLU) as a contiguous NCHW tensor."""
        lib = self.lib
        N, H, W, Cc = act.N, act.H, act.W, act.C
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=self.device)
        s = self.src(act)
        self.keep.append(s)
        self._add(self.fwd, 'nhwc_to_nchw', lib.addk_nhwc_to_nchw, C.byref(s), N, H * W, y.data_ptr(), rd=self.lz(act), wr=[y])
        out = OutRef(y)
        if self.want_grad and act.needs_grad:
            def emit_bwd():
                gs = self.grad(act.raw)
                acc = self.acc_flag(gs)
                rows = lib.addk_ew_rows(N * H * W, Cc)
                slab = self._dab(act, rows)
                if slab is not None:
                    act.bn.slabs.append((slab, rows))
                out.bwd_cmd = self._add(self.bwd, 'out_grad', lib.addk_nchw_grad_to_nhwc, None, C.byref(s), N, H * W, gs.ptr, gs.ld,
                                        acc, slab.ptr if slab is not None else None, rd=self.lz(act), wr=[gs, slab])
            self._bwd_emitters.append(emit_bwd)
        return out


def require_device(x):
    """The product path is HIP-only: no CPU fallback exists (tests use oracle/ as the CPU checker)."""
    if not x.is_cuda:
        raise L.AddkError('addk runs on the MI355X HIP path only: got a tensor on %s.  There is no CPU fallback.' % x.device)


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def _in_stage(lib, inref, N, Cc, HW, ptr, ld, stream):
    return lib.addk_nchw_to_nhwc(inref.ptr, N, Cc, HW, ptr, ld, stream)


class InRef:
    """Per-call input binding (pointer patched before each replay)."""

    def __init__(self, shape):
        self.shape, self.ptr, self.grad, self.keep = shape, None, None, None

    def bind(self, x):
        assert tuple(x.shape) == self.shape and x.dtype == torch.float32, 'input changed shape/dtype'
        require_device(x)
        x = x.contiguous()
        self.keep, self.ptr = x, x.data_ptr()


class OutRef:
    """A plan output: tensor + how its incoming gradient pointer is patched per backward call."""

    def __init__(self, y):
        self.y = y
        self.dy_ptr, self.dy_scale, self.dynamic = None, None, True
        self.bwd_args = None     # ResizeBwdArgs (logits path)
        self.fused_ce = False    # logits consumed by the fused up-sampling + cross-entropy launch (train.TrainStep): y is None
        self.bwd_cmd = None      # generic nhwc path

    def set_grad(self, gy):
        gy = gy.contiguous()
        if self.bwd_args is not None:
            self.bwd_args.dy = gy.data_ptr()
        elif self.bwd_cmd is not None:
            self.bwd_cmd.args[0] = gy.data_ptr()
        return gy


class NbtCounter:
    """All `num_batches_tracked` counters of a plan live in ONE int64 tensor (each module's buffer is a 0-dim view
    of it), so a forward bumps them with a single add instead of one tiny launch per BatchNorm."""

    def __init__(self, nbt):
        self.mods = list(nbt)
        self.counts = [int(nbt[m]) for m in self.mods]
        self.flat = self.inc = None

    def _flatten(self):
        dev = self.mods[0].num_batches_tracked.device
        self.flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in self.mods]).to(torch.int64)
        for i, m in enumerate(self.mods):
            m._buffers['num_batches_tracked'] = self.flat[i]
        self.inc = torch.tensor(self.counts, dtype=torch.int64, device=dev)

    def bump(self):
        if not self.mods:
            return
        m0 = self.mods[0].num_batches_tracked
        if self.flat is None or m0.data_ptr() != self.flat.data_ptr() or self.mods[-1].num_batches_tracked.data_ptr() != self.flat[-1].data_ptr():
            self._flatten()          # first use, or another plan / .to() re-pointed the buffers
        self.flat.add_(self.inc)
