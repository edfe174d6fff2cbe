"""Multi-process host logic on CPU (gloo, world_size 2): the SyncBN statistics exchange, flat-gradient averaging,
parameter broadcast and sharding.  Kernel launches are stubbed (no GPU here); the arithmetic of the exchange is
checked against the oracle's global-batch BatchNorm definition."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _syncbn_math(rank, world):
    """Each rank holds a shard; (sum, sumsq) in fp64 -> all-reduce through SyncBNComm -> the finalize formulas of
    csrc/bn.hip restated in torch -> must equal F.batch_norm over the concatenated batch; same for backward."""
    import addk  # noqa: F401
    import oracle
    from addk.parallel import SyncBNComm
    from addk.plan import LateVec
    from _util import rand_tensor
    comm = SyncBNComm()
    C = 12
    shards = [rand_tensor(91, 'sbn_%d' % r, (2, C, 5, 7)) * (1 + 0.3 * r) + 0.2 * r for r in range(world)]
    x = shards[rank].double()
    w, b = rand_tensor(91, 'sbn_w', (C,)).double() * 0.2 + 1, rand_tensor(91, 'sbn_b', (C,)).double() * 0.2

    class FakeBuf:            # LateVec.view() only needs .t (and .ptr for the binders)
        def __init__(self, t):
            self.t, self.ptr = t, 0
    stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))], dim=1).reshape(-1).contiguous()   # [C][2] fp64
    # the plan stores fp64 pairs in fp32 slots; a second vector shares the arena, as the BatchNorms of one level do
    arena = FakeBuf(torch.cat([stats.view(torch.float32), torch.full((8,), float(rank + 1)).double().view(torch.float32)]))
    cmds = []

    class G:
        fwd = cmds

        def _add(self, lst, name, fn, *args, **kw):
            import types
            c = types.SimpleNamespace(name=name, fn=fn, args=list(args), payload=None)
            lst.append(c)
            return c
    g = G()
    v1, v2 = LateVec(4 * C, f64=True), LateVec(16, f64=True)
    comm.emit_allreduce(g, g.fwd, v1); comm.emit_allreduce(g, g.fwd, v2)
    v1.bind(arena, 0); v2.bind(arena, 4 * C)
    # the merged form plan.Graph._level_batch emits for one level: ONE plain all-reduce of the arena
    assert comm._allreduce(arena.t.view(torch.float64), 0) == 0
    tri = world * (world + 1) / 2
    assert torch.equal(v2.view(), torch.full((8,), tri, dtype=torch.float64))
    buf = FakeBuf(v1.view().clone().view(torch.float32))
    tot = buf.t.view(torch.float64).reshape(C, 2)
    count = float(world * 2 * 5 * 7)
    mean = tot[:, 0] / count
    var = tot[:, 1] / count - mean * mean
    invstd = (var + 1e-5).rsqrt()
    a = w * invstd
    bb = b - mean * a
    y = x * a.view(1, -1, 1, 1) + bb.view(1, -1, 1, 1)
    ref = oracle.global_batch_norm([s.double() for s in shards], torch.zeros(C).double(), torch.ones(C).double(), w, b)[rank]
    err_fwd = float((y - ref).abs().max())
    # backward: local (dmean_tot, dvar) all-reduced, then c1/c2 with the GLOBAL count
    full = torch.cat([s.double() for s in shards]).requires_grad_(True)
    probe = torch.cat([rand_tensor(92, 'sbn_p%d' % r, (2, C, 5, 7)).double() for r in range(world)])
    yy = torch.nn.functional.batch_norm(full, None, None, w, b, True, 0.1, 1e-5)
    (yy * probe).sum().backward()
    gref = full.grad[rank * 2:(rank + 1) * 2]
    dz = probe[rank * 2:(rank + 1) * 2]
    dA, dB = (dz * x).sum((0, 2, 3)), dz.sum((0, 2, 3))
    t = dA - mean * dB
    dvar = -0.5 * w * t * invstd ** 3
    dmean_tot = -a * dB - 2 * mean * dvar
    dmv = torch.stack([dmean_tot, dvar], dim=1).reshape(-1).float().contiguous()
    cmds2 = []
    v3 = LateVec(2 * C)
    comm.emit_allreduce(g, cmds2, v3)
    v3.bind(FakeBuf(dmv), 0)
    for c in cmds2:                      # the un-merged form (a level with level batching disabled)
        assert c.fn(*c.args, 0) == 0
    dmv = dmv.double().reshape(C, 2)
    c1, c2 = dmv[:, 0] / count, 2 * dmv[:, 1] / count
    gx = dz * a.view(1, -1, 1, 1) + c1.view(1, -1, 1, 1) + c2.view(1, -1, 1, 1) * x
    err_bwd = float((gx - gref).abs().max() / gref.abs().max())
    return err_fwd, err_bwd, comm.calls


def test_syncbn_exchange_matches_global_batch_norm():
    for err_fwd, err_bwd, calls in _spawn(_syncbn_math):
        assert err_fwd < 1e-9 and err_bwd < 1e-5 and calls == 2        # packed forward arena, backward vector


def _grads_and_broadcast(rank, world):
    from addk import parallel
    torch.manual_seed(rank)
    m = torch.nn.Linear(5, 3)
    parallel.broadcast_params(m)
    w0 = m.weight.detach().clone()
    flat = torch.full((7,), float(rank + 1))
    parallel.allreduce_grads(flat)
    return w0.numpy().tolist(), flat.tolist(), parallel.shard_indices(10, rank, world)


def test_grad_allreduce_broadcast_and_sharding():
    r = _spawn(_grads_and_broadcast)
    assert r[0][0] == r[1][0]                            # identical parameters after the rank-0 broadcast
    assert r[0][1] == r[1][1] == [3.0] * 7               # sum over ranks; TrainStep scales by 1/world in the SGD kernel
    assert sorted(r[0][2] + r[1][2]) == list(range(10))  # disjoint shards covering the data set


def _plan_with_syncbn(rank, world):
    """Build the ADD training plan with cross-rank BatchNorm enabled (launches stubbed): one all-reduce per BatchNorm
    application in forward and one in backward."""
    import collections
    import addk.plan as P
    from addk import parallel
    from addk.modeling.ADD import ADD
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, make_args
    P.Graph.run = lambda self, cmds, stream: None
    P.require_device = lambda x: None
    P.current_stream = lambda: 0
    parallel.init_sync_bn()
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=True), 0).train()
    outs = m(torch.randn(2, 3, 65, 129))
    plan = next(iter(m._plans().values()))
    f = collections.Counter(n for n, _, _ in plan.g.fwd)
    b = collections.Counter(n for n, _, _ in plan.g.bwd)

    def exchanged(lst):      # vectors exchanged: single all-reduces + members of the packed (one-level) ones
        return sum(1 for c in lst if c.name == 'allreduce') + sum(c.members for c in lst if c.name == 'allreduce_packed')
    calls = f['allreduce'] + f['allreduce_packed'] + b['allreduce'] + b['allreduce_packed']
    def total(lst, name):
        return sum(1 for c in lst if c.name == name) + sum(c.args[1] for c in lst if c.name == name + '_batch')
    return exchanged(plan.g.fwd), exchanged(plan.g.bwd), total(plan.g.fwd, 'slab_reduce'), total(plan.g.bwd, 'bn_bwd_coeffs'), len(outs), calls


def test_add_plan_emits_one_allreduce_per_batchnorm():
    for fa, ba, sr, co, n, calls in _spawn(_plan_with_syncbn):
        assert (fa, ba, sr, co, n) == (312, 312, 312, 312, 2)
        # the exchanges of one dependency level are ONE all-reduce of a shared arena; the count cannot fall below the depth of
        # the network in BatchNorms (~150 per direction: every exchange on the critical path waits for the one before it)
        assert calls < 400


def _train_step_collectives(rank, world):
    """The FULL fused train step (forward, CE, backward, bucketed gradient all-reduce, SGD) replayed with stubbed kernels:
    every rank must issue the identical collective sequence (deadlock safety), the SyncBN exchanges must be one per
    dependency level, and the bucketed gradient all-reduce must equal a single all-reduce of the flat buffer."""
    import ctypes
    import addk.plan as P
    import addk.train as T
    from addk import parallel
    from addk.modeling.ADD import ADD
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, make_args

    def dry_run(self, cmds, stream):             # collectives run for real (gloo); kernel launches are skipped
        for c in cmds:
            if c.name in ('allreduce', 'allreduce_packed', 'grad_allreduce'):
                assert c.fn(*c.args, stream) == 0
    P.Graph.run = dry_run
    P.require_device = lambda x: None
    T._plan.require_device = lambda x: None
    P.current_stream = lambda: 0
    torch.cuda.current_stream = lambda *a, **k: type('S', (), {'cuda_stream': 0})()
    comm = parallel.init_sync_bn()
    comm.log = []
    torch.manual_seed(0)
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=True), 0).train()
    ts = T.TrainStep(m, (2, 3, 65, 129), sync_comm=comm, use_graph=False, nstreams=2)
    ts._sgd = lambda st: None
    assert ts.world == world and ts.gsync is not None
    # buckets: contiguous, disjoint, covering the flat gradient buffer
    bk = sorted(ts.gsync.buckets)
    assert bk[0][0] == 0 and bk[-1][1] == ts.flat_g.numel() and all(a[1] == b[0] for a, b in zip(bk, bk[1:])) and len(bk) >= 2
    # every bucket's all-reduce sits behind the last launch that writes into it
    names = [c.name for c in ts.g.bwd]
    pos = [i for i, n in enumerate(names) if n == 'grad_allreduce']
    assert len(pos) == len(bk) and pos[0] < len(names) - 1, 'first bucket must be issued before the end of the backward list'
    base = ts.flat_g.untyped_storage().data_ptr()
    for i in pos:
        lo, hi = ts.g.bwd[i].wr[0][1], ts.g.bwd[i].wr[0][2]
        later = [c.name for c in ts.g.bwd[i + 1:] if c.name != 'grad_allreduce' and any(r[0] == base and r[1] < hi and lo < r[2] for r in c.wr)]
        assert not later, later
    ts.flat_g.fill_(float(rank + 1))
    ts.step()
    tri = world * (world + 1) / 2
    assert torch.equal(ts.flat_g, torch.full_like(ts.flat_g, tri))       # bucketed == one all-reduce of the whole buffer
    nstats = sum(1 for k in comm.log if k[0] == 'stats')
    return comm.log, nstats, len(bk)


def test_train_step_collective_sequence_is_identical_on_all_ranks():
    r = _spawn(_train_step_collectives)
    assert r[0][0] == r[1][0] and len(r[0][0]) > 0        # same kinds, sizes and dtypes in the same order on both ranks
    assert r[0][1] < 400                                   # SyncBN: one exchange per dependency level (312 + 312 BatchNorm calls)
    kinds = [k[0] for k in r[0][0]]
    assert 'grad' in kinds and kinds.index('grad') < len(kinds) - 1 - kinds[::-1].index('stats'), 'a gradient bucket overlaps the backward pass'


def _ddp_without_syncbn(rank, world):
    """The reference's DDP without --sync-bn: gradients are still averaged (world came from the SyncBN comm)."""
    import addk.plan as P
    import addk.train as T
    from addk.modeling.ADD import ADD
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, make_args
    P.Graph.run = lambda self, cmds, stream: [c.fn(*c.args, stream) for c in cmds if c.name == 'grad_allreduce'] and None
    P.require_device = lambda x: None
    T._plan.require_device = lambda x: None
    P.current_stream = lambda: 0
    torch.cuda.current_stream = lambda *a, **k: type('S', (), {'cuda_stream': 0})()
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=False), 0).train()
    ts = T.TrainStep(m, (2, 3, 65, 129), sync_comm=None, use_graph=False)
    seen = {}
    ts._sgd = lambda st: seen.setdefault('scale', 1.0 / ts.world)
    ts.flat_g.fill_(float(rank + 1))
    ts.step()
    return ts.world, float(ts.flat_g.min()), float(ts.flat_g.max()), seen['scale'], sum(1 for c in ts.g.fwd if 'allreduce' in c.name)


def test_gradients_are_averaged_without_syncbn():
    for world, lo, hi, scale, nstat in _spawn(_ddp_without_syncbn):
        assert world == 2 and lo == hi == 3.0 and scale == 0.5 and nstat == 0


def test_shard_indices_equal_torch_distributed_sampler():
    """parallel.shard_indices reproduces torch.utils.data.DistributedSampler (dataloaders/__init__.py:33: default shuffle=True,
    seed 0, never set_epoch) for every rank, incl. the wrap-around padding, other epochs/seeds and shuffle=False."""
    from torch.utils.data import DistributedSampler
    from addk import parallel
    for n, world in ((10, 4), (2975, 8), (7, 3), (16, 16), (5, 8)):
        ds = list(range(n))
        for rank in range(world):
            for kw in (dict(), dict(shuffle=False), dict(seed=3)):
                s = DistributedSampler(ds, num_replicas=world, rank=rank, **kw)
                assert list(s) == parallel.shard_indices(n, rank, world, **kw), (n, world, rank, kw)
            s = DistributedSampler(ds, num_replicas=world, rank=rank)
            s.set_epoch(5)
            assert list(s) == parallel.shard_indices(n, rank, world, epoch=5)


@pytest.mark.parametrize('mode', ['capture_ok', 'capture_refused', 'capture_hang'])
def test_bench_multi_rank_control_flow_world2(mode):
    """bench.py's N > 1 control flow (bench.timed_region: eager timing, capture trial under the watchdog) on two gloo ranks with a stub
    step (tests/tools/bench_flow_driver.py) — the rehearsal of what the driver's first real multi-GPU run will execute:
      capture_ok       rank 0 prints exactly ONE JSON line with n_gpus 2, the faster (captured) mode wins, both ranks exit 0;
      capture_refused  one line, eager numbers, hip_graph false, both ranks exit 0;
      capture_hang     the watchdog fires: rank 0 still prints exactly one line — the eager measurement with capture_hang true and the
                       phase it was stuck in — and BOTH ranks exit with code 3 (a hang must not look like a success)."""
    import json
    import subprocess
    port = 29500 + ((os.getpid() + hash(mode)) % 2000)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', ADDK_BENCH_CAPTURE_LIMIT='1.0')
    env.pop('ADDK_BENCH_TRY_CAPTURE', None)
    drv = os.path.join(ROOT, 'tests', 'tools', 'bench_flow_driver.py')
    procs = [subprocess.Popen([sys.executable, drv, mode], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    rcs = [p.returncode for p in procs]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith('{')]
    assert len(lines0) == 1 and not lines1, (outs, rcs)
    d = json.loads(lines0[0])
    assert d['n_gpus'] == 2 and d['value'] > 0
    if mode == 'capture_hang':
        assert rcs == [3, 3], (rcs, outs[0][1][-500:], outs[1][1][-500:])
        assert d.get('capture_hang') is True and d['capture_phase'] == 'capture' and d['config']['hip_graph'] is False
        assert 'watchdog fired' in outs[1][1]
    else:
        assert rcs == [0, 0], (rcs, outs[0][1][-800:], outs[1][1][-800:])
        assert 'capture_hang' not in d
        if mode == 'capture_ok':
            assert d['config']['hip_graph'] is True and set(d['ms_per_step_by_mode']) == {'eager_list', 'hip_graph'}
            assert d['ms_per_step_by_mode']['hip_graph'] < d['ms_per_step_by_mode']['eager_list']
        else:
            assert d['config']['hip_graph'] is False and isinstance(d['ms_per_step_by_mode']['hip_graph'], str)


# ---- the small-message exchange of the SyncBN statistics (addk.parallel.SmallComm, csrc/comm.hip) on host memory ---------------------------
def _small_comm_protocol(rank, world):
    """Handle exchange through the process group, collective agreement, the start-up self-test against dist.all_reduce, then 9 exchanges of
    both dtypes (odd count: both flag / data sets are reused several times): every result equals the rank-ordered sum bit for bit and is
    identical on both ranks; the SyncBNComm wrapper routes fitting vectors through the mailboxes and oversized ones through the stock call."""
    from addk.parallel import SmallComm, SyncBNComm
    from _host_mailbox import HostMailbox
    cpu = torch.device('cpu')
    sc = SmallComm.create(max_bytes=4096, device=cpu, transport=HostMailbox(), ctl_device=cpu)
    assert sc is not None and sc.t.status() == (4, 0)
    out = []
    for i in range(9):
        dt = torch.float64 if i % 2 == 0 else torch.float32
        n = 16 + 8 * i
        mine = [torch.from_numpy(np.random.default_rng(1000 * r + i).standard_normal(n)).to(dt) for r in range(world)]
        want = mine[0].clone()
        for r in range(1, world):
            want = want + mine[r]
        got = mine[rank].clone()
        assert sc.fits(got) and sc.allreduce(got, 0) == 0
        assert torch.equal(got, want), (i, float((got - want).abs().max()))
        out.append(got.double().sum().item())
    assert sc.check() == 13
    comm = SyncBNComm(small=sc)
    comm.log = []
    small = torch.full((64,), float(rank + 1), dtype=torch.float64)
    big = torch.full((4096,), float(rank + 1), dtype=torch.float64)          # 32 KB > max_bytes: stock all_reduce
    comm._allreduce(small, 0); comm._allreduce(big, 0)
    tot = float(sum(range(1, world + 1)))
    assert torch.all(small == tot) and torch.all(big == tot) and sc.calls == 10 and comm.calls == 2
    comm.close()
    return out


def _small_comm_timeout(rank, world):
    """Rank 1 never publishes exchange 6: rank 0's poll is bounded — it returns, the error word names the exchange and the missing rank, check()
    raises; rank 1 (which received rank 0's flag) finishes clean.  Nobody hangs."""
    from addk.parallel import SmallComm
    from addk._lib import AddkError
    from _host_mailbox import HostMailbox
    cpu = torch.device('cpu')
    sc = SmallComm.create(max_bytes=1024, device=cpu, transport=HostMailbox(timeout_s=0.5, skip=(6,) if rank == 1 else ()), ctl_device=cpu)
    assert sc is not None
    v = torch.ones(8, dtype=torch.float64)
    sc.allreduce(v, 0)                      # exchange 5: fine
    assert torch.all(v == world)
    sc.allreduce(torch.ones(8, dtype=torch.float64), 0)      # exchange 6
    try:
        sc.check()
        res = 'clean'
    except AddkError as e:
        res = str(e)
    dist.barrier()
    sc.close()
    return res


def _small_comm_refused(rank, world):
    """One rank cannot allocate its mailbox: EVERY rank gets None (the decision is collective) and the stock collective stays in use."""
    from addk.parallel import SmallComm, SyncBNComm
    from _host_mailbox import HostMailbox

    class Broken(HostMailbox):
        def alloc(self, world, max_bytes):
            raise RuntimeError('no hipIpc here')
    cpu = torch.device('cpu')
    sc = SmallComm.create(max_bytes=1024, device=cpu, transport=Broken() if rank == 1 else HostMailbox(), ctl_device=cpu)
    comm = SyncBNComm(small=None)
    comm._small_tried = True
    v = torch.full((8,), float(rank + 1), dtype=torch.float64)
    comm._allreduce(v, 0)
    return sc is None and bool(torch.all(v == 3.0))


def test_small_comm_protocol_on_two_ranks():
    a, b = _spawn(_small_comm_protocol)
    assert a == b


def test_small_comm_timeout_is_bounded_and_reported():
    a, b = _spawn(_small_comm_timeout)
    assert 'exchange 6 timed out waiting for rank 1' in a, a
    assert b == 'clean', b


def test_small_comm_is_all_or_nothing():
    assert _spawn(_small_comm_refused) == [True, True]
