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
    from addk.plan import Vec
    from _util import rand_tensor
    comm = SyncBNComm()
    C = 12
    shards = [rand_tensor(91, 'sbn_%d' % r, (2, C, 5, 7)) * (1 + 0.3 * r) + 0.2 * r for r in range(world)]
    x = shards[rank].double()
    w, b = rand_tensor(91, 'sbn_w', (C,)).double() * 0.2 + 1, rand_tensor(91, 'sbn_b', (C,)).double() * 0.2

    class FakeBuf:            # Vec.view() only needs .t
        def __init__(self, t):
            self.t = t
    stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))], dim=1).reshape(-1).contiguous()   # [C][2] fp64
    buf = FakeBuf(stats.view(torch.float32))       # the plan stores fp64 pairs in an fp32 buffer
    cmds = []

    class G:
        fwd = cmds

        def _add(self, lst, name, fn, *args, **kw):
            lst.append((name, fn, args))
            import types
            return types.SimpleNamespace()
    g = G()
    comm.emit_allreduce(g, g.fwd, Vec(buf, 0, 4 * C))
    for name, fn, args in cmds:
        assert fn(*args, 0) == 0
    # the grouped form used for the BatchNorms of one dependency level (falls back to a loop on gloo)
    t1, t2 = torch.full((3,), float(rank + 1)), torch.full((5,), 2.0 * (rank + 1), dtype=torch.float64)
    assert comm._allreduce_multi([t1, t2], 0) == 0
    tri = world * (world + 1) / 2
    assert torch.equal(t1, torch.full((3,), tri)) and torch.equal(t2, torch.full((5,), 2.0 * tri, dtype=torch.float64))
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
    comm.emit_allreduce(g, cmds2, Vec(FakeBuf(dmv), 0, 2 * C))
    for name, fn, args in cmds2:
        fn(*args, 0)
    dmv = dmv.double().reshape(C, 2)
    c1, c2 = dmv[:, 0] / count, 2 * dmv[:, 1] / count
    gx = dz * a.view(1, -1, 1, 1) + c1.view(1, -1, 1, 1) + c2.view(1, -1, 1, 1) * x
    err_bwd = float((gx - gref).abs().max() / gref.abs().max())
    return err_fwd, err_bwd, comm.calls


def test_syncbn_exchange_matches_global_batch_norm():
    for err_fwd, err_bwd, calls in _spawn(_syncbn_math):
        assert err_fwd < 1e-9 and err_bwd < 1e-5 and calls == 3        # forward, grouped probe, backward


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

    def exchanged(lst):      # tensors exchanged: single all-reduces + members of the grouped (one-level) ones
        return sum(1 for c in lst if c.name == 'allreduce') + sum(len(c.args[0]) for c in lst if c.name == 'allreduce_multi')
    calls = f['allreduce'] + f['allreduce_multi'] + b['allreduce'] + b['allreduce_multi']
    def total(lst, name):
        return sum(1 for c in lst if c.name == name) + sum(c.args[1] for c in lst if c.name == name + '_batch')
    return exchanged(plan.g.fwd), exchanged(plan.g.bwd), total(plan.g.fwd, 'slab_reduce'), total(plan.g.bwd, 'bn_bwd_coeffs'), len(outs), calls


def test_add_plan_emits_one_allreduce_per_batchnorm():
    for fa, ba, sr, co, n, calls in _spawn(_plan_with_syncbn):
        assert (fa, ba, sr, co, n) == (312, 312, 312, 312, 2)
        assert calls < 400          # the exchanges of one dependency level share a grouped collective call
