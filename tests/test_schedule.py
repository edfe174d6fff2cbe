"""Host logic of the launch-list scheduler (plan.schedule) and of the level re-ordering pass (Graph._level_batch) on
synthetic command DAGs: dependencies survive, the stream assignment respects the constraint that makes hipGraph capture
safe on ROCm 7.2 (two side streams never wait on each other in both directions), implied waits are pruned."""
import random

import pytest
import torch

import addk.plan as P


def _cmds(n, nbuf, seed, pinned_every=0):
    """Random commands over `nbuf` buffers of 64 'channels': each reads 0-2 regions and writes one."""
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        def reg():
            b = rnd.randrange(nbuf); lo = rnd.choice([0, 16, 32]); return (1000 + b, lo, lo + rnd.choice([16, 32]))
        c = P.Cmd('k%d' % i, None, (), rd=[reg() for _ in range(rnd.randrange(3))], wr=[reg()],
                  pin=bool(pinned_every and i % pinned_every == 0))
        out.append(c)
    return out


def _deps(cmds):
    """Reference dependency relation of the sequential list: RAW, WAW, WAR on overlapping regions."""
    deps = [set() for _ in cmds]
    for i, c in enumerate(cmds):
        for j in range(i):
            d = cmds[j]
            if any(P._overlap(a, b) for a in c.rd for b in d.wr) or any(P._overlap(a, b) for a in c.wr for b in d.wr) or \
               any(P._overlap(a, b) for a in c.wr for b in d.rd):
                deps[i].add(j)
    return deps


def _happens_before(cmds):
    """Transitive closure of (stream program order + recorded waits): hb[i] = set of commands surely finished before i."""
    hb = [set() for _ in cmds]
    last = {}
    for i, c in enumerate(cmds):
        if c.stream in last:
            hb[i] |= hb[last[c.stream]] | {last[c.stream]}
        for j in c.waits:
            hb[i] |= hb[j] | {j}
        last[c.stream] = i
    return hb


@pytest.mark.parametrize('nstreams', [1, 2, 3, 4, 6])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_schedule_preserves_every_dependency(nstreams, seed):
    cmds = _cmds(300, 12, seed, pinned_every=17)
    P.schedule(cmds, nstreams)
    deps, hb = _deps(cmds), _happens_before(cmds)
    for i, c in enumerate(cmds):
        assert 0 <= c.stream < nstreams and (not c.pin or c.stream == 0)
        assert deps[i] <= hb[i], 'command %d may start before %s' % (i, sorted(deps[i] - hb[i])[:3])
        for j in c.waits:
            assert cmds[j].event, 'wait on a command that records no event'
            assert cmds[j].stream != c.stream


@pytest.mark.parametrize('nstreams', [3, 4, 6])
def test_side_streams_never_wait_on_each_other_both_ways(nstreams):
    """hipStreamEndCapture on ROCm 7.2 dumps core when two captured side streams wait on each other (scripts/capture_probe.py):
    a side stream s may only wait on side streams t > s; the origin stream 0 may wait on / be waited on by anybody."""
    for seed in range(4):
        cmds = _cmds(400, 10, 100 + seed)
        P.schedule(cmds, nstreams)
        for c in cmds:
            for j in c.waits:
                t = cmds[j].stream
                assert c.stream == 0 or t == 0 or t > c.stream, (c.stream, t)


def test_implied_waits_are_pruned():
    # A(s0) -> B(s1) -> C(s?) with C also reading A's output: the wait on A is implied by the wait on B
    a = P.Cmd('a', None, (), rd=[], wr=[(1, 0, 16)])
    b = P.Cmd('b', None, (), rd=[(1, 0, 16)], wr=[(2, 0, 16)])
    filler = [P.Cmd('f%d' % i, None, (), rd=[], wr=[(10 + i, 0, 16)]) for i in range(3)]
    c = P.Cmd('c', None, (), rd=[(1, 0, 16), (2, 0, 16)], wr=[(3, 0, 16)])
    cmds = [a] + filler + [b, c]
    P.schedule(cmds, 3)
    hb = _happens_before(cmds)
    assert {0, 4} <= hb[5]
    assert len(c.waits) <= 1


class _FakeGraph:
    """Just enough of Graph for _level_batch (no batchable payloads: pure re-ordering)."""
    _BATCHED = {}
    _level_batch = P.Graph._level_batch


@pytest.mark.parametrize('seed', [3, 4, 5, 6])
def test_level_order_is_a_valid_topological_order(seed):
    cmds = _cmds(250, 14, seed)
    deps = _deps(cmds)
    ident = {id(c): i for i, c in enumerate(cmds)}
    lst = list(cmds)
    _FakeGraph()._level_batch(lst)
    assert sorted(ident[id(c)] for c in lst) == list(range(len(cmds)))          # a permutation
    pos = {ident[id(c)]: k for k, c in enumerate(lst)}
    for i in range(len(cmds)):
        for j in deps[i]:
            assert pos[j] < pos[i], 'dependency %d -> %d inverted by the level order' % (j, i)


def test_tensor_regions_are_byte_intervals():
    flat = torch.zeros(100)
    a, b = flat[:40], flat[40:]
    ra, rb = P._region(a), P._region(b)
    assert ra[0] == rb[0] and not P._overlap(ra, rb)            # two views of one flat buffer do not alias
    assert P._overlap(P._region(flat), ra) and P._overlap(P._region(flat[30:50]), rb)
