"""The small-message SyncBN exchange (csrc/comm.hip, addk.parallel.SmallComm) on the GPU:

 * world 1, in process: mailbox allocation, hipIpc export, sequence numbers, both dtypes, capture in a hipGraph;
 * world 2 — TWO PROCESSES SHARING THE ONE GPU, their mailboxes mapped into each other through hipIpc: every exchange equal, bit for bit,
   to `dist.all_reduce` (gloo carries the control plane and the checker), eager and replayed from a captured hipGraph.  The rehearsal runs
   ONCE in fresh child processes under a time limit; a failed or timed-out child fails the test, nothing is retried.
Latency over xGMI is NOT what this measures (one device): DESIGN.md §7."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_exchange_is_the_identity_and_counts_its_sequence():
    import addk  # noqa: F401
    from addk.parallel import HipMailbox
    dev = torch.device('cuda:0')
    t = HipMailbox()
    h = t.alloc(1, 4096)
    assert len(h) == 64
    t.open(0, 1, 4096, [h])
    st = torch.cuda.current_stream().cuda_stream
    for i, dt in enumerate((torch.float64, torch.float32, torch.float64)):
        v = torch.randn(128, device=dev, dtype=dt)
        w = v.clone()
        assert t.allreduce(w, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(v, w)
        assert t.status() == (i + 1, 0)
    # capturable: the sequence number lives in device memory
    v = torch.arange(64, device=dev, dtype=torch.float64)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(3):
                assert t.allreduce(v, s.cuda_stream) == 0
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    assert t.status() == (3 + 6, 0) and torch.equal(v, torch.arange(64, device=dev, dtype=torch.float64))
    # argument checks come back as status codes with a message, not as faults
    big = torch.zeros(1024, device=dev, dtype=torch.float64)
    assert t.allreduce(big, st) != 0 and 'exceed' in t.lib.addk_last_error().decode()
    del g
    t.close()


CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
rank, world = int(sys.argv[1]), 2
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE='2')
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
import addk
from addk.parallel import SmallComm
dev, cpu = torch.device('cuda:0'), torch.device('cpu')
sc = SmallComm.create(max_bytes=1 << 15, device=dev, ctl_device=cpu)
assert sc is not None, 'small-message path refused'
st = torch.cuda.current_stream().cuda_stream
for i in range(24):
    dt = torch.float64 if i %% 3 else torch.float32
    n = 8 * (1 + 37 * i %% 500)
    mine = torch.from_numpy(np.random.default_rng(77 * rank + i).standard_normal(n)).to(dt)
    ref = mine.clone(); dist.all_reduce(ref)
    got = mine.to(dev)
    assert sc.allreduce(got, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref), (i, float((got.cpu() - ref).abs().max()))
# captured: 5 exchanges per replay, 3 replays, the operand rewritten between replays
v = torch.zeros(256, device=dev, dtype=torch.float64)
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(5):
            assert sc.allreduce(v, s.cuda_stream) == 0
for rep in range(3):
    v.fill_(float(rank + 1 + rep))
    torch.cuda.synchronize(); dist.barrier()
    g.replay()
    torch.cuda.synchronize()
    want = float(2 * rep + 3) * 2.0 ** 4        # the sum over the two ranks, then doubled by each of the four further exchanges
    assert torch.all(v == want), (rep, float(v[0]), want)
seq = sc.check()
assert seq == 4 + 24 + 15, seq
dist.barrier()
del g
sc.close()
dist.destroy_process_group()
print('rank %%d ok: %%d exchanges' %% (rank, seq))
'''


def test_two_processes_on_one_gpu_exchange_through_hipipc_mailboxes(tmp_path):
    script = tmp_path / 'comm_child.py'
    script.write_text(CHILD % {'root': ROOT})
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail('the two-process rehearsal did not finish in 240 s (not retried)')
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d exited with %d:\n%s\n%s' % (r, p.returncode, so[-2000:], se[-4000:])
        assert 'rank %d ok: 43 exchanges' % r in so
