"""Test infrastructure: the mailbox protocol of csrc/comm.hip (push into every rank's mailbox, flag, bounded poll of the own mailbox,
rank-ordered sum, two flag / data sets) restated over POSIX shared memory, so that the host side of `addk.parallel.SmallComm` — handle
exchange through the process group, collective agreement, the self-test against `dist.all_reduce`, the error path of a timed-out
exchange — runs on two gloo ranks without a GPU.  Same layout as the device code: flags[2][16] on 64-byte lines, data[2][world][slot]."""
import time
from multiprocessing import shared_memory

import numpy as np
import torch

MAXW, LINE = 16, 64


class HostMailbox:
    def __init__(self, timeout_s=2.0, skip=()):
        self.timeout_s = timeout_s
        self.skip = set(skip)           # exchange numbers (1-based) this rank does NOT publish: the fault the bounded poll has to survive
        self.own, self.peers, self.seq, self.err = None, [], 0, 0

    def _bytes(self, world, max_bytes):
        self.slot = (max_bytes + 63) // 64 * 64
        return 2 * MAXW * LINE + 2 * world * self.slot

    def alloc(self, world, max_bytes):
        self.own = shared_memory.SharedMemory(create=True, size=self._bytes(world, max_bytes))
        self.own.buf[:] = bytes(len(self.own.buf))
        name = self.own.name.encode()
        assert len(name) < 64
        return name + b'\0' * (64 - len(name))

    def open(self, rank, world, max_bytes, handles):
        self.rank, self.world = rank, world
        self._bytes(world, max_bytes)
        self.peers = [self.own if r == rank else shared_memory.SharedMemory(name=h.rstrip(b'\0').decode()) for r, h in enumerate(handles)]
        self.box = [np.ndarray((len(p.buf) // 8,), dtype=np.uint64, buffer=p.buf) for p in self.peers]

    def _flag(self, set_, r):
        return (set_ * MAXW + r) * LINE // 8

    def _data(self, set_, r):
        return (2 * MAXW * LINE + (set_ * self.world + r) * self.slot) // 8

    def allreduce(self, t, stream):
        seq = self.seq + 1
        set_ = seq & 1
        raw = t.numpy().view(np.uint64)
        n = raw.size
        peers = [r for r in range(self.world) if r != self.rank]          # the own share never leaves the rank (csrc/comm.hip)
        if seq not in self.skip:
            for r in peers:
                self.box[r][self._data(set_, self.rank):self._data(set_, self.rank) + n] = raw
            for r in peers:
                self.box[r][self._flag(set_, self.rank)] = seq
        t0 = time.monotonic()
        for r in peers:
            while self.box[self.rank][self._flag(set_, r)] < seq:
                if time.monotonic() - t0 > self.timeout_s:
                    self.err = (1 << 63) | (seq << 8) | r
                    break
                time.sleep(0.0005)
        dt = np.float64 if t.dtype == torch.float64 else np.float32
        share = lambda r: raw.copy().view(dt) if r == self.rank else self.box[self.rank][self._data(set_, r):self._data(set_, r) + n].copy().view(dt)
        acc = share(0)
        for r in range(1, self.world):
            acc = acc + share(r)
        t.copy_(torch.from_numpy(acc.copy()))
        self.seq = seq
        return 0

    def status(self):
        return self.seq, self.err

    def close(self):
        self.box = []
        for r, p in enumerate(self.peers):
            p.close()
        if self.own is not None:
            try:
                self.own.unlink()
            except FileNotFoundError:
                pass
        self.own, self.peers = None, []
