"""Probe: which cross-stream wait patterns survive hipGraph stream capture on this ROCm (fork/join over k streams)."""
import sys, torch
k = int(sys.argv[1]); nodes = int(sys.argv[2]); mode = sys.argv[3]
dev = torch.device('cuda:0')
xs = [torch.zeros(1 << 16, device=dev) for _ in range(k)]
side = [torch.cuda.Stream(device=dev) for _ in range(k - 1)]
def body(main):
    streams = [main] + side
    fork = torch.cuda.Event(); fork.record(main)
    for s in side: s.wait_event(fork)
    last = [None] * k
    for i in range(nodes):
        j = i % k
        with torch.cuda.stream(streams[j]):
            a, b = last[(j + 1) % k], last[(j + 2) % k]
            if mode == 'next' and i % 7 == 3 and a is not None: streams[j].wait_event(a)
            if mode == 'prev' and i % 7 == 3 and b is not None: streams[j].wait_event(b)
            if mode == 'alt' and i % 7 == 3 and a is not None: streams[j].wait_event(a)
            if mode == 'alt' and i % 7 == 5 and b is not None: streams[j].wait_event(b)
            if mode == 'both' and i % 7 == 3 and a is not None and b is not None:
                streams[j].wait_event(a); streams[j].wait_event(b)
            if mode == 'star':
                if j == 0:
                    if i % 7 == 3 and last[1] is not None: streams[0].wait_event(last[1])
                    if i % 7 == 6 and last[2] is not None: streams[0].wait_event(last[2])
                elif i % 5 == 2 and last[0] is not None: streams[j].wait_event(last[0])
            if mode == 'sidemulti' and j == 1:
                if i % 7 == 3 and last[0] is not None: streams[1].wait_event(last[0])
                if i % 7 == 6 and last[2] is not None: streams[1].wait_event(last[2])
            if mode == 'mutual12':
                if j == 1 and i % 7 == 3 and last[2] is not None: streams[1].wait_event(last[2])
                if j == 2 and i % 7 == 5 and last[1] is not None: streams[2].wait_event(last[1])
            if mode == 'ordered':       # side s waits only on sides t > s; everybody <-> main
                if j == 0 and i % 5 == 1:
                    for t in range(1, k):
                        if last[t] is not None: streams[0].wait_event(last[t])
                if j > 0 and i % 7 == 3 and last[0] is not None: streams[j].wait_event(last[0])
                if j > 0 and j + 1 < k and i % 7 == 5 and last[j + 1] is not None: streams[j].wait_event(last[j + 1])
            xs[j].add_(1.0)
            e = torch.cuda.Event(); e.record(streams[j]); last[j] = e
    for s in side:
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
s0 = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s0):
    body(s0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(torch.cuda.current_stream())
g.replay(); torch.cuda.synchronize()
print('ok', k, nodes, mode)
