"""How long does the host spend in one hipGraph replay of the training step, and does a replay wait for the previous one?
Prints the host time of back-to-back replays (no sync in between) and the wall time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import addk
from addk.modeling.ADD import ADD
from addk.train import TrainStep
from bench import NETWORK_ARCH, C_INDEX, make_args, synthetic_batch, init_weights
dev = torch.device('cuda:0')
g0 = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0)
init_weights(m); m.to(dev)
ts = TrainStep(m, (2, 3, 1024, 2048), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)
x, t = synthetic_batch(2, 1024, 2048, 1, dev)
ts.load_batch(x, t)
for _ in range(3): ts.step()
torch.cuda.synchronize()
host = []
t00 = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter(); ts.step(); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t00) / 10
print('host ms per replay:', ' '.join('%.2f' % (1e3 * h) for h in host), '| wall ms per step %.2f' % (1e3 * wall))
one = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts.step(); t1 = time.perf_counter(); torch.cuda.synchronize(); one.append((t1 - t0, time.perf_counter() - t0))
print('isolated replay: host %.2f ms, wall %.2f ms' % (1e3 * np.median([a for a, _ in one]), 1e3 * np.median([b for _, b in one])))
