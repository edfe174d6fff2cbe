import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import addk
from addk import parallel
from addk.modeling.ADD import ADD
from addk.train import TrainStep
from bench import NETWORK_ARCH, C_INDEX, make_args, synthetic_batch
sync = len(sys.argv) > 1 and sys.argv[1] == 'sync'
dev = torch.device('cuda:0')
comm = None
if sync:
    dist.init_process_group('nccl', rank=0, world_size=1)
    comm = parallel.init_sync_bn(force=True)
g0 = np.load('/root/repo/searched_arch/autodeeplab/genotype.npy')
torch.manual_seed(1)
m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20, sync_bn=sync), 0).to(dev).train()
ts = TrainStep(m, (2, 3, 1024, 2048), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, sync_comm=comm, use_graph=False)
x, t = synthetic_batch(2, 1024, 2048, 1, dev)
ts.load_batch(x, t)
for _ in range(3): ts.step()
torch.cuda.synchronize()
host, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); ts.step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); tot.append(t2 - t0)
print('sync' if sync else 'local', 'host enqueue ms', 1e3 * np.median(host), 'total ms', 1e3 * np.median(tot), 'cmds', len(ts.g.fwd) + len(ts.g.bwd))
