"""A/B of the drop-in call pattern (model(x); loss.backward(); torch.optim.SGD.step()) under environment switches:
    TAG=default python scripts/dropin_ab.py ; TAG=nobatch ADDK_BATCH_RESIZE_BWD=0 python scripts/dropin_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from addk.modeling.ADD import ADD
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
m = ADD(bench.NETWORK_ARCH, bench.C_INDEX, g, 19, bench.make_args(20), 0)
bench.init_weights(m); m.to(dev)
x, t = bench.synthetic_batch(2, 1024, 2048, 1, dev)
r = bench.drop_in_step(m, x, t, steps=8)
print(os.environ.get('TAG'), round(r['ms_per_step'], 3), flush=True)
