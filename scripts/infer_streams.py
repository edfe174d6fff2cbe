import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import addk
from addk.modeling.ADD import ADD
from addk.synth import fill_params
from bench import NETWORK_ARCH, C_INDEX, make_args, per_exit_latency, segment_roofline, synthetic_batch
dev = torch.device('cuda:0')
g0 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'searched_arch', 'autodeeplab', 'genotype.npy'))
m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0); fill_params(m, 1001); m.to(dev)
x, t = synthetic_batch(2, 1024, 2048, 1, dev)
with torch.no_grad():
    s = segment_roofline(m, x, 'eval')
    print('streams', os.environ.get('ADDK_STREAMS', '2'), 'eval fwd two-stream batched ms', round(s['forward_ms_two_streams_batched'], 3), 'single', round(s['forward_ms_single_stream'], 3))
    s = segment_roofline(m, x, 'train')
    print('   train fwd batched ms', round(s['forward_ms_two_streams_batched'], 3))
    p = per_exit_latency(m, dev)
    print('   per-exit 1024x2048', round(p['1024x2048']['early_exit_ms'], 3), round(p['1024x2048']['final_exit_ms'], 3))
