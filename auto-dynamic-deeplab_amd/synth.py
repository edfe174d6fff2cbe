"""Synthetic, reproducible parameters and tensors (there is no network for checkpoints or datasets): every value is drawn
from a generator keyed on (seed, NAME), never on RNG call order, so the same state_dict key gets the same numbers in any
module tree that has it — the CPU oracle and the HIP model are filled identically and `bench.py` can assert its loss against
the oracle's.  Used by bench.py, __graft_entry__.smoke() and the tests."""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def rand_tensor(seed, name, shape, scale=1.0):
    return torch.from_numpy((_rng(seed, name).standard_normal(shape) * scale).astype(np.float32))


@torch.no_grad()
def fill_params(module, seed):
    """Deterministic, RNG-order-independent parameter fill keyed on state_dict names.
    Conv weights ~ N(0, 2/fan_in) (kaiming scale), BN gamma ~ 1+0.2N, beta ~ 0.2N,
    running_mean ~ 0.3N, running_var ~ U(0.5,1.5).  Returns a float64 checksum."""
    chk = 0.0
    for name, t in module.state_dict().items():
        if name.endswith('num_batches_tracked'):
            t.zero_()
            continue
        r = _rng(seed, name)
        shp = tuple(t.shape)
        if name.endswith('running_var'):
            v = 0.5 + r.random(shp)
        elif name.endswith('running_mean'):
            v = 0.3 * r.standard_normal(shp)
        elif t.dim() == 1 and name.endswith('weight'):
            v = 1.0 + 0.2 * r.standard_normal(shp)
        elif t.dim() == 1:
            v = 0.2 * r.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = r.standard_normal(shp) * np.sqrt(2.0 / fan_in)
        v = v.astype(np.float32)
        t.copy_(torch.from_numpy(v).reshape(shp))
        chk += float(np.abs(v.astype(np.float64)).sum())
    return chk
