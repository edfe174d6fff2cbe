"""Shared helpers for tests and for tests/golden/make_golden.py (no reference code here)."""
import zlib
from types import SimpleNamespace

import numpy as np
import torch

GENOTYPE_AUTODEEPLAB = np.array(  # searched_arch/autodeeplab/genotype.npy (data; unsorted rows 8,9 — SURVEY Q1)
    [[0, 7], [1, 4], [2, 4], [3, 6], [5, 4], [8, 4], [11, 5], [13, 5], [19, 7], [18, 5]], dtype=np.int64)
GENOTYPE_BASELINE_2 = np.array(   # searched_arch/searched_baseline/genotype_2.npy (data; rows 4,5 unsorted)
    [[0, 7], [1, 4], [2, 6], [4, 4], [8, 6], [5, 4], [9, 6], [11, 7], [14, 7], [16, 5]], dtype=np.int64)
GENOTYPE_40_1 = np.array(         # searched_arch/40_5e_38_lr/genotype_1.npy (data; three unsorted pairs)
    [[1, 5], [0, 4], [2, 4], [4, 4], [5, 4], [8, 5], [13, 5], [12, 4], [18, 5], [14, 4]], dtype=np.int64)
NETWORK_PATH_BASELINE = [0, 1, 2, 2, 3, 2, 2, 1, 2, 1, 1, 2]  # searched_baseline/network_path.npy (data)
ARCH_C2 = dict(network_arch=[1, 2, 2, 2, 3, 2, 2, 1, 1, 1, 1, 2], C_index=[5], low_level_layer=0)  # train.py:75-79
ARCH_C3 = dict(network_arch=[1, 2, 3, 2, 2, 3, 2, 3, 2, 3, 2, 3], C_index=[3, 7], low_level_layer=0)  # train.py:80-83
ARCH_C4 = dict(network_arch=[1, 2, 3, 3, 2, 3, 3, 3, 3, 3, 2, 2], C_index=[2, 5, 8], low_level_layer=0)  # train.py:84-87


def make_args(F=20, B=5, sync_bn=False):
    return SimpleNamespace(F=F, B=B, sync_bn=sync_bn)


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


def probe_weights(seed, name, shape):
    """Fixed pseudo-random cotangent used to turn an output into a scalar loss."""
    return rand_tensor(seed, 'probe:' + name, shape)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
