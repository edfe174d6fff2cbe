"""Shared by tests/golden/make_golden_fp64.py (which runs the real reference in double precision in the build container) and the GPU
gradient gates that read its fixture tests/golden/grads64.npz: the sampling rule and the seeded targets.  No reference code here."""
import os

import numpy as np
import torch

SAMPLES = 128          # gradient elements kept per parameter tensor (all of them when the tensor is smaller)
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'grads64.npz')


def sample_index(numel):
    """Positions of a flattened tensor the fixture holds: evenly spread, first and last element included."""
    if numel <= SAMPLES:
        return np.arange(numel, dtype=np.int64)
    return np.unique(np.linspace(0, numel - 1, SAMPLES).round().astype(np.int64))


def target(hw, seed=62, n=2, ignore=True):
    """Uniform labels in [0, 19), 5 % set to the ignore index 255 (seed + 1) — the targets of tests/test_gpu_configs.py::_target."""
    t = torch.from_numpy(np.random.default_rng(seed).integers(0, 19, (n,) + tuple(hw))).long()
    if ignore:
        t[torch.from_numpy(np.random.default_rng(seed + 1).random((n,) + tuple(hw)) < 0.05)] = 255
    return t


class Grads64:
    """One case of the fixture: reference-held fp64 gradients (sub-sampled) + whole-tensor statistics."""

    def __init__(self, case):
        z = np.load(PATH)
        pre = case + '/'
        assert pre + 'names' in z.files, 'fixture %s has no case %r (run tests/golden/make_golden_fp64.py %s)' % (PATH, case, case)
        self.case = case
        self.chk = float(z[pre + 'chk'])
        self.loss64, self.loss32 = float(z[pre + 'loss64']), float(z[pre + 'loss32'])
        names = str(z[pre + 'names']).split('\n')
        counts = z[pre + 'counts']
        g64 = torch.from_numpy(z[pre + 'g64']).double()
        d32 = torch.from_numpy(z[pre + 'd32']).double()
        off = np.concatenate([[0], np.cumsum(counts)])
        self.g64 = {n: g64[off[i]:off[i + 1]] for i, n in enumerate(names)}
        self.d32 = {n: d32[off[i]:off[i + 1]] for i, n in enumerate(names)}
        self.stat = {n: z[pre + 'stat'][i] for i, n in enumerate(names)}       # [max |g64|, sum g64^2, sum (g32-g64)^2, max |g32-g64|], whole tensor

    def names(self):
        return list(self.g64)

    @staticmethod
    def sampled(grad):
        """The held positions of a full gradient tensor (any device / dtype), as fp64 on the CPU."""
        flat = grad.detach().reshape(-1)
        idx = torch.from_numpy(sample_index(flat.numel())).to(flat.device)
        return flat[idx].double().cpu()

    def rel_err(self, name, grad):
        """max |g - g64| over the held positions / max |g64| over the WHOLE tensor (tests/_util.rel_err with a sampled numerator)."""
        return float((self.sampled(grad) - self.g64[name]).abs().max() / (self.stat[name][0] + 1e-300))

    def rms_err(self, name, grad):
        """rms of (g - g64) over the held positions / rms of g64 over the whole tensor: every element counts, not only the largest."""
        d = self.sampled(grad) - self.g64[name]
        return float((d ** 2).mean().sqrt() / ((self.stat[name][1] / max(1, grad.numel())) ** 0.5 + 1e-300))

    def ref32_rel_err(self, name):
        """The reference's own fp32 arithmetic, same metric, same positions."""
        return float(self.d32[name].abs().max() / (self.stat[name][0] + 1e-300))

    def ref32_rms_err(self, name, numel):
        return float((self.d32[name] ** 2).mean().sqrt() / ((self.stat[name][1] / max(1, numel)) ** 0.5 + 1e-300))

    # whole-vector statistics over a set of tensors (the train-mode spread test): sampled rel-L2 and cosine against fp64
    def rel_l2(self, grads):
        num = sum(float(((self.sampled(grads[n]) - self.g64[n]) ** 2).sum()) for n in self.g64)
        return (num / sum(float((self.g64[n] ** 2).sum()) for n in self.g64)) ** 0.5

    def cos(self, grads):
        dot = sum(float((self.sampled(grads[n]) * self.g64[n]).sum()) for n in self.g64)
        na = sum(float((self.sampled(grads[n]) ** 2).sum()) for n in self.g64) ** 0.5
        return dot / (na * sum(float((self.g64[n] ** 2).sum()) for n in self.g64) ** 0.5)

    def ref32_rel_l2(self):
        return (sum(float((self.d32[n] ** 2).sum()) for n in self.g64) / sum(float((self.g64[n] ** 2).sum()) for n in self.g64)) ** 0.5

    def ref32_cos(self):
        g32 = {n: self.g64[n] + self.d32[n] for n in self.g64}
        dot = sum(float((g32[n] * self.g64[n]).sum()) for n in self.g64)
        return dot / (sum(float((g32[n] ** 2).sum()) for n in self.g64) ** 0.5 * sum(float((self.g64[n] ** 2).sum()) for n in self.g64) ** 0.5)
