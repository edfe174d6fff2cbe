"""BatchNorm classes injected into every op of the path (reference: modeling/sync_batchnorm/batchnorm.py:38-125,
selected at ADD.py:128).  They are parameter/buffer holders with the reference's state_dict entries
(weight, bias, running_mean, running_var, num_batches_tracked); the arithmetic runs in the addk kernels:
per-channel (sum, sumsq) fused into the producing conv's epilogue, then — for the Synchronized flavour with
world_size > 1 — one RCCL all-reduce of 2C floats over xGMI before the finalize kernel, so every rank
normalises with the statistics of the GLOBAL batch (F.batch_norm on the concatenated batch)."""
import torch.nn as nn


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    sync = True


class SynchronizedBatchNorm1d(nn.BatchNorm1d):
    sync = True


class SynchronizedBatchNorm3d(nn.BatchNorm3d):
    sync = True
