"""Stage 3 of the reference: training the Earlier-Decision-Maker (train_edm.py:146-185).

 * `make_edm_data` — train_edm.py:146-167: run the trained multi-exit network in eval mode over a loader and record, per
   batch, the gated feature (`ADD.get_feature`, ADD.py:327-377) and the normalized Shannon entropy of the first exit's
   prediction (operations.py:161-170).  Features stay on the device (the reference copies them to the host and through
   `feature.npy`; `save=` writes the same two files).
 * `EDMTrainer` — train_edm.py:108,169-185: Adam(lr 1e-3) on the EDM, nn.L1Loss between `edm(feature)` ([bs, 1]) and the
   entropies ([bs]).  NB the reference passes those two shapes to L1Loss as they are, which BROADCASTS to a [bs, bs]
   difference matrix (PyTorch warns about it): the loss is mean_ij |out_i - ent_j|, not the per-sample L1.  `faithful=True`
   (default) reproduces that; False trains against the matching target.
The EDM forward/backward run on the addk kernels (modeling.ADD.EDM is a plan module with autograd); the optimizer and the
scalar loss are host-side torch, as in the reference (a 128-64-32-1 MLP: nothing to accelerate)."""
import numpy as np
import torch
import torch.nn as nn

from .modeling.operations import normalized_shannon_entropy


@torch.no_grad()
def make_edm_data(model, loader, device=None, save=None):
    """Returns (features [num_batches, B, C, h, w], entropies [num_batches]) — the tensors train_edm.py:164-165 builds."""
    model.eval()
    feats, ents = [], []
    for sample in loader:
        image = sample['image'] if isinstance(sample, dict) else sample[0]
        if device is not None:
            image = image.to(device, non_blocking=True)
        output, feature = model.get_feature(image)
        ents.append(normalized_shannon_entropy(output))
        feats.append(feature.clone())
    features = torch.stack(feats)
    entropies = torch.tensor(ents, dtype=torch.float32, device=features.device)
    if save is not None:
        np.save(save + 'feature', features.cpu().numpy())
        np.save(save + 'entropy', entropies.cpu().numpy())
    return features, entropies


class EDMTrainer:
    def __init__(self, edm, lr=1e-3, faithful=True):
        self.edm = edm
        self.optimizer = torch.optim.Adam(edm.parameters(), lr=lr)
        self.criterion = nn.L1Loss()
        self.faithful = faithful

    def loss(self, feature, entropy):
        out = self.edm(feature)                          # [bs, 1]; EDM.forward squeezes the stored batch axis (ADD.py:516)
        if self.faithful:
            return (out - entropy.reshape(1, -1)).abs().mean()      # what nn.L1Loss()([bs,1], [bs]) computes after broadcasting
        return self.criterion(out.reshape(-1), entropy.reshape(-1))

    def step(self, feature, entropy):
        self.edm.train()
        loss = self.loss(feature, entropy)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def train_epoch(self, features, entropies, batch_size=16, generator=None):
        """One pass over the recorded set in shuffled batches (train_edm.py:169-185); returns the summed loss it logs."""
        n = features.shape[0]
        perm = torch.randperm(n, generator=generator).to(features.device)
        total = 0.0
        for i in range(0, n, batch_size):
            idx = perm[i:i + batch_size]
            total += float(self.step(features[idx], entropies[idx]))
        return total
