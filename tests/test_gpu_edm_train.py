"""EDM stage (reference train_edm.py:146-185) on the GPU path against the CPU oracle: the recorded (feature, entropy) pairs
and two Adam steps of the L1 regression, including the reference's broadcasting quirk of nn.L1Loss([bs,1], [bs])."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import oracle                                                             # noqa: E402
from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor, rel_err   # noqa: E402


def test_edm_data_and_training_match_the_oracle():
    import addk  # noqa: F401
    from addk.edm_train import EDMTrainer, make_edm_data
    from addk.modeling.ADD import ADD, EDM
    dev = torch.device('cuda:0')
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), 0)
    mo = oracle.ADD(*args); fill_params(mo, 700)
    ma = ADD(*args); ma.load_state_dict(mo.state_dict()); ma.to(dev)
    mo.eval()
    images = [rand_tensor(300 + i, 'edm_img', (1, 3, 65, 129)) for i in range(4)]
    feats, ents = make_edm_data(ma, [{'image': im} for im in images], device=dev)
    fo, eo = [], []
    with torch.no_grad():
        for im in images:
            out, f = mo.get_feature(im)
            fo.append(f); eo.append(oracle.normalized_shannon_entropy(out))
    fo = torch.stack(fo)
    assert tuple(feats.shape) == tuple(fo.shape) and feats.shape[:3] == (4, 1, 400) and tuple(ents.shape) == (4,)
    assert rel_err(feats, fo) <= 1e-3
    assert np.allclose(ents.cpu().numpy(), np.array(eo, dtype=np.float32), rtol=1e-4, atol=1e-6)
    # two Adam steps, reference form: L1Loss()(edm(feature) [bs,1], entropy [bs]) broadcasts to [bs,bs]
    edo = oracle.EDM(); fill_params(edo, 701)
    eda = EDM(); eda.load_state_dict(edo.state_dict()); eda.to(dev)
    tr = EDMTrainer(eda, lr=1e-3)
    opt = torch.optim.Adam(edo.parameters(), lr=1e-3)
    crit = nn.L1Loss()
    ent_o = torch.tensor(eo, dtype=torch.float)
    import warnings
    for _ in range(2):
        la = tr.step(feats, ents)
        edo.train()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            lo = crit(edo(fo.clone()), ent_o)             # the reference's call, shapes and all
        opt.zero_grad(); lo.backward(); opt.step()
        assert abs(float(la) - float(lo)) <= 1e-4 * abs(float(lo)), (float(la), float(lo))
    for (k, a), (_, o) in zip(eda.state_dict().items(), edo.state_dict().items()):
        assert rel_err(a, o) <= 2e-3, k                  # Adam's first steps are sign-like: lr-sized moves on every weight
