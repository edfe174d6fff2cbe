"""Host-logic tests (CPU, no GPU): build the launch plans with kernel launches stubbed out and check the
structure the planner produces — op counts, virtual-concat wiring, gradient first-touch flags, state_dict
compatibility with the reference layout.  No arithmetic is checked here (that is tests/test_gpu_*.py)."""
import collections

import numpy as np
import pytest
import torch

import addk
import addk.plan as P
import addk.module as M
import oracle
from _util import ARCH_C2, ARCH_C3, GENOTYPE_AUTODEEPLAB, GENOTYPE_BASELINE_2, NETWORK_PATH_BASELINE, make_args


@pytest.fixture()
def dry(monkeypatch):
    """Stub launches; allow CPU tensors.  Plans are built exactly as on the GPU box."""
    calls = collections.Counter()

    def fake_run(self, cmds, stream):
        for name, fn, args in cmds:
            calls[name] += 1
    monkeypatch.setattr(P.Graph, 'run', fake_run)
    monkeypatch.setattr(P, 'require_device', lambda x: None)
    monkeypatch.setattr(P, 'current_stream', lambda: 0)
    return calls


def _add(F=4, arch=ARCH_C2, sync=False):
    from addk.modeling.ADD import ADD
    return ADD(arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(F, sync_bn=sync), arch['low_level_layer'])


def test_state_dict_keys_match_reference_layout():
    """Same keys and shapes as the oracle (which test_oracle_golden pins to the reference's state_dict)."""
    for F, arch in ((4, ARCH_C2), (4, ARCH_C3)):
        m = _add(F, arch)
        o = oracle.ADD(arch['network_arch'], arch['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(F), arch['low_level_layer'])
        sm, so = m.state_dict(), o.state_dict()
        assert set(sm) == set(so)
        for k in so:
            assert tuple(sm[k].shape) == tuple(so[k].shape), k
        m.load_state_dict(so)
    from addk.modeling.baseline_model import Baselin_Model
    b = Baselin_Model(NETWORK_PATH_BASELINE, [5], GENOTYPE_BASELINE_2, 19, make_args(20), 1)
    ob = oracle.Baselin_Model(NETWORK_PATH_BASELINE, [5], GENOTYPE_BASELINE_2, 19, make_args(20), 1)
    assert set(b.state_dict()) == set(ob.state_dict())
    assert len(_add(20).state_dict()) == 1998          # 1998 entries at config 2


def test_dense_conv_weights_are_channels_last():
    m = _add(4)
    w = m.decoder._conv[1].weight
    assert w.shape == (256, 304, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    sd = {k: v.clone().contiguous() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert m.decoder._conv[1].weight.is_contiguous(memory_format=torch.channels_last)   # copy_ keeps the layout


def test_add_plan_structure_train(dry):
    m = _add(4)
    m.train()
    x = torch.randn(2, 3, 65, 129)
    outs = m(x)
    assert len(outs) == 2 and all(tuple(o.shape) == (2, 19, 65, 129) for o in outs)
    plan = next(iter(m._plans().values()))
    names = collections.Counter(n for n, _, _ in plan.g.fwd)
    # 312 BatchNorm applications per forward at C=2, every one a finalize launch in training
    # (mutually independent ones of one dependency level are merged into table-driven batch launches)
    def total(lst, name):
        return sum(1 for c in lst if c.name == name) + sum(c.args[1] for c in lst if c.name == name + '_batch')
    assert total(plan.g.fwd, 'bn_finalize') == 312 and names['bn_finalize'] + names['bn_finalize_batch'] < 200
    # 486 convs in the reference = 168 depthwise + 318 dense; the ASPP image-pool fold adds one tiny GEMM per exit
    assert names['dw_fwd'] == 168
    nconv = names['conv_fwd'] + sum(c.args[1][1] for c in plan.g.fwd if c.name == 'conv_fwd_batch')
    assert nconv == 318 + 2                    # (same-level pointwise convs share batched launches)
    assert names['affine_sum'] == 60           # 12 cells x 5 blocks
    assert names['resize_nchw'] == 2
    # backward: every conv has a wgrad per source, every BN a bn_bwd
    bn = collections.Counter(n for n, _, _ in plan.g.bwd)
    assert total(plan.g.bwd, 'bn_bwd') == 312 and total(plan.g.bwd, 'bn_bwd_apply') == 312
    assert bn['dw_bwd'] == 168 and bn['affine_sum_bwd'] == 60 and 1 <= bn['dw_wreduce_batch'] <= 3
    loss = sum(o.sum() for o in outs)
    loss.backward()
    assert 1 <= dry['conv_wgrad_batch'] <= 40            # ~640 weight gradients in a few batched launches
    assert m.stem0[0].weight.grad is not None and m.aspp.conv1.weight.grad is not None
    assert int(m.aspp.bn1.num_batches_tracked) == 2 and int(m.stem0[1].num_batches_tracked) == 1     # shared head: Q4


def test_add_plan_eval_and_odd_even_resize_counts(dry):
    m = _add(4).eval()
    with torch.no_grad():
        m(torch.randn(1, 3, 65, 129))
        m(torch.randn(1, 3, 64, 128))
    plans = list(m._plans().values())
    r = [collections.Counter(n for n, _, _ in p.g.fwd) for p in plans]
    assert r[0]['bn_eval_affine_batch'] == 1 and r[0]['bn_finalize'] == 0       # 312 BatchNorms, one batched launch
    # even-sized inputs drift off the 2^k+1 ladder and need more resizes (61 vs 46 interpolate calls;
    # the ASPP/decoder-internal ones are not standalone launches here)
    assert r[1]['resize_fwd'] > r[0]['resize_fwd']
    assert not plans[0].g.bwd


def test_c3_uses_conv_aspp_and_three_exits(dry):
    m = _add(4, ARCH_C3).eval()
    with torch.no_grad():
        outs = m(torch.randn(1, 3, 65, 129))
    assert len(outs) == 3 and len(m.conv_aspp) == 1


def test_positional_op_binding_q1(dry):
    """Rows 8,9 of the autodeeplab genotype are [19,7],[18,5]: branch 18 must run _ops[8] (dil_conv_5x5)."""
    from addk.modeling.operations import DilConv, SepConv
    m = _add(4)
    c = m.cells[0]
    assert isinstance(c._ops[8], DilConv) and isinstance(c._ops[9], SepConv)
    import addk.plan as P
    seen = []
    orig_init = P.Graph.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self.bindings = seen                 # emit_blocks records (genotype row, branch index) pairs here
    P.Graph.__init__ = init
    try:
        for mode in ('eval', 'train'):
            del seen[:]
            m.train(mode == 'train')
            with torch.set_grad_enabled(mode == 'train'):
                m(torch.randn(2, 3, 33, 65))
            per_cell = seen[:10]
            # row k of the genotype is bound to the k-th ACTIVE branch in ascending branch order, whatever order the launches of a
            # block are emitted in (inference emits a block's closing SepConv last: it writes the branch sum itself)
            assert per_cell == list(zip(range(10), sorted(int(v) for v in GENOTYPE_AUTODEEPLAB[:, 0]))), (mode, per_cell)
            assert (8, 18) in per_cell and (9, 19) in per_cell
    finally:
        P.Graph.__init__ = orig_init


def test_dynamic_plan_segments(dry):
    from addk.modeling.ADD import EDM
    m = _add(20).eval()
    edm = EDM().eval()
    x = torch.randn(1, 3, 65, 129)
    plan = m._dynamic_plan(x, edm)
    assert len(plan.trunk_end) == 1 and len(plan.heads) == 1 and plan.final is not None
    h0, h1 = plan.head_rng[0]
    assert plan.trunk_end[0] == h0 < h1 < len(plan.g.fwd)


def test_cpu_tensor_fails_loudly():
    m = _add(4).eval()
    with pytest.raises(addk.AddkError):
        m(torch.randn(1, 3, 33, 65))
