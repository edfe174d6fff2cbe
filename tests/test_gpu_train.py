"""GPU tests of the fused training step (addk.train.TrainStep), the drop-in criterion and the evaluator."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

import oracle                                                   # noqa: E402
from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor, rel_err   # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _models(dev, Fv=4, seed=600):
    from addk.modeling.ADD import ADD
    args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(Fv), 0)
    mo = oracle.ADD(*args)
    fill_params(mo, seed)
    ma = ADD(*args)
    ma.load_state_dict(mo.state_dict())
    return ma.to(dev), mo


def _batch(n, hw, seed=5):
    x = rand_tensor(seed, 'ts_x', (n, 3) + hw)
    r = np.random.default_rng(seed)
    t = torch.from_numpy(r.integers(0, 19, (n,) + hw)).long()
    t[torch.from_numpy(r.random((n,) + hw) < 0.05)] = 255
    return x, t


def _teardown_group(steps):
    """Process-group lifecycle of the world-1 tests: every TrainStep releases its captured graph (whose nodes are RCCL kernels of
    this communicator) and its outstanding gradient-bucket handles, the device drains, and only then the group is destroyed
    (TrainStep.close; the reverse order — communicator first, graphs and handles afterwards — was round 3's teardown)."""
    import torch.distributed as dist
    from addk import parallel
    for ts in steps:
        ts.close()
    del steps[:]
    torch.cuda.synchronize()
    parallel.disable_sync_bn()
    if dist.is_initialized():
        dist.destroy_process_group()


def test_criterion_matches_torch(dev):
    from addk.loss import CrossEntropyLoss
    x = rand_tensor(3, 'ce_x', (2, 19, 33, 65)) * 3
    _, t = _batch(2, (33, 65))
    w = torch.rand(19) + 0.5
    for weight in (None, w):
        xa = x.clone().to(dev).requires_grad_(True)
        xo = x.clone().requires_grad_(True)
        la = CrossEntropyLoss(weight=weight, ignore_index=255).to(dev)(xa, t.to(dev))
        lo = nn.CrossEntropyLoss(weight=weight, ignore_index=255)(xo, t)
        (la * 0.5).backward(); (lo * 0.5).backward()
        assert abs(la.item() - lo.item()) < 1e-5 * abs(lo.item())
        assert rel_err(xa.grad, xo.grad) < 1e-5
    # generic (non-19) class count path
    x7 = rand_tensor(3, 'ce_x7', (1, 7, 9, 11)); t7 = torch.randint(0, 7, (1, 9, 11))
    xa = x7.clone().to(dev).requires_grad_(True); xo = x7.clone().requires_grad_(True)
    la = CrossEntropyLoss().to(dev)(xa, t7.to(dev)); lo = nn.CrossEntropyLoss(ignore_index=255)(xo, t7)
    la.backward(); lo.backward()
    assert abs(la.item() - lo.item()) < 1e-5 and rel_err(xa.grad, xo.grad) < 1e-5

@pytest.mark.parametrize('lo_hw,hi_hw', [((16, 32), (128, 256)), ((17, 33), (65, 129)), ((9, 17), (65, 129)), ((33, 65), (33, 65)),
                                         ((129, 257), (1025, 2049))], ids=['x8', 'x3.8', 'x7.6', 'x1', 'config4_odd'])
def test_fused_upsample_cross_entropy_matches_fp64_reference(dev, lo_hw, hi_hw):
    """addk_ce_upsample_fwd_bwd (the training step's loss head) against F.interpolate(bilinear, align_corners=False) +
    nn.CrossEntropyLoss(weight, ignore_index) + autograd in fp64: loss and d loss / d low-resolution logits, with class
    weights, ignored pixels, a padded pixel stride, accumulation into an existing gradient, and run-to-run bit equality."""
    import ctypes as C
    import torch.nn.functional as Fn
    import addk._lib as L
    lib = L.load()
    N, (H, W), (OH, OW), ld = 2, lo_hw, hi_hw, 24
    assert lib.addk_ce_upsample_supported(N, H, W, OH, OW, 19) == 1
    x = rand_tensor(31, 'ceu_x', (N, H, W, 19)) * 3
    _, t = _batch(N, hi_hw, seed=9)
    w = torch.rand(19, generator=torch.Generator().manual_seed(4)) + 0.5
    g0 = rand_tensor(32, 'ceu_g0', (N, H, W, ld))
    for weight, acc in ((None, 0), (w, 1)):
        x64 = x.double().requires_grad_(True)
        up = Fn.interpolate(x64.permute(0, 3, 1, 2), size=hi_hw, mode='bilinear', align_corners=False)
        lo = nn.CrossEntropyLoss(weight=None if weight is None else weight.double(), ignore_index=255)(up, t) * 0.5
        lo.backward()
        # ATen evaluates the source coordinates in the tensor's dtype: at non-integer scales the fp32 interpolation weights
        # differ from the fp64 ones by ~1e-5 (coordinate ~1e2 x 2^-24), so the fp32 reference is the tighter anchor there
        x32 = x.clone().requires_grad_(True)
        up32 = Fn.interpolate(x32.permute(0, 3, 1, 2), size=hi_hw, mode='bilinear', align_corners=False)
        (nn.CrossEntropyLoss(weight=weight, ignore_index=255)(up32, t) * 0.5).backward()
        xa = torch.zeros((N, H, W, ld), device=dev); xa[..., :19] = x.to(dev)
        ta = t.to(dev); wa = weight.to(dev) if weight is not None else None
        wsum = torch.zeros(1, device=dev); ws1 = torch.zeros(int(lib.addk_ce_ws_floats(N, OH * OW)), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        L.check(lib.addk_ce_count(ta.data_ptr(), N * OH * OW, wa.data_ptr() if wa is not None else None, 255, 19, wsum.data_ptr(), ws1.data_ptr(), st))
        res = []
        for rep in range(2):
            loss = torch.full((1,), 0.25, device=dev)
            g = g0.clone().to(dev)
            ws = torch.zeros(int(lib.addk_ce_upsample_ws_floats(N, H, W)), device=dev)
            a = L.CeUpsampleArgs()
            a.logits, a.ld, a.N, a.H, a.W, a.C, a.OH, a.OW = xa.data_ptr(), ld, N, H, W, 19, OH, OW
            a.target, a.class_w, a.ignore_index = ta.data_ptr(), wa.data_ptr() if wa is not None else None, 255
            a.wsum, a.scale, a.loss_out = wsum.data_ptr(), 0.5, loss.data_ptr()
            a.g, a.ldg, a.accumulate, a.ws = g.data_ptr(), ld, acc, ws.data_ptr()
            L.check(lib.addk_ce_upsample_fwd_bwd(C.byref(a), st))
            torch.cuda.synchronize()
            res.append((loss.cpu(), g.cpu()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])        # deterministic
        la, ga = res[0]
        assert abs(float(la) - 0.25 - float(lo)) <= 2e-6 * abs(float(lo)), (float(la) - 0.25, float(lo))
        base = g0[..., :19].double() if acc else 0
        assert rel_err(ga[..., :19].double(), x64.grad + base) <= 5e-5
        assert rel_err(ga[..., :19].double(), x32.grad.double() + base) <= 1e-5
        assert torch.equal(ga[..., 19:], g0[..., 19:])                                            # padding channels untouched
    assert lib.addk_ce_upsample_supported(N, H, W, OH, OW, 7) == 0 and lib.addk_ce_upsample_supported(1, 4, 4, 128, 128, 19) == 0


def test_train_step_forward_backward_and_sgd(dev):
    from addk.train import TrainStep
    ma, mo = _models(dev)
    m64 = oracle.ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4), 0)
    m64.load_state_dict(mo.state_dict()); m64.double()
    x, t = _batch(2, (65, 129))
    ts = TrainStep(ma, (2, 3, 65, 129), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, use_graph=False)
    ts.load_batch(x.to(dev), t.to(dev))
    ts.forward_backward_only()
    torch.cuda.synchronize()
    res = {}
    for name, m, xx in (('o32', mo, x), ('o64', m64, x.double())):
        m.train()
        loss = oracle.cross_entropy_mean_exits(m(xx), t)
        loss.backward()
        res[name] = (loss.item(), {k: p.grad.double() for k, p in m.named_parameters()})
    la = ts.loss.item()
    assert abs(la - res['o64'][0]) <= 3 * abs(res['o32'][0] - res['o64'][0]) + 1e-4 * abs(la)
    names = {p: k for k, p in ma.named_parameters()}
    ga = {names[p]: g.double().cpu() for p, g in ts.grads().items() if g is not None}
    g64 = res['o64'][1]

    def rel_l2(gx):
        return (sum(float(((gx[k] - g64[k]) ** 2).sum()) for k in g64) / sum(float((g64[k] ** 2).sum()) for k in g64)) ** 0.5
    assert set(ga) == set(g64)
    assert rel_l2(ga) <= 2.0 * rel_l2(res['o32'][1]) + 1e-3
    # the fused SGD kernel against torch.optim.SGD semantics applied to the SAME gradients (exact arithmetic check)
    p0 = ts.flat_p.clone(); g0 = ts.flat_g.clone()
    ts2_before = ts.mom_buf.clone()
    assert float(ts2_before.abs().max()) == 0.0
    ts.step()            # eager: fwd+bwd (same inputs -> same grads) + sgd
    torch.cuda.synchronize()
    d = g0 + 4e-5 * p0
    buf = d
    expect = p0 - 0.05 * (d + 0.9 * buf)
    # BN running statistics changed between the two forwards, gradients did not (training BN ignores them)
    assert rel_err(ts.flat_g, g0) < 1e-6
    assert rel_err(ts.flat_p, expect) < 1e-6
    assert rel_err(ts.mom_buf, buf) < 1e-6


def test_train_step_fused_loss_head_equals_three_kernel_form(dev, monkeypatch):
    """ADDK_FUSE_CE=0 keeps resize -> cross-entropy -> resize backward as three launches over full-resolution tensors; the
    default fused head must give the same loss and gradients (same arithmetic, different summation order)."""
    from addk.train import TrainStep
    x, t = _batch(2, (65, 129))
    out = {}
    for fuse in ('0', '1'):
        monkeypatch.setenv('ADDK_FUSE_CE', fuse)
        ma, _ = _models(dev)
        ts = TrainStep(ma, (2, 3, 65, 129), use_graph=False)
        names = [c.name for c in ts.g.bwd]
        assert ('ce_upsample' in names) == (fuse == '1') and ('resize_nchw_bwd' in names) == (fuse == '0')
        ts.load_batch(x.to(dev), t.to(dev))
        ts.forward_backward_only()
        torch.cuda.synchronize()
        out[fuse] = (ts.loss.item(), ts.flat_g.clone())
    assert abs(out['0'][0] - out['1'][0]) <= 2e-6 * abs(out['0'][0])
    assert rel_err(out['1'][1], out['0'][1]) <= 2e-4


def test_train_step_hipgraph_equals_eager(dev):
    from addk.train import TrainStep
    losses = {}
    for mode in (False, True):
        ma, _ = _models(dev)
        x, t = _batch(2, (33, 65))
        ts = TrainStep(ma, (2, 3, 33, 65), use_graph=mode)
        ts.load_batch(x.to(dev), t.to(dev))
        ls = []
        for _ in range(4):
            ls.append(ts.step().item())
        losses[mode] = ls
        assert (ts.graph is not None) == mode
    assert losses[True] == losses[False], (losses[True], losses[False])      # deterministic kernels: bitwise equal
    assert losses[True][-1] < losses[True][0]                                # and the loss goes down


def test_drop_in_step_replays_two_hipgraphs_and_equals_eager(dev, monkeypatch):
    """The reference's own call pattern (train.py:227-240: outputs = model(image); loss.backward(); optimizer.step()) on the addk
    modules: from the third call on the forward list and the backward list of the plan are replayed as two hipGraphs (module.Plan).
    Losses of five SGD steps and the final parameters are bitwise those of the eager launch loop (ADDK_GRAPH_MODULE=0); a batch
    whose data changes between steps goes through the plan-owned input / gradient buffers correctly."""
    from addk.loss import CrossEntropyLoss
    res = {}
    real_graph = torch.cuda.graph

    class _Refuse:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            raise RuntimeError('capture refused (test stub)')

        def __exit__(self, *a):
            return False
    for mode in ('1', '0', 'refused'):
        monkeypatch.setenv('ADDK_GRAPH_MODULE', '0' if mode == '0' else '1')
        monkeypatch.setattr(torch.cuda, 'graph', _Refuse if mode == 'refused' else real_graph)
        ma, _ = _models(dev)
        ma.train()
        crit = CrossEntropyLoss(ignore_index=255)
        opt = torch.optim.SGD(ma.parameters(), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)
        ls = []
        import warnings
        with warnings.catch_warnings(record=True) as wlog:
            warnings.simplefilter('always')
            for i in range(5):
                x, t = _batch(2, (33, 65), seed=5 + (i % 2))          # two alternating batches: replay must pick up new input data
                ys = ma(x.to(dev))
                loss = sum(crit(y, t.to(dev)) for y in ys) / len(ys)
                opt.zero_grad()
                loss.backward()
                opt.step()
                ls.append(loss.item())
        plans = [p for m in ma.modules() for p in getattr(m, '__dict__', {}).get('_addk_plans', {}).values()]
        assert plans and all((p.graph is not None and p.bgraph is not None) == (mode == '1') for p in plans if p.g.want_grad)
        if mode == 'refused':        # a refused capture is reported once and the plan keeps the eager launch loop
            assert sum('capture of the forward list refused' in str(w.message) for w in wlog) == 1
        res[mode] = (ls, torch.cat([p.detach().reshape(-1) for p in ma.parameters()]).cpu())
    monkeypatch.setattr(torch.cuda, 'graph', real_graph)
    assert res['refused'][0] == res['0'][0] and torch.equal(res['refused'][1], res['0'][1])
    assert res['1'][0] == res['0'][0], (res['1'][0], res['0'][0])
    assert torch.equal(res['1'][1], res['0'][1])


@pytest.mark.parametrize('size', [(33, 65), (256, 512)], ids=['33x65', '256x512'])
def test_schedule_and_batching_do_not_change_a_bit(dev, size, monkeypatch):
    """The same step as (a) the plain sequential launch list on one stream and (b) the level-ordered list with table-driven
    batched launches on two streams must produce bit-identical losses and gradients: every kernel's arithmetic is
    independent of the schedule and gradient accumulation order is fixed by the dependency chain, so any difference is a
    missing dependency (a race) or a batched kernel that differs from its single form.  256x512 reaches the tiled
    depthwise, register-streaming and batched pointwise kernels."""
    from addk.train import TrainStep
    res = {}
    for tag, streams, level in (('plain', '1', '0'), ('scheduled', '2', '1')):
        monkeypatch.setenv('ADDK_STREAMS', streams)
        monkeypatch.setenv('ADDK_LEVEL_BATCH', level)
        ma, _ = _models(dev, Fv=20 if size[0] > 100 else 4)
        x, t = _batch(2, size)
        ts = TrainStep(ma, (2, 3) + size, use_graph=False)
        ts.load_batch(x.to(dev), t.to(dev))
        ts.forward_backward_only()
        torch.cuda.synchronize()
        g0 = ts.flat_g.clone()
        ls = [ts.step().item() for _ in range(2)]
        res[tag] = (ts.loss.item(), g0, ls, len(ts.g.fwd) + len(ts.g.bwd))
    assert res['scheduled'][3] < res['plain'][3]                  # the batching pass merged launches
    assert torch.equal(res['plain'][1], res['scheduled'][1]), float((res['plain'][1] - res['scheduled'][1]).abs().max())
    assert res['plain'][2] == res['scheduled'][2]


def test_evaluator_and_argmax(dev, golden):
    from addk.metrics import Evaluator, argmax_logits
    g = golden('misc')
    ev = Evaluator(19, dev)
    gt = torch.from_numpy(g['eval/gt'].astype(np.int64)); pr = torch.from_numpy(g['eval/pred'].astype(np.int64))
    ev.add_batch(gt.to(dev), pr.to(dev))
    assert np.array_equal(ev.confusion_matrix.cpu().numpy(), g['eval/cm'])
    assert abs(ev.Mean_Intersection_over_Union() - float(g['eval/miou'])) < 1e-6
    assert abs(float(ev.Pixel_Accuracy()) - float(g['eval/pa'])) < 1e-6
    assert abs(float(ev.Frequency_Weighted_Intersection_over_Union()) - float(g['eval/fwiou'])) < 1e-6
    x = rand_tensor(4, 'am', (2, 19, 17, 33))
    assert torch.equal(argmax_logits(x.to(dev)).cpu(), x.argmax(1))


def test_syncbn_path_world1_matches_local_bn(dev):
    """The N>1 code path on one GPU: SynchronizedBatchNorm2d + RCCL (world_size 1, exchange forced) must reproduce the
    local-BatchNorm step (slab_reduce -> all_reduce -> bn_finalize, bn_bwd -> all_reduce(dmean,dvar) -> coefficients,
    flat-gradient all-reduce) to rounding."""
    import os
    import torch.distributed as dist
    from addk import parallel
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29611')
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        x, t = _batch(2, (65, 129))
        res, live = {}, []
        for sync in (False, True):
            args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=sync), 0)
            mo = oracle.ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4), 0)
            fill_params(mo, 600)
            m = ADD(*args)
            m.load_state_dict(mo.state_dict())
            m.to(dev)
            comm = parallel.init_sync_bn(force=True) if sync else None
            ts = TrainStep(m, (2, 3, 65, 129), use_graph=False, sync_comm=comm)
            live.append(ts)
            ts.load_batch(x.to(dev), t.to(dev))
            ts.forward_backward_only()
            g0, l0 = ts.flat_g.clone(), ts.loss.item()
            losses = [ts.step().item() for _ in range(2)]
            res[sync] = (l0, g0, losses, comm.calls if comm else 0)
            parallel.disable_sync_bn()
        # 312 forward + 312 backward exchanges per pass; those of one dependency level share a grouped RCCL call
        assert 3 * 150 <= res[True][3] < 3 * 624
        assert abs(res[True][0] - res[False][0]) < 1e-6 * abs(res[False][0])           # forward: same statistics
        gs, gl = res[True][1].double(), res[False][1].double()
        assert float((gs - gl).norm() / gl.norm()) < 1e-3                                # backward: same gradient (rounding only)
        # later steps drift apart chaotically (fp32 rounding of the exchanged (dmean, dvar) amplified by the network)
        assert abs(res[True][2][0] - res[False][2][0]) < 1e-6 * abs(res[False][2][0])
        assert abs(res[True][2][1] - res[False][2][1]) < 1e-3 * abs(res[False][2][1])
    finally:
        _teardown_group(live)


def test_syncbn_captured_step_world1_equals_eager_and_capture_failure_applies_one_update(dev, monkeypatch):
    """The DEFAULT path of a step with collectives at world 1 (forced exchanges): the whole step incl. its RCCL calls captured
    in one hipGraph in thread-local capture mode (train.TrainStep._capture) — loss and parameters must equal the eager
    launch list bit for bit over three steps (regression test of the capture abort fixed in round 2).  Then the fallback: a
    capture that raises must leave exactly ONE update applied by that call (it used to run the step twice)."""
    import os
    import torch.distributed as dist
    from addk import parallel
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29612')
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        x, t = _batch(2, (65, 129))
        live = []

        def build(**kw):
            args = (ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=True), 0)
            m = ADD(*args)
            fill_params(m, 600)
            m.to(dev)
            comm = parallel.init_sync_bn(force=True)
            ts = TrainStep(m, (2, 3, 65, 129), sync_comm=comm, **kw)
            live.append(ts)
            ts.load_batch(x.to(dev), t.to(dev))
            return ts
        eager = build(use_graph=False)
        le = [eager.step().item() for _ in range(3)]
        pe = eager.flat_p.clone()
        cap = build()                                  # default: captured (world == 1)
        assert cap.use_graph and cap.has_coll
        lc = [cap.step().item() for _ in range(3)]
        assert cap.graph is not None, 'the collective step was not captured'
        assert lc == le, (lc, le)
        assert torch.equal(cap.flat_p, pe)
        # a runtime that refuses the capture: eager replay from then on, ONE update per call
        one = build(use_graph=False)
        l1 = one.step().item()
        p1 = one.flat_p.clone()

        class _Refuse:
            def __init__(self, *a, **k):
                pass

            def __enter__(self):
                raise RuntimeError('capture refused (test stub)')

            def __exit__(self, *a):
                return False
        bad = build()
        monkeypatch.setattr(torch.cuda, 'graph', _Refuse)
        lb = bad.step().item()
        monkeypatch.undo()
        assert not bad.use_graph and bad.graph is None and bad.steps == 1
        assert lb == l1 and torch.equal(bad.flat_p, p1), 'the fallback applied a second update'
        assert bad.step().item() == le[1]              # and continues on the eager list
    finally:
        _teardown_group(live)
