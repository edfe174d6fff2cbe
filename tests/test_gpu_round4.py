"""GPU tests of the round-4 kernels, all through the C ABI.

* the 1x1 convolutions that sample a bilinear resize themselves (addk_src.rs_hw, pw.hip RS forms; reference ADD.py:76-77,84-90:
  F.interpolate in front of `preprocess` / `pre_preprocess`): BIT-identical to addk_resize_fwd followed by the plain launch, and both
  against an fp64 F.interpolate + conv2d; the training copy-out equals the stand-alone resize's output bit for bit;
* whole cells / the whole network with the fold on and off give identical outputs and gradients."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    assert torch.cuda.is_available()
    import addk  # noqa: F401
    from addk import _lib as L
    L.load()
    return L


@pytest.fixture()
def dev():
    return torch.device('cuda:0')


# name, N, (source H, W), (consumer H, W), K, Cout, lazy BN on the source, extra padded channels (ld > C), batched
RS_SHAPES = [
    ('up2_odd_k80', 2, (32, 64), (63, 127), 80, 80, True, 0),        # pw_kernel KG = 5 (dense features of level 2 into an even-size-drifted map)
    ('up2_k40', 1, (63, 127), (125, 253), 40, 40, True, 0),          # pw_kernel KG = 3, CT = 3
    ('down2_k40', 2, (125, 253), (63, 127), 40, 80, False, 0),       # down-sampling (ADD.py:84: fit() runs both ways)
    ('down4_k128', 1, (128, 256), (32, 64), 128, 80, True, 0),       # pwk_kernel: cell 1's prev_prev (stem2 output), 4x down
    ('up_k800', 1, (16, 32), (31, 63), 800, 160, True, 0),           # pwk_kernel, K = 800 (level-3 concat into level 2)
    ('same_w_k200', 2, (33, 64), (65, 64), 200, 40, True, 0),        # only H differs (fit() compares H alone)
    ('ld_pad_k80', 1, (20, 40), (39, 79), 80, 40, True, 8),          # source is a channel slice of a wider buffer
    ('tiny_tail_k48', 1, (9, 17), (33, 65), 48, 48, True, 0),        # P = 2145: not a multiple of 16
]


def _mk_src(L, x, a, b, C_, ld, relu, rs_hw=0):
    s = L.Src()
    s.x, s.ld, s.C, s.relu, s.rs_hw = x.data_ptr(), ld, C_, int(relu), rs_hw
    if a is not None:
        s.a, s.b = a.data_ptr(), b.data_ptr()
    return s


@pytest.mark.parametrize('shape', RS_SHAPES, ids=[s[0] for s in RS_SHAPES])
def test_conv1x1_samples_resize_itself_bit_identical(L, dev, shape):
    lib = L.load()
    name, N, (SH, SW), (H, W), K, Cout, lazy, padc = shape
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    ld = K + padc
    xs = torch.randn(N, SH, SW, ld, generator=g).to(dev)
    a = (torch.rand(K, generator=g) + 0.5).to(dev) if lazy else None
    b = (torch.randn(K, generator=g) * 0.3).to(dev) if lazy else None
    w = (torch.randn(Cout, K, generator=g) / K ** 0.5).to(dev)
    P = N * H * W
    rows = lib.addk_conv_rows(P, Cout)

    def conv(src, rs_y=None):
        ar = L.ConvArgs()
        ar.src[0] = src
        ar.nsrc = 1
        ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, H, W
        ar.KH = ar.KW = 1
        ar.stride, ar.pad, ar.dil, ar.Cout = 1, 0, 1, Cout
        ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = K, K, 0, Cout
        y = torch.full((P, Cout), float('nan'), device=dev)
        slab = torch.zeros(rows, Cout, 2, device=dev, dtype=torch.float64)
        ar.w, ar.y, ar.stats, ar.stats_ld = w.data_ptr(), y.data_ptr(), slab.data_ptr(), Cout
        if rs_y is not None:
            ar.rs_y, ar.rs_ldy = rs_y.data_ptr(), K
        if src.rs_hw:
            assert lib.addk_conv_fwd_resample_ok(C.byref(ar)) == 1, 'the pointwise kernels should cover %s' % name
        L.check(lib.addk_conv_fwd(C.byref(ar), st), 'conv_fwd')
        return y, slab.sum(0)

    # unfused: stand-alone resize of the raw map, then the plain 1x1 launch with the lazy affine + ReLU
    rz = torch.full((P, K), float('nan'), device=dev)
    ra = L.ResizeArgs()
    ra.src = _mk_src(L, xs, None, None, K, ld, False)
    ra.N, ra.H, ra.W, ra.OH, ra.OW = N, SH, SW, H, W
    ra.y, ra.ldy, ra.nchw_out = rz.data_ptr(), K, 0
    L.check(lib.addk_resize_fwd(C.byref(ra), st), 'resize_fwd')
    y0, s0 = conv(_mk_src(L, rz, a, b, K, K, True))
    # folded, with and without the training copy-out
    copy = torch.full((P, K), float('nan'), device=dev)
    y1, s1 = conv(_mk_src(L, xs, a, b, K, ld, True, (SH << 16) | SW))
    y2, s2 = conv(_mk_src(L, xs, a, b, K, ld, True, (SH << 16) | SW), rs_y=copy)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and torch.equal(y2, y0), 'folded 1x1 differs from resize + 1x1 (max %.3e)' % float((y1 - y0).abs().max())
    assert torch.equal(s1, s0) and torch.equal(s2, s0)
    assert torch.equal(copy, rz), 'the copy-out is not the resized tensor'
    # and against fp64 torch: F.interpolate(bilinear, align_corners=False) -> affine -> ReLU -> 1x1
    xd = xs[..., :K].permute(0, 3, 1, 2).double().cpu()
    r = F.interpolate(xd, size=(H, W), mode='bilinear', align_corners=False)
    if lazy:
        r = r * a.double().cpu().view(1, K, 1, 1) + b.double().cpu().view(1, K, 1, 1)
    ref = F.conv2d(r.relu(), w.double().cpu().view(Cout, K, 1, 1)).permute(0, 2, 3, 1).reshape(P, Cout)
    err = float((y1.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_resample_is_refused_where_no_kernel_covers_it(L, dev):
    """A 3x3 convolution, a wide (256-channel) head or a tiny map must not be handed a sampled source: the query says no, and
    a launch that insists fails loudly instead of reading the small map as if it had the consumer's size."""
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    xs = torch.randn(1, 8, 8, 80, device=dev)
    w = torch.randn(256, 9 * 80, device=dev)
    y = torch.empty(40 * 40, 256, device=dev)
    # (kernel size, K, Cout, consumer map side, covered?)
    for k, K, cout, oh, covered in ((3, 80, 80, 15, False),       # not a 1x1
                                    (1, 64, 256, 40, False),      # K = 64 is the streaming kernel's, which stops at 160 output channels
                                    (1, 64, 80, 15, False),       # ... and needs >= 1024 pixels
                                    (1, 80, 80, 15, True)):       # the register-stationary kernel takes any map size
        ar = L.ConvArgs()
        ar.src[0] = _mk_src(L, xs, None, None, K, 80, True, (8 << 16) | 8)
        ar.nsrc = 1
        ar.N, ar.H, ar.W, ar.OH, ar.OW = 1, oh, oh, oh, oh
        ar.KH = ar.KW = k
        ar.stride, ar.pad, ar.dil, ar.Cout = 1, k // 2, 1, cout
        ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = k * k * K, K, 0, cout
        ar.w, ar.y = w.data_ptr(), y.data_ptr()
        ok = lib.addk_conv_fwd_resample_ok(C.byref(ar))
        assert ok == int(covered), (k, K, cout, oh)
        rc = lib.addk_conv_fwd(C.byref(ar), st)
        if covered:
            assert rc == 0
        else:
            assert rc < 0 and b'resampled' in lib.addk_last_error()
    torch.cuda.synchronize()


def _cell_net(dev, F_=20, seed=11):
    import oracle  # noqa: F401
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args
    from addk.modeling.ADD import ADD
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(F_), 0)
    fill_params(m, seed)
    return m.to(dev)


def test_whole_network_fold_on_equals_fold_off(dev, monkeypatch):
    """ADD F=20 at an even size (the size class whose dense connections need resizes): eval logits, train-mode logits,
    loss and every parameter gradient are IDENTICAL with the resizes folded into their 1x1 consumers and with stand-alone launches;
    the folded plans hold (almost) no resize launch."""
    from _util import rand_tensor
    from addk.train import TrainStep
    x = rand_tensor(3, 'fold_x', (2, 3, 256, 512)).to(dev)
    t = torch.from_numpy(np.random.default_rng(4).integers(0, 19, (2, 256, 512))).long().to(dev)
    res = {}
    for fold in ('1', '0'):
        monkeypatch.setenv('ADDK_FOLD_RESIZE', fold)
        m = _cell_net(dev)
        m.eval()
        with torch.no_grad():
            ye = [y.clone() for y in m(x)]
        plan = next(iter(m._plans().values()))
        n_rs_eval = sum(1 for c in plan.g.fwd if c.name == 'resize_fwd')
        ts = TrainStep(m, (2, 3, 256, 512), use_graph=False)
        ts.load_batch(x, t)
        ts.forward_backward_only()
        torch.cuda.synchronize()
        n_rs_train = sum(1 for c in ts.g.fwd if c.name == 'resize_fwd')
        res[fold] = (ye, float(ts.loss.item()), ts.flat_g.clone(), n_rs_eval, n_rs_train)
        del ts, m
        torch.cuda.empty_cache()
    a, b = res['1'], res['0']
    assert a[3] < b[3] - 30 and a[4] < b[4] - 30, 'the fold removed too few resize launches: %s vs %s' % (a[3:], b[3:])
    for ya, yb in zip(a[0], b[0]):
        assert torch.equal(ya, yb)
    assert a[1] == b[1]
    assert torch.equal(a[2], b[2])


@pytest.mark.parametrize('shape', [(1, 64, 128), (1, 65, 129), (2, 33, 65), (3, 8, 12)], ids=['config4_1024x2048', 'config4_1025x2049', 'bs2_odd', 'bs3_tiny'])
def test_fused_edm_head_matches_oracle_and_generic_path(dev, shape, monkeypatch):
    """The Earlier-Decision-Maker head as ONE launch (csrc/edm.hip; reference ADD.py:502-525) against oracle.EDM (pinned to the reference by
    tests/golden/dynamic.npz) and against the five generic launches it replaces; the plan holds a single `edm_head` command, the result is
    bit-identical run to run (the last-arriving workgroup adds the partial rows in a fixed order) and the ticket word is back at zero."""
    import oracle
    from _util import fill_params
    from addk.modeling.ADD import EDM
    N, H, W = shape
    eo = oracle.EDM()
    fill_params(eo, 701)
    eo.eval()
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, 400, H, W, generator=g)
    with torch.no_grad():
        ref = eo(x.clone())
    outs = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('ADDK_FUSE_EDM', fuse)
        ea = EDM()
        ea.load_state_dict(eo.state_dict())
        ea.to(dev).eval()
        with torch.no_grad():
            y1 = ea(x.clone().to(dev)).clone()
            y2 = ea(x.clone().to(dev)).clone()
        plan = next(iter(ea._plans().values()))
        names = [c.name for c in plan.g.fwd]
        if fuse == '1':
            assert names.count('edm_head') == 1 and not any(n.startswith(('conv', 'gap')) for n in names), names
        else:
            assert 'edm_head' not in names
        assert torch.equal(y1, y2)
        outs[fuse] = y1.cpu()
    assert tuple(outs['1'].shape) == tuple(ref.shape) == (N, 1)
    e_ref = float((outs['1'].double() - ref.double()).abs().max() / ref.double().abs().max())
    e_gen = float((outs['1'].double() - outs['0'].double()).abs().max() / ref.double().abs().max())
    assert e_ref < 1e-4 and e_gen < 1e-4, (e_ref, e_gen)


def test_dynamic_inference_gate_reads_the_pinned_word_the_fused_head_writes(dev):
    """config 4 plumbing: DynamicPlan hands the fused head a pinned host word; after the trunk segment's event the host reads the SAME value
    the device tensor holds, for the eager calls and for the captured hipGraph replays alike, and the exit decision follows the threshold."""
    import oracle
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args, rand_tensor
    from addk.modeling.ADD import ADD, EDM
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(20), 0)
    fill_params(m, 12)
    m.to(dev).eval()
    e = EDM(); fill_params(e, 701); e.to(dev).eval()
    x = rand_tensor(9, 'dynx', (1, 3, 129, 257)).to(dev)
    with torch.no_grad():
        confs = []
        for call in range(5):                      # calls 3+ replay captured graphs
            y, early, secs, conf = m.dynamic_inference(x, threshold=1e9, confidence='edm', edm=e)
            plan = m._dynamic_plan(x, e)
            assert plan.conf_fused == [True]
            assert early == 0 or early == 1
            confs.append(float(conf.reshape(-1)[0]))
            assert float(plan._conf_host[0]) == confs[-1], (float(plan._conf_host[0]), confs[-1])
        assert len(set(confs)) == 1
        c = confs[0]
        # ADD.py:421: `if confidence_value > threshold` the image goes ON to the next cells; otherwise it leaves at this exit
        assert m.dynamic_inference(x, threshold=c - 1.0, confidence='edm', edm=e)[1] == 0
        assert m.dynamic_inference(x, threshold=c + 1.0, confidence='edm', edm=e)[1] == 1


# ---- data gradient of the classifier (1x1, 256 -> 19): the register-weights / readlane kernel (pw.hip k1s_dgrad_kernel) -------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 33, 65, 256, 19, True, False), (1, 40, 50, 128, 19, False, True), (2, 17, 31, 256, 32, True, True),
                                   (1, 128, 256, 256, 19, True, False)], ids=['c256_k19', 'c128_noaffine_acc', 'c256_k32_acc', 'classifier_full_map'])
def test_classifier_data_gradient_matches_fp64_and_the_generic_kernel(shape):
    """decoder.py last_conv backward: g = relu'(a x + b) * a * (dy W), (dA, dB) = sum over pixels of (m dz x, m dz); odd pixel counts (a wave's
    second pixel of the last trip missing), accumulate, no lazy BatchNorm on the destination; against fp64 and against ADDK_K1S-less dispatch."""
    import ctypes as C
    import torch
    from addk import _lib as L
    N, H, W, Cn, K, affine, accumulate = shape
    lib = L.load()
    dev = torch.device('cuda:0')
    torch.manual_seed(sum(map(int, shape[:5])))
    P = N * H * W
    dy = torch.randn(P, K, device=dev)
    w = 0.1 * torch.randn(K, Cn, device=dev)
    x = torch.randn(P, Cn, device=dev)
    a = torch.rand(Cn, device=dev) + 0.5
    b = 0.2 * torch.randn(Cn, device=dev)
    g0 = torch.randn(P, Cn, device=dev)
    rows = lib.addk_conv_rows(P, Cn)

    def run():
        da = L.ConvDgradArgs()
        da.dy, da.lddy, da.Cout = dy.data_ptr(), K, K
        da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, H, W, 1, 1, 1, 0, 1
        da.w, da.ldw, da.cin_total, da.w_choff = w.data_ptr(), Cn, Cn, 0
        da.dst.x, da.dst.ld, da.dst.C, da.dst.relu = x.data_ptr(), Cn, Cn, 1
        if affine:
            da.dst.a, da.dst.b = a.data_ptr(), b.data_ptr()
        g = g0.clone() if accumulate else torch.full((P, Cn), float('nan'), device=dev)
        dab = torch.full((rows, Cn, 2), float('nan'), device=dev, dtype=torch.float64)
        da.g, da.ldg, da.accumulate, da.dab = g.data_ptr(), Cn, int(accumulate), dab.data_ptr()
        L.check(lib.addk_conv_dgrad(C.byref(da), torch.cuda.current_stream().cuda_stream), 'conv_dgrad')
        torch.cuda.synchronize()
        return g, dab.sum(0)
    g, dab = run()
    a64, b64 = (a.double(), b.double()) if affine else (torch.ones(Cn, device=dev, dtype=torch.float64), torch.zeros(Cn, device=dev, dtype=torch.float64))
    dz = dy.double() @ w.double()
    m = (a64 * x.double() + b64) > 0
    ref_g = torch.where(m, dz * a64, torch.zeros_like(dz)) + (g0.double() if accumulate else 0)
    ref_dab = torch.stack([(dz * x.double() * m).sum(0), (dz * m).sum(0)], 1)
    assert torch.isfinite(g).all() and torch.isfinite(dab).all()
    assert float((g.double() - ref_g).abs().max() / ref_g.abs().max()) < 2e-6
    assert float((dab - ref_dab).abs().max() / ref_dab.abs().max()) < 2e-6


# ---- backward of the cell block sum (ADD.py:108): the two-pixels-per-trip vector form (elementwise.hip affine_sum_bwd_vec_kernel) ----------------
@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(16002, 80, 2, False, (1, 0), (0, 0)), (63250, 40, 2, True, (1, 1), (0, 1)), (4097, 160, 3, False, (0, 1, 1), (0, 0, 1)),
                                   (777, 40, 4, True, (1, 1, 1, 1), (1, 0, 1, 0)), (16002, 80, 1, False, (1,), (1,))],
                         ids=['l2_two_terms', 'l1_relu_out_acc', 'l3_three_terms', 'four_terms_odd', 'one_term'])
def test_block_sum_backward_matches_fp64(shape):
    """g_i (+)= a_i mask_i dout, (dA_i, dB_i) = sum over pixels of (mask dout x_i, mask dout): pixel counts that leave the last trip's second pixel
    missing, terms without a gradient output (statistics only), with accumulation, with a ReLU on the block output."""
    import ctypes as C
    import torch
    from addk import _lib as L
    P, Cc, nterm, relu_out, want_g, acc = shape
    lib = L.load()
    dev = torch.device('cuda:0')
    torch.manual_seed(P + Cc + nterm)
    dout = torch.randn(P, Cc, device=dev)
    xs = [torch.randn(P, Cc, device=dev) for _ in range(nterm)]
    a_ = [torch.rand(Cc, device=dev) + 0.5 for _ in range(nterm)]
    b_ = [0.3 * torch.randn(Cc, device=dev) for _ in range(nterm)]
    relu_t = [(i % 2) == 0 for i in range(nterm)]
    out = sum(torch.relu(a_[i] * xs[i] + b_[i]) if relu_t[i] else a_[i] * xs[i] + b_[i] for i in range(nterm))
    if relu_out:
        out = torch.relu(out)
    g0 = [torch.randn(P, Cc, device=dev) for _ in range(nterm)]
    rows = lib.addk_ew_rows(P, Cc)
    ba = L.AffineSumBwdArgs()
    gs, dabs = [], []
    for i in range(nterm):
        ba.term[i].x, ba.term[i].a, ba.term[i].b, ba.term[i].ld, ba.term[i].C, ba.term[i].relu = xs[i].data_ptr(), a_[i].data_ptr(), b_[i].data_ptr(), Cc, Cc, int(relu_t[i])
        g = g0[i].clone() if acc[i] else torch.full((P, Cc), float('nan'), device=dev)
        dab = torch.full((rows, Cc, 2), float('nan'), device=dev, dtype=torch.float64)
        gs.append(g); dabs.append(dab)
        if want_g[i]:
            ba.g[i], ba.ldg[i], ba.accumulate[i] = g.data_ptr(), Cc, int(acc[i])
        ba.dab[i] = dab.data_ptr()
    ba.nterm, ba.P, ba.C, ba.dout, ba.lddo = nterm, P, Cc, dout.data_ptr(), Cc
    ba.out, ba.ldo, ba.relu_out = out.data_ptr(), Cc, int(relu_out)
    L.check(lib.addk_affine_sum_bwd(C.byref(ba), torch.cuda.current_stream().cuda_stream), 'affine_sum_bwd')
    torch.cuda.synchronize()
    d = dout.double() * ((out > 0).double() if relu_out else 1.0)
    for i in range(nterm):
        m = ((a_[i] * xs[i] + b_[i]) > 0).double() if relu_t[i] else torch.ones_like(d)      # the kernel's mask is fmaf(a, x, b) > 0: same sign except within an ulp of 0
        dm = d * m
        ref_dab = torch.stack([(dm * xs[i].double()).sum(0), dm.sum(0)], 1)
        got = dabs[i].sum(0)
        assert torch.isfinite(got).all()
        assert float((got - ref_dab).abs().max() / ref_dab.abs().max()) < 1e-6, i
        if want_g[i]:
            ref_g = dm * a_[i].double() + (g0[i].double() if acc[i] else 0)
            assert torch.isfinite(gs[i]).all()
            assert float((gs[i].double() - ref_g).abs().max() / ref_g.abs().max()) < 1e-6, i
