"""GPU tests of the specialised 3x3 kernels (conv3.hip halo-patch forward / data gradient, wgrad.hip halo-patch weight
gradient) through the C ABI: each launch is compared with an fp64 PyTorch reference of the same operation
(relu(a*x+b) -> conv2d -> batch statistics, and its autograd) and with the generic kernels on identical buffers.
The whole-network parity tests use maps too small to reach these kernels (they need >= 8192 pixels), so they are pinned
here.  Tolerance: 2e-5 of the reference's max-abs for fp32 products with fp32 accumulation over K <= 2736.  Shapes cover the wide heads (3x3, dilation 1-18, 64/128-channel column blocks) and
the cells' dense dilated convolutions (dil_conv_3x3 / dil_conv_5x5, 40/80/160 channels, odd map sizes)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5
FAST_ALL = 31

SHAPES = [
    # name,            N,  H,   W, source channels, Cout, dil, k
    ('two_src_d1',     1, 40, 256, (32, 16), 128, 1, 3),
    ('odd_tail_d2',    2, 33, 129, (24,), 64, 2, 3),
    ('aspp_like_d6',   1, 70, 128, (48,), 256, 6, 3),
    ('wide_blocks_d1', 2, 64, 256, (32,), 256, 1, 3),
    ('max_dil_d18',    1, 64, 128, (16,), 64, 18, 3),
    ('tworow_odd',     2, 65, 300, (24,), 128, 1, 3),      # full-width 3x3 at dilation 1: TWO-ROW tiles (2 x 64 px), odd row count, 300 = 4 x 64 + 44
    ('tworow_c64',     1, 129, 200, (32,), 64, 1, 3),      # ... on the 4-wave form of the <= 64-channel launches (wave pairs split by tile row), dgrad too
    ('cell_dil5_40',   1, 70, 125, (40,), 40, 2, 5),       # dil_conv_5x5 at level 1 (3 column tiles, K tail 40 = 16+16+8)
    ('cell_dil5_80',   2, 63, 127, (80,), 80, 2, 5),       # level 2 (5 column tiles)
    ('cell_dil3_40',   1, 70, 125, (40,), 40, 2, 3),       # dil_conv_3x3
    ('cell_dil3_160',  2, 40, 104, (32,), 160, 2, 3),      # two 5-tile column blocks
    # [r5] the 16-wide-tile kernel of the <= 48-channel 5x5 (conv3n.hip): two sources whose chunks are 16 + 8 | 16 (pair and quad K-steps), 48 outputs (no padding
    # row), an odd row count (the second row of the last pair lies beyond the map), a 200-pixel row (second column tile 72 wide)
    ('n16_two_src',    2, 33, 200, (24, 16), 48, 2, 5),
    ('n16_k8_d1',      2, 40, 130, (8,), 40, 1, 5),        # a single 8-channel chunk: quad steps only, dilation 1
    ('n16_c36',        1, 90, 100, (36,), 36, 2, 5),       # 36 outputs: the third 16-row tile is a quarter full
    ('l3_dil5_160',    2, 32,  64, (48,), 160, 2, 5),      # level-3 map: 3-wave column blocks on QUARTER-width (32-pixel) tiles, one accumulator tile per wave
    ('l3_dil3_160',    2, 32,  64, (32,), 160, 2, 3),
    # the wide pointwise heads on the split kernel as a plain GEMM (KS = 1, conv3.hip c3_geometry_ok): ASPP 1x1 400 -> 256 and the
    # 1280 -> 256 concat conv, whose image-pool branch enters as a per-image bias (aspp_train.py:44-58)
    ('pw_aspp_400',    2, 64, 128, (400,), 256, 1, 1),
    ('pw_cat_1024',    1, 64, 128, (256, 256, 256, 256), 256, 1, 1),
    ('pw_odd_208',     2, 33,  65, (208,), 192, 1, 1),     # odd map, 192 output channels (the narrowest shape that takes this path)
]
BIAS_N = {'pw_cat_1024', 'pw_odd_208'}
SPLIT_ONLY = {'l3_dil5_160', 'l3_dil3_160'}       # maps below the fp32 halo kernel's 8192-pixel floor: the split kernel's launch shapes only


@pytest.fixture(scope='module')
def lib():
    assert torch.cuda.is_available()
    import addk
    from addk import _lib as L
    prec = L.load().addk_get_conv_precision()
    yield L
    L.load().addk_set_fast_paths(FAST_ALL)
    L.load().addk_set_conv_precision(prec)
    L.load().addk_set_split_min_channels(-1)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


PREC = {'fp32': 0, 'f16x3': 1, 'bf16x6': 2}


def _run(L, fast, shape, data, prec='fp32'):
    """One forward + data gradient + weight gradient through the C ABI with the given fast-path mask and arithmetic of the
    halo-patch convolutions (exact fp32 MFMA / split-bf16 with 6 or 3 product terms)."""
    lib = L.load()
    lib.addk_set_fast_paths(fast)
    L.check(lib.addk_set_conv_precision(PREC[prec]), 'set_conv_precision')
    lib.addk_set_split_min_channels(0)          # reach the narrow (<= 64 channel) variants of the split kernel too
    name, N, H, W, Cs, Cout, dil, ks = shape
    taps, pad = ks * ks, dil * (ks // 2)
    dev = data['w'].device
    st = torch.cuda.current_stream().cuda_stream
    ctot = sum(Cs)
    P = N * H * W
    keep = []
    ar = L.ConvArgs()
    for i, Ci in enumerate(Cs):
        ar.src[i].x, ar.src[i].a, ar.src[i].b = data['x'][i].data_ptr(), data['a'][i].data_ptr(), data['b'][i].data_ptr()
        ar.src[i].ld, ar.src[i].C, ar.src[i].relu = Ci, Ci, 1
    ar.nsrc = len(Cs)
    ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, H, W
    ar.KH = ar.KW = ks
    ar.stride, ar.pad, ar.dil, ar.Cout = 1, pad, dil, Cout
    ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = taps * ctot, ctot, 0, Cout
    y = torch.empty(P, Cout, device=dev)
    rows = lib.addk_conv_rows(P, Cout)
    slab = torch.zeros(rows, Cout, 2, device=dev, dtype=torch.float64)
    ar.w, ar.y, ar.stats, ar.stats_ld = data['w'].data_ptr(), y.data_ptr(), slab.data_ptr(), Cout
    if 'bias_n' in data:
        ar.bias_n = data['bias_n'].data_ptr()
    npk = int(lib.addk_conv_fwd_pack_floats(C.byref(ar)))
    assert (npk > 0) == (bool(fast & 2) and not (ks == 1 and prec == 'fp32')), 'forward: halo-patch kernel coverage (%d floats, mask %d)' % (npk, fast)
    if npk:
        wp = torch.empty(npk, device=dev); keep.append(wp)
        ar.wpack, ar.wpack_floats = wp.data_ptr(), npk
    L.check(lib.addk_conv_fwd(C.byref(ar), st), 'conv_fwd')
    out = {'y': y, 'stats': slab.sum(0)}
    choff = 0
    out['g'], out['dab'], out['dw'] = [], [], torch.zeros_like(data['w'])
    for i, Ci in enumerate(Cs):
        da = L.ConvDgradArgs()
        da.dy, da.lddy, da.Cout = data['dy'].data_ptr(), Cout, Cout
        da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, H, W, ks, ks, 1, pad, dil
        da.w, da.ldw, da.cin_total, da.w_choff = data['w'].data_ptr(), taps * ctot, ctot, choff
        da.dst = ar.src[i]
        g = torch.empty(P, Ci, device=dev)
        r2 = lib.addk_conv_rows(P, Ci)
        dab = torch.zeros(r2, Ci, 2, device=dev, dtype=torch.float64)
        da.g, da.ldg, da.accumulate, da.dab = g.data_ptr(), Ci, 0, dab.data_ptr()
        npk = int(lib.addk_conv_dgrad_pack_floats(C.byref(da)))
        if npk:
            wp = torch.empty(npk, device=dev); keep.append(wp)
            da.wpack, da.wpack_floats = wp.data_ptr(), npk
        L.check(lib.addk_conv_dgrad(C.byref(da), st), 'conv_dgrad')
        out['g'].append(g); out['dab'].append(dab.sum(0))
        wa = L.ConvWgradArgs()
        wa.dy, wa.lddy, wa.Cout = data['dy'].data_ptr(), Cout, Cout
        wa.N, wa.H, wa.W, wa.OH, wa.OW, wa.KH, wa.KW, wa.stride, wa.pad, wa.dil = N, H, W, H, W, ks, ks, 1, pad, dil
        wa.src = ar.src[i]
        wa.dw, wa.ldw, wa.cin_total, wa.w_choff, wa.accumulate = out['dw'].data_ptr(), taps * ctot, ctot, choff, 0
        wa.ws_floats = lib.addk_conv_wgrad_ws(P, Cout, Ci, taps)
        ws = torch.empty(int(wa.ws_floats), device=dev); keep.append(ws)
        wa.ws = ws.data_ptr()
        cfg = (C.c_int32 * 4)()
        L.check(lib.addk_conv_wgrad_config(C.byref(wa), cfg), 'wgrad_config')
        if Ci >= 16 and ks == 3 and Cout % 64 == 0:
            assert (cfg[0] == 5) == bool(fast & 4), 'weight gradient: kernel kind %d with mask %d' % (cfg[0], fast)
        elif Ci >= 16 and Cout <= 160 and dil <= 2:
            assert (cfg[0] == 7) == bool(fast & 16), 'cell dilated conv: kernel kind %d with mask %d' % (cfg[0], fast)
        L.check(lib.addk_conv_wgrad(C.byref(wa), st), 'conv_wgrad')
        choff += Ci
    torch.cuda.synchronize()
    return out


def _reference(shape, data):
    name, N, H, W, Cs, Cout, dil, ks = shape
    xs = [x.double().view(N, H, W, -1).permute(0, 3, 1, 2).contiguous().requires_grad_(True) for x in data['x']]
    as_ = [a.double().requires_grad_(True) for a in data['a']]
    bs = [b.double().requires_grad_(True) for b in data['b']]
    z = torch.cat([F.relu(a.view(1, -1, 1, 1) * x + b.view(1, -1, 1, 1)) for x, a, b in zip(xs, as_, bs)], 1)
    w = data['w'].double().view(Cout, ks, ks, -1).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(z, w, padding=dil * (ks // 2), dilation=dil)
    if 'bias_n' in data:
        y = y + data['bias_n'].double().view(N, Cout, 1, 1)
    dy = data['dy'].double().view(N, H, W, Cout).permute(0, 3, 1, 2)
    y.backward(dy)
    yl = y.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
    return {'y': yl, 'stats': torch.stack([yl.sum(0), (yl * yl).sum(0)], 1),
            'g': [x.grad.permute(0, 2, 3, 1).reshape(-1, x.shape[1]) for x in xs],
            'dab': [torch.stack([a.grad, b.grad], 1) for a, b in zip(as_, bs)],
            'dw': w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)}


@pytest.mark.parametrize('prec', ['fp32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('shape', SHAPES, ids=[s[0] for s in SHAPES])
def test_halo_patch_kernels_match_fp64_reference_and_generic(lib, shape, prec):
    """bf16x6 (x = h + m + l in bf16, six product terms on the bf16 matrix pipe) and f16x3 (x = h + l in fp16 under a power-of-two scale,
    three product terms on the fp16 matrix pipe: the default) are held to the SAME 2e-5 bound as the exact fp32 MFMA kernel."""
    name, N, H, W, Cs, Cout, dil, ks = shape
    if name in SPLIT_ONLY and prec == 'fp32':
        pytest.skip('split-kernel launch shape')
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name)))
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    P = N * H * W
    data = {'x': [rnd(P, c) for c in Cs], 'a': [rnd(c) for c in Cs], 'b': [0.3 * rnd(c) for c in Cs],
            'w': 0.1 * rnd(Cout, ks * ks * sum(Cs)), 'dy': rnd(P, Cout)}
    if name in BIAS_N:
        data['bias_n'] = rnd(N, Cout)
    ref = _reference(shape, data)
    fast = _run(lib, FAST_ALL, shape, data, prec)
    slow = _run(lib, 0, shape, data)
    tol = TOL
    bad, log = [], []
    for tag, got in (('fast', fast), ('generic', slow)):
        errs = {'y': _rel(got['y'], ref['y']), 'stats': _rel(got['stats'], ref['stats']), 'dw': _rel(got['dw'], ref['dw'])}
        for i in range(len(Cs)):
            errs['g%d' % i] = _rel(got['g'][i], ref['g'][i])
            errs['dab%d' % i] = _rel(got['dab'][i], ref['dab'][i])
        bad += ['%s/%s %.2e' % (tag, k, v) for k, v in errs.items() if not v <= (tol if tag == 'fast' else TOL)]
        log.append('%s[%s] %s: %s' % (name, prec, tag, ' '.join('%s %.1e' % kv for kv in errs.items())))
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/halo_kernel_errors.txt', 'a') as f:
        f.write('\n'.join(log) + '\n')
    assert not bad, 'beyond %.0e of the fp64 reference: %s' % (tol, ', '.join(bad))
    assert _rel(fast['y'], slow['y']) <= 1e-5 and _rel(fast['dw'], slow['dw']) <= 1e-5



RANGE_CASES = [
    # name,            x scale per source,   weight scale, dy scale,  per-pixel octaves
    ('tiny_dy',        (1.0, 1.0),           0.1,          1e-9,      0),        # gradients of a 4M-pixel mean loss: far below fp16's smallest normal
    ('huge_act',       (3e5, 3e5),           1e-3,         1.0,       0),        # activations beyond fp16's largest finite number
    ('rising_src',     (1e-6, 1e4),          0.1,          1e-3,      0),        # the second source is 10 orders above the first: the running scale drops mid-tile
    ('falling_src',    (1e4, 1e-6),          0.1,          1e-3,      0),
    ('zero_first',     (0.0, 1.0),           30.0,         1e-6,      0),        # an all-zero first chunk (scale 2^126), then ordinary values
    ('octaves',        (1.0, 1.0),           0.1,          1e-4,      12),       # magnitudes spread over 24 octaves inside every tensor
    ('all_zero',       (0.0, 0.0),           0.0,          0.0,       0),        # zero operands: scales 2^126, results exactly zero (no NaN from the inverse scales)
]


@pytest.mark.parametrize('case', RANGE_CASES, ids=[c[0] for c in RANGE_CASES])
@pytest.mark.parametrize('shape', [('two_src_d1', 1, 40, 256, (32, 16), 128, 1, 3), ('n16_two_src', 2, 33, 200, (24, 16), 48, 2, 5), ('two_src_c64', 1, 129, 200, (32, 32), 64, 1, 3)],
                         ids=['c3b_128', 'c3n_48', 'c3b_64'])
def test_split_fp16_holds_fp32_accuracy_over_the_whole_fp32_range(lib, shape, case):
    """f16x3 keeps every operand inside fp16's range with exact power-of-two scales (weights: per tensor at pack time; activations and
    gradients: a running scale per tile / per workgroup): forward, data gradient and weight gradient stay within the fp32 kernels' 2e-5
    of fp64 for gradients of 1e-9, activations of 3e5, sources ten orders of magnitude apart (the accumulators are rescaled mid-tile),
    an all-zero first chunk and 24-octave spreads."""
    cname, xs, ws, dys, octv = case
    name, N, H, W, Cs, Cout, dil, ks = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name + cname)))
    rnd = lambda *s: torch.randn(*s, generator=gen)
    spread = lambda t: t * torch.exp2(torch.randint(-octv, octv + 1, t.shape, generator=gen).float()) if octv else t
    P = N * H * W
    data = {'x': [(spread(rnd(P, c)) * sc).to(dev) for c, sc in zip(Cs, xs)], 'a': [(0.5 + rnd(c).abs()).to(dev) for c in Cs],
            'b': [(0.1 * sc * rnd(c)).to(dev) for c, sc in zip(Cs, xs)],
            'w': (spread(rnd(Cout, ks * ks * sum(Cs))) * ws).to(dev), 'dy': (spread(rnd(P, Cout)) * dys).to(dev)}
    ref = _reference(shape, data)
    got = _run(lib, FAST_ALL, shape, data, 'f16x3')
    errs = {'y': _rel(got['y'], ref['y']), 'dw': _rel(got['dw'], ref['dw'])}
    for i in range(len(Cs)):
        if xs[i] != 0.0:
            errs['g%d' % i] = _rel(got['g'][i], ref['g'][i])
    assert all(torch.isfinite(v).all() for v in [got['y'], got['dw']] + got['g']), 'non-finite output'
    if cname == 'all_zero':
        assert not got['y'].any() and not got['dw'].any() and not any(g.any() for g in got['g'])
    bad = ['%s %.2e' % kv for kv in errs.items() if not kv[1] <= TOL]
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/halo_kernel_errors.txt', 'a') as f:
        f.write('range %s %s: %s\n' % (name, cname, ' '.join('%s %.1e' % kv for kv in errs.items())))
    assert not bad, '%s / %s beyond %.0e of fp64: %s' % (name, cname, TOL, ', '.join(bad))

DW_SHAPES = [
    # name,          N,  H,   W,   C, k, stride, dil
    ('sep5_l2',      2, 63, 127,  80, 5, 1, 1),
    ('sep3_l1',      1, 125, 253, 40, 3, 1, 1),
    ('sep5_stride2', 2, 64, 128,  80, 5, 2, 1),
    ('sep3_dil2',    1, 70,  90,  24, 3, 1, 2),
    ('sep5_l3',      2, 32,  64, 160, 5, 1, 1),
    ('sep3_odd_c',   1, 50,  70,  36, 3, 2, 1),
]


def _dw_run(L, fast, shape, data, grads=False):
    lib = L.load()
    lib.addk_set_fast_paths(fast)
    name, N, H, W, Cc, k, s, d = shape
    pad = d * (k // 2)
    OH, OW = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    dev = data['x'].device
    st = torch.cuda.current_stream().cuda_stream
    ar = L.DwArgs()
    ar.src.x, ar.src.a, ar.src.b = data['x'].data_ptr(), data['a'].data_ptr(), data['b'].data_ptr()
    ar.src.ld, ar.src.C, ar.src.relu = Cc, Cc, 1
    ar.N, ar.H, ar.W, ar.OH, ar.OW, ar.KH, ar.KW, ar.stride, ar.pad, ar.dil = N, H, W, OH, OW, k, k, s, pad, d
    y = torch.empty(N * OH * OW, Cc, device=dev)
    ar.w, ar.y, ar.ldy = data['w'].data_ptr(), y.data_ptr(), Cc
    L.check(lib.addk_dw_fwd(C.byref(ar), st), 'dw_fwd')
    out = {'y': y}
    if grads:
        ba = L.DwBwdArgs()
        ba.dy, ba.lddy = data['dy'].data_ptr(), Cc
        ba.N, ba.H, ba.W, ba.OH, ba.OW, ba.KH, ba.KW, ba.stride, ba.pad, ba.dil = N, H, W, OH, OW, k, k, s, pad, d
        ba.src = ar.src
        ba.w = data['w'].data_ptr()
        rows = lib.addk_dw_rows(N * H * W, Cc)
        g = torch.empty(N * H * W, Cc, device=dev)
        dab = torch.zeros(rows, Cc, 2, device=dev, dtype=torch.float64)
        dw = torch.zeros(Cc, k * k, device=dev)
        ws = torch.empty(rows * Cc * k * k, device=dev)
        ba.g, ba.ldg, ba.accumulate, ba.dab, ba.dw, ba.dw_accumulate, ba.ws = g.data_ptr(), Cc, 0, dab.data_ptr(), dw.data_ptr(), 0, ws.data_ptr()
        L.check(lib.addk_dw_bwd(C.byref(ba), st), 'dw_bwd')
        out.update(g=g, dab=dab.sum(0), dw=dw)
    torch.cuda.synchronize()
    return out, (OH, OW)


@pytest.mark.parametrize('shape', DW_SHAPES, ids=[s[0] for s in DW_SHAPES])
def test_depthwise_tiled_kernels_match_fp64_reference_and_generic(lib, shape):
    name, N, H, W, Cc, k, s, d = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name)))
    rnd = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
    pad = d * (k // 2)
    OH, OW = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    data = {'x': rnd(N * H * W, Cc), 'a': rnd(Cc), 'b': 0.3 * rnd(Cc), 'w': 0.3 * rnd(Cc, k * k), 'dy': rnd(N * OH * OW, Cc)}
    x = data['x'].double().view(N, H, W, Cc).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    a, b = data['a'].double().requires_grad_(True), data['b'].double().requires_grad_(True)
    w = data['w'].double().view(Cc, 1, k, k).clone().requires_grad_(True)
    yr = F.conv2d(F.relu(a.view(1, -1, 1, 1) * x + b.view(1, -1, 1, 1)), w, stride=s, padding=pad, dilation=d, groups=Cc)
    yr.backward(data['dy'].double().view(N, OH, OW, Cc).permute(0, 3, 1, 2))
    ref = {'y': yr.detach().permute(0, 2, 3, 1).reshape(-1, Cc), 'g': x.grad.permute(0, 2, 3, 1).reshape(-1, Cc),
           'dab': torch.stack([a.grad, b.grad], 1), 'dw': w.grad.view(Cc, k * k)}
    fast, _ = _dw_run(lib, FAST_ALL, shape, data, grads=True)
    slow, _ = _dw_run(lib, 0, shape, data, grads=True)
    bad = []
    for tag, got in (('fast', fast), ('generic', slow)):
        bad += ['%s/%s %.2e' % (tag, kk, _rel(got[kk], ref[kk])) for kk in ('y', 'g', 'dab', 'dw') if not _rel(got[kk], ref[kk]) <= TOL]
    assert not bad, 'beyond %.0e of the fp64 reference: %s' % (TOL, ', '.join(bad))


@pytest.mark.parametrize('N,H,W,OH,OW,Cc', [(2, 64, 128, 256, 512, 19), (1, 33, 65, 129, 257, 19), (1, 40, 50, 80, 100, 6)])
def test_logits_upsample_backward_tiled_matches_autograd(lib, N, H, W, OH, OW, Cc):
    """NCHW gradient of the up-sampled logits -> NHWC gradient of the low-resolution logits (decoder.py:28)."""
    L = lib
    l = L.load()
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(N * 1000 + H)
    dy = torch.randn(N, Cc, OH, OW, generator=gen).to(dev)
    a = torch.randn(Cc, generator=gen).to(dev)
    x = torch.zeros(N, Cc, H, W, device=dev, dtype=torch.float64, requires_grad=True)
    y = F.interpolate(x * a.double().view(1, -1, 1, 1), size=(OH, OW), mode='bilinear', align_corners=False)
    y.backward(dy.double() * 0.5)
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, Cc)
    scale = torch.full((1,), 0.5, device=dev)
    bz = torch.zeros(Cc, device=dev)
    outs = []
    for fast in (FAST_ALL, 0):
        l.addk_set_fast_paths(fast)
        ld = (Cc + 3) // 4 * 4
        g = torch.zeros(N * H * W, ld, device=dev)
        ar = L.ResizeBwdArgs()
        ar.dy, ar.lddy, ar.nchw_in, ar.dy_scale = dy.data_ptr(), 0, 1, scale.data_ptr()
        ar.src.x, ar.src.a, ar.src.b, ar.src.ld, ar.src.C, ar.src.relu = g.data_ptr(), a.data_ptr(), bz.data_ptr(), ld, Cc, 0
        ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, OH, OW
        ar.g, ar.ldg, ar.accumulate = g.data_ptr(), ld, 0
        L.check(l.addk_resize_bwd(C.byref(ar), torch.cuda.current_stream().cuda_stream), 'resize_bwd')
        torch.cuda.synchronize()
        outs.append(g[:, :Cc].clone())
        assert _rel(outs[-1], ref) <= TOL, 'mask %d: %.2e' % (fast, _rel(outs[-1], ref))
    assert torch.equal(outs[0], outs[1]) or _rel(outs[0], outs[1]) <= 1e-6


@pytest.mark.parametrize('name,N,H,W,Ci,Cout,k,s,d', [
    ('pw40', 2, 63, 127, 40, 40, 1, 1, 1), ('pw80', 1, 64, 128, 80, 80, 1, 1, 1), ('glue200', 1, 50, 90, 200, 40, 1, 1, 1),
    ('reduce_s2', 2, 128, 96, 80, 40, 1, 2, 1), ('dense3_s2', 2, 97, 129, 48, 96, 3, 2, 1), ('pw160', 2, 32, 64, 160, 160, 1, 1, 1),
    ('stem0_like', 2, 256, 511, 3, 64, 3, 2, 1)])           # 3 input channels (ADD.py:153-157): all 27 (tap, channel) columns in one workgroup (wgrad_st_kernel), odd width
@pytest.mark.parametrize('prec,dys', [('bf16x6', 1.0), ('f16x3', 1.0), ('f16x3', 1e-9)], ids=['fp32_mfma', 'f16x3', 'f16x3_tiny_dy'])
def test_register_streaming_wgrad_matches_fp64_reference(lib, name, N, H, W, Ci, Cout, k, s, d, prec, dys):
    """Weight gradients of the narrow cell convolutions (1x1, strided, multi-tile) on wgrad_rs_kernel, and of stem0 on its few-input-channel
    sibling wgrad_st_kernel, vs autograd in fp64: the fp32-MFMA form (every mode but f16x3) and the split-fp16 form with per-wave running scales
    (f16x3; also with gradients of 1e-9, far below fp16's range)."""
    L = lib
    l = L.load()
    L.check(l.addk_set_conv_precision(PREC[prec]), 'set_conv_precision')
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name)))
    rnd = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
    pad = d * (k // 2)
    OH, OW = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    x, a, b, dy = rnd(N * H * W, Ci), rnd(Ci), 0.3 * rnd(Ci), dys * rnd(N * OH * OW, Cout)
    xr = x.double().view(N, H, W, Ci).permute(0, 3, 1, 2)
    z = F.relu(a.double().view(1, -1, 1, 1) * xr + b.double().view(1, -1, 1, 1))
    w = torch.zeros(Cout, Ci, k, k, device=dev, dtype=torch.float64, requires_grad=True)
    F.conv2d(z, w, stride=s, padding=pad, dilation=d).backward(dy.double().view(N, OH, OW, Cout).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)
    outs = {}
    for fast in (31, 0):
        l.addk_set_fast_paths(fast)
        wa = L.ConvWgradArgs()
        wa.dy, wa.lddy, wa.Cout = dy.data_ptr(), Cout, Cout
        wa.N, wa.H, wa.W, wa.OH, wa.OW, wa.KH, wa.KW, wa.stride, wa.pad, wa.dil = N, H, W, OH, OW, k, k, s, pad, d
        wa.src.x, wa.src.a, wa.src.b, wa.src.ld, wa.src.C, wa.src.relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), Ci, Ci, 1
        dw = torch.zeros(Cout, k * k * Ci, device=dev)
        wa.dw, wa.ldw, wa.cin_total, wa.w_choff, wa.accumulate = dw.data_ptr(), k * k * Ci, Ci, 0, 0
        wa.ws_floats = l.addk_conv_wgrad_ws(N * OH * OW, Cout, Ci, k * k)
        ws = torch.empty(int(wa.ws_floats), device=dev)
        wa.ws = ws.data_ptr()
        cfg = (C.c_int32 * 4)()
        L.check(l.addk_conv_wgrad_config(C.byref(wa), cfg), 'wgrad_config')
        assert (cfg[0] == (8 if Ci <= 4 else 6)) == (fast == 31), 'kernel kind %d with mask %d' % (cfg[0], fast)
        L.check(l.addk_conv_wgrad(C.byref(wa), torch.cuda.current_stream().cuda_stream), 'conv_wgrad')
        torch.cuda.synchronize()
        outs[fast] = dw
        assert _rel(dw, ref) <= TOL, 'mask %d: %.2e' % (fast, _rel(dw, ref))


SEP_SHAPES = [
    # name,        N,  H,   W,  C, k
    ('sep3_c40',   2, 61, 125, 40, 3),      # level-1 cells (F=20): 3 channel groups, odd map
    ('sep5_c40',   1, 70, 125, 40, 5),
    ('sep3_c80',   2, 33,  65, 80, 3),      # level-2 cells: 5 channel groups
    ('sep5_c80',   2, 63, 127, 80, 5),
    ('sep5_small', 1,  9,  11, 48, 5),      # map smaller than a tile row, every pixel touches the border
    ('sep5_c40_l1', 2, 128, 256, 40, 5),    # the level-1 map of config 2: two rows per wave (R = 2), 512 workgroups
    ('sep3_c40_l1', 2, 125, 253, 40, 3),    # ... and its odd-sized sibling after an up-path
    ('sep3_c36',   1, 40,  70, 36, 3),      # channel count below the padded pixel stride (zero-filled padding quad)
]


@pytest.mark.parametrize('shape', SEP_SHAPES, ids=[s[0] for s in SEP_SHAPES])
def test_fused_sepconv_half_matches_fp64_reference(lib, shape):
    """addk_sep_fwd (ReLU/lazy-BN prologue -> depthwise k x k -> pointwise 1x1 -> statistics in ONE launch) through the C ABI
    against an fp64 PyTorch evaluation of the same half of SepConv (operations.py:51-54): training form (raw output, fp64
    (sum, sumsq) slab, depthwise output written for the backward pass) and the inference form whose epilogue applies the
    op's frozen BatchNorm and adds the other branches of the cell block (ADD.py:108)."""
    L = lib
    lb = L.load()
    lb.addk_set_fast_paths(FAST_ALL)
    name, N, H, W, Cc, k = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name)))
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    P = N * H * W
    x, a, b = rnd(P, Cc), rnd(Cc), 0.3 * rnd(Cc)
    wdw, wpw = 0.3 * rnd(Cc, k * k), 0.2 * rnd(Cc, Cc)
    st = torch.cuda.current_stream().cuda_stream
    # fp64 reference
    xr = x.double().view(N, H, W, Cc).permute(0, 3, 1, 2)
    z = F.relu(a.double().view(1, -1, 1, 1) * xr + b.double().view(1, -1, 1, 1))
    t_ref = F.conv2d(z, wdw.double().view(Cc, 1, k, k), padding=k // 2, groups=Cc)
    y_ref = F.conv2d(t_ref, wpw.double().view(Cc, Cc, 1, 1))
    flat = lambda v: v.permute(0, 2, 3, 1).reshape(P, Cc)
    ar = L.SepArgs()
    ar.src.x, ar.src.a, ar.src.b, ar.src.ld, ar.src.C, ar.src.relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
    ar.N, ar.H, ar.W, ar.K, ar.Cout, ar.ldw = N, H, W, k, Cc, Cc
    ar.dw_w, ar.pw_w = wdw.data_ptr(), wpw.data_ptr()
    y, t = torch.empty(P, Cc, device=dev), torch.empty(P, Cc, device=dev)
    ar.y, ar.ldy, ar.t, ar.ldt = y.data_ptr(), Cc, t.data_ptr(), Cc
    rows = max(lb.addk_conv_rows(P, Cc), lb.addk_sep_rows(C.byref(ar))) + 3          # three rows nobody owns: must come back zero
    assert lb.addk_sep_rows(C.byref(ar)) > 0
    slab = torch.full((rows, Cc, 2), float('nan'), device=dev, dtype=torch.float64)
    ar.stats, ar.stats_ld, ar.stats_rows = slab.data_ptr(), Cc, rows
    assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
    L.check(lb.addk_sep_fwd(C.byref(ar), st), 'sep_fwd')
    torch.cuda.synchronize()
    yl = flat(y_ref)
    errs = {'t': _rel(t, flat(t_ref)), 'y': _rel(y, yl), 'stats': _rel(slab.sum(0), torch.stack([yl.sum(0), (yl * yl).sum(0)], 1))}
    # training form with the statistics finalized by the launch's LAST workgroup (csrc/bnfin.h): (a, b, mean, invstd, running
    # statistics) against fp64, twice in a row (the ticket counter must come back to zero), bit-identical run to run
    gam, bet = 1 + 0.2 * rnd(Cc), 0.2 * rnd(Cc)
    rm0, rv0 = 0.3 * rnd(Cc), 0.5 + torch.rand(Cc, generator=gen).to(dev)
    nws = (int(lb.addk_bn_fin_ws_bytes(lb.addk_sep_rows(C.byref(ar)), Cc)) + 3) // 4
    counter = torch.zeros(nws, dtype=torch.int32, device=dev)
    outs = []
    for rep in range(2):
        fa, fb, fm, fi = (torch.full((Cc,), float('nan'), device=dev) for _ in range(4))
        rm, rv = rm0.clone(), rv0.clone()
        slab.fill_(float('nan'))
        ar.fin.count, ar.fin.gamma, ar.fin.beta = float(P), gam.data_ptr(), bet.data_ptr()
        ar.fin.running_mean, ar.fin.running_var, ar.fin.momentum, ar.fin.eps = rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5
        ar.fin.a, ar.fin.b, ar.fin.mean, ar.fin.invstd = fa.data_ptr(), fb.data_ptr(), fm.data_ptr(), fi.data_ptr()
        ar.fin_counter = counter.data_ptr()
        assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
        L.check(lb.addk_sep_fwd(C.byref(ar), st), 'sep_fwd (fused finalize)')
        torch.cuda.synchronize()
        ncnt = 1 + (lb.addk_sep_rows(C.byref(ar)) + 15) // 16
        assert int(counter[:ncnt].abs().sum()) == 0, 'ticket counters not reset'
        outs.append([v.clone() for v in (fa, fb, fm, fi, rm, rv)])
    assert all(torch.equal(u, v) for u, v in zip(*outs)), 'fused finalize is not reproducible'
    mean64, var64 = yl.mean(0), yl.var(0, unbiased=False)
    inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
    fa, fb, fm, fi, rm, rv = outs[0]
    errs['fin_a'] = _rel(fa, gam.double() * inv64)
    errs['fin_b'] = _rel(fb, bet.double() - mean64 * gam.double() * inv64)
    errs['fin_mean'], errs['fin_invstd'] = _rel(fm, mean64), _rel(fi, inv64)
    errs['fin_rm'] = _rel(rm, 0.9 * rm0.double() + 0.1 * mean64)
    errs['fin_rv'] = _rel(rv, 0.9 * rv0.double() + 0.1 * var64 * P / (P - 1))
    ar.fin.a, ar.fin_counter = None, None
    # inference form: y = ea*acc + eb + relu(a1*u1 + b1) + u2
    ea, eb = 1 + 0.2 * rnd(Cc), 0.2 * rnd(Cc)
    u1, a1, b1, u2 = rnd(P, Cc), rnd(Cc), 0.3 * rnd(Cc), rnd(P, Cc)
    ar.t, ar.stats = None, None
    ar.ea, ar.eb, ar.nterm = ea.data_ptr(), eb.data_ptr(), 2
    ar.term[0].x, ar.term[0].a, ar.term[0].b, ar.term[0].ld, ar.term[0].C, ar.term[0].relu = u1.data_ptr(), a1.data_ptr(), b1.data_ptr(), Cc, Cc, 1
    ar.term[1].x, ar.term[1].ld, ar.term[1].C, ar.term[1].relu = u2.data_ptr(), Cc, Cc, 0
    y2 = torch.empty(P, Cc, device=dev)
    ar.y = y2.data_ptr()
    assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
    L.check(lb.addk_sep_fwd(C.byref(ar), st), 'sep_fwd (inference epilogue)')
    torch.cuda.synchronize()
    ref2 = ea.double() * yl + eb.double() + F.relu(a1.double() * u1.double() + b1.double()) + u2.double()
    errs['y_sum'] = _rel(y2, ref2)
    # and against the two unfused launches on the same buffers
    da = L.DwArgs()
    da.src = ar.src
    da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, H, W, k, k, 1, k // 2, 1
    t2 = torch.empty(P, Cc, device=dev)
    da.w, da.y, da.ldy = wdw.data_ptr(), t2.data_ptr(), Cc
    L.check(lb.addk_dw_fwd(C.byref(da), st), 'dw_fwd')
    torch.cuda.synchronize()
    errs['t_vs_unfused'] = _rel(t, t2)
    bad = ['%s %.2e' % kv for kv in errs.items() if not kv[1] <= TOL]
    assert not bad, '%s beyond %.0e: %s' % (name, TOL, ', '.join(bad))


@pytest.mark.parametrize('shape', SEP_SHAPES, ids=[s[0] for s in SEP_SHAPES])
def test_fused_sepconv_half_backward_matches_fp64_autograd(lib, shape):
    """addk_sep_bwd (pointwise data gradient on the matrix cores -> LDS -> depthwise backward, ONE launch) through the C ABI
    against fp64 autograd of y = pw(dw(relu(a*x + b))): gradient wrt x (first touch and accumulate), the (dA, dB) sums of the
    input's lazy BatchNorm, the depthwise weight gradient from the per-workgroup workspace rows; bit-identical run to run."""
    L = lib
    lb = L.load()
    lb.addk_set_fast_paths(FAST_ALL)
    name, N, H, W, Cc, k = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(7 + sum(map(ord, name)))
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    P = N * H * W
    x, a, b = rnd(P, Cc), rnd(Cc), 0.3 * rnd(Cc)
    wdw, wpw, dy = 0.3 * rnd(Cc, k * k), 0.2 * rnd(Cc, Cc), rnd(P, Cc)
    g0 = rnd(P, Cc)
    # fp64 autograd
    xr = x.double().view(N, H, W, Cc).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ar_, br_ = a.double().requires_grad_(True), b.double().requires_grad_(True)
    wd = wdw.double().view(Cc, 1, k, k).requires_grad_(True)
    z = F.relu(ar_.view(1, -1, 1, 1) * xr + br_.view(1, -1, 1, 1))
    y = F.conv2d(F.conv2d(z, wd, padding=k // 2, groups=Cc), wpw.double().view(Cc, Cc, 1, 1))
    y.backward(dy.double().view(N, H, W, Cc).permute(0, 3, 1, 2))
    flat = lambda v: v.permute(0, 2, 3, 1).reshape(P, Cc)
    ba = L.SepBwdArgs()
    ba.dy, ba.lddy, ba.N, ba.H, ba.W, ba.K = dy.data_ptr(), Cc, N, H, W, k
    ba.src.x, ba.src.a, ba.src.b, ba.src.ld, ba.src.C, ba.src.relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
    ba.Cout, ba.ldw, ba.dw_w, ba.pw_w = Cc, Cc, wdw.data_ptr(), wpw.data_ptr()
    rows = lb.addk_sep_bwd_rows(C.byref(ba))
    assert rows > 0
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for acc in (0, 1, 0):
        g = g0.clone() if acc else torch.full((P, Cc), float('nan'), device=dev)
        dab = torch.full((rows, Cc, 2), float('nan'), device=dev, dtype=torch.float64)
        ws = torch.full((rows, Cc, k * k), float('nan'), device=dev)
        ba.g, ba.ldg, ba.accumulate, ba.dab, ba.ws = g.data_ptr(), Cc, acc, dab.data_ptr(), ws.data_ptr()
        L.check(lb.addk_sep_bwd(C.byref(ba), st), 'sep_bwd')
        torch.cuda.synchronize()
        outs.append((g, dab.sum(0), ws.double().sum(0)))
    assert all(torch.equal(u, v) for u, v in zip(outs[0], outs[2])), 'not reproducible'
    gx = flat(xr.grad)
    errs = {'dx': _rel(outs[0][0], gx), 'dx_acc': _rel(outs[1][0], gx + g0.double()),
            'dab': _rel(outs[0][1], torch.stack([ar_.grad, br_.grad], 1)), 'dw': _rel(outs[0][2], wd.grad.view(Cc, k * k))}
    bad = ['%s %.2e' % kv for kv in errs.items() if not kv[1] <= TOL]
    assert not bad, '%s beyond %.0e: %s' % (name, TOL, ', '.join(bad))


S2_SHAPES = [
    # name,        N,   H,   W, source channels, Cout
    ('stem2_even', 2,  66, 258, (64,), 128),
    ('stem2_odd',  1, 129, 257, (64,), 128),              # odd input: the last output row / column reads the image edge, no padding row below
    ('s2_two_src', 1,  96, 400, (24, 8), 128),            # K tail (8 of 16 channels in the last chunk), 200 = 128 + 72 output columns
]


@pytest.mark.parametrize('prec', ['bf16x6', 'f16x3'])
@pytest.mark.parametrize('shape', S2_SHAPES, ids=[s[0] for s in S2_SHAPES])
def test_stride2_conv_on_split_kernel_matches_fp64_reference(lib, shape, prec):
    """stem2 (ADD.py:140-144: ReLU, 3x3 stride-2 conv 64 -> 128, BatchNorm) on the split-bf16 kernel: the forward as
    conv3b_kernel<ST = 2> (de-interleaved patch rows), the data gradient as four parity-class launches (1 / 2 / 2 / 4 taps, strided
    scatter).  Output, batch statistics, data gradient and its (dA, dB) sums against fp64 autograd and against the generic kernel."""
    name, N, H, W, Cs, Cout = shape
    lb = lib.load()
    dev = torch.device('cuda:0')
    gen = torch.Generator(device='cpu').manual_seed(sum(map(ord, name)))
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    P, ctot = N * OH * OW, sum(Cs)
    xs, as_, bs = [rnd(N * H * W, c) for c in Cs], [rnd(c) for c in Cs], [0.3 * rnd(c) for c in Cs]
    w = 0.1 * rnd(Cout, 9 * ctot)
    dy = rnd(P, Cout)
    x64 = [x.double().view(N, H, W, -1).permute(0, 3, 1, 2).contiguous().requires_grad_(True) for x in xs]
    a64, b64 = [a.double().requires_grad_(True) for a in as_], [b.double().requires_grad_(True) for b in bs]
    z = torch.cat([F.relu(a.view(1, -1, 1, 1) * x + b.view(1, -1, 1, 1)) for x, a, b in zip(x64, a64, b64)], 1)
    y64 = F.conv2d(z, w.double().view(Cout, 3, 3, ctot).permute(0, 3, 1, 2), stride=2, padding=1)
    y64.backward(dy.double().view(N, OH, OW, Cout).permute(0, 3, 1, 2))
    ref = y64.detach().permute(0, 2, 3, 1).reshape(P, Cout)
    gref = [x.grad.permute(0, 2, 3, 1).reshape(N * H * W, -1) for x in x64]
    dabref = [torch.stack([a.grad, b.grad], 1) for a, b in zip(a64, b64)]
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for tag, fast in (('split', FAST_ALL), ('generic', 0)):
        lb.addk_set_fast_paths(fast)
        lib.check(lb.addk_set_conv_precision(PREC[prec]), 'set_conv_precision')
        lb.addk_set_split_min_channels(0)
        ar = lib.ConvArgs()
        for i, Ci in enumerate(Cs):
            ar.src[i].x, ar.src[i].a, ar.src[i].b = xs[i].data_ptr(), as_[i].data_ptr(), bs[i].data_ptr()
            ar.src[i].ld, ar.src[i].C, ar.src[i].relu = Ci, Ci, 1
        ar.nsrc = len(Cs)
        ar.N, ar.H, ar.W, ar.OH, ar.OW = N, H, W, OH, OW
        ar.KH = ar.KW = 3
        ar.stride, ar.pad, ar.dil, ar.Cout = 2, 1, 1, Cout
        ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = 9 * ctot, ctot, 0, Cout
        y = torch.full((P, Cout), float('nan'), device=dev)
        rows = lb.addk_conv_rows(P, Cout)
        slab = torch.zeros(rows, Cout, 2, device=dev, dtype=torch.float64)
        ar.w, ar.y, ar.stats, ar.stats_ld = w.data_ptr(), y.data_ptr(), slab.data_ptr(), Cout
        npk = int(lb.addk_conv_fwd_pack_floats(C.byref(ar)))
        assert (npk > 0) == (tag == 'split'), 'stride-2 forward: split-kernel coverage (%d floats)' % npk
        keep = [torch.empty(max(npk, 1), device=dev)]
        if npk:
            ar.wpack, ar.wpack_floats = keep[0].data_ptr(), npk
        lib.check(lb.addk_conv_fwd(C.byref(ar), st), 'conv_fwd')
        gs, dabs = [], []
        choff = 0
        for i, Ci in enumerate(Cs):
            da = lib.ConvDgradArgs()
            da.dy, da.lddy, da.Cout = dy.data_ptr(), Cout, Cout
            da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, OH, OW, 3, 3, 2, 1, 1
            da.w, da.ldw, da.cin_total, da.w_choff = w.data_ptr(), 9 * ctot, ctot, choff
            da.dst = ar.src[i]
            g = torch.full((N * H * W, Ci), float('nan'), device=dev)
            r2 = lb.addk_conv_rows(N * H * W, Ci)
            dab = torch.zeros(r2, Ci, 2, device=dev, dtype=torch.float64)
            da.g, da.ldg, da.accumulate, da.dab = g.data_ptr(), Ci, 0, dab.data_ptr()
            npk = int(lb.addk_conv_dgrad_pack_floats(C.byref(da)))
            assert (npk > 0) == (tag == 'split' and 32 <= Ci <= 64), 'stride-2 data gradient: split-kernel coverage (%d floats, %d channels)' % (npk, Ci)
            if npk:
                keep.append(torch.empty(npk, device=dev))
                da.wpack, da.wpack_floats = keep[-1].data_ptr(), npk
            lib.check(lb.addk_conv_dgrad(C.byref(da), st), 'conv_dgrad')
            gs.append(g); dabs.append(dab)
            choff += Ci
        torch.cuda.synchronize()
        outs[tag] = (y, slab.sum(0), gs, [d.sum(0) for d in dabs])
    tol = TOL
    rs = torch.stack([ref.sum(0), (ref * ref).sum(0)], 1)
    for tag, (y, s, gs, dabs) in outs.items():
        errs = {'y': _rel(y, ref), 'stats': _rel(s, rs)}
        for i in range(len(Cs)):
            errs['g%d' % i] = _rel(gs[i], gref[i])
            errs['dab%d' % i] = _rel(dabs[i], dabref[i])
        bad = ['%s %.2e' % kv for kv in errs.items() if not kv[1] <= (tol if tag == 'split' else TOL)]
        assert not bad, '%s[%s] %s: %s' % (name, prec, tag, ', '.join(bad))
