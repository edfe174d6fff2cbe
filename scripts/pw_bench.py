"""Stand-alone times of the pointwise (1x1) launches through the C ABI: forward (with statistics) and data gradient, a chain of dependent launches.
   python scripts/pw_bench.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import addk
from addk import _lib as L

lib = L.load()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def bench(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, H, W, K, Cout in [('l1 40->40', 2, 125, 253, 40, 40), ('l1 80->40', 2, 125, 253, 80, 40), ('l1 40->40 even', 2, 128, 256, 40, 40),
                               ('l2 80->80', 2, 63, 127, 80, 80), ('l3 160->160', 2, 32, 64, 160, 160)]:
    P = N * H * W
    x, a, b = torch.randn(P, K, device=dev), torch.randn(K, device=dev), torch.randn(K, device=dev)
    w = 0.1 * torch.randn(Cout, K, device=dev)
    y = torch.empty(P, Cout, device=dev)
    ar = L.ConvArgs()
    ar.src[0].x, ar.src[0].a, ar.src[0].b, ar.src[0].ld, ar.src[0].C, ar.src[0].relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), K, K, 1
    ar.nsrc = 1
    ar.N, ar.H, ar.W, ar.OH, ar.OW, ar.KH, ar.KW, ar.stride, ar.pad, ar.dil, ar.Cout = N, H, W, H, W, 1, 1, 1, 0, 1, Cout
    ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = K, K, 0, Cout
    rows = lib.addk_conv_rows(P, Cout)
    slab = torch.zeros(rows, Cout, 2, device=dev, dtype=torch.float64)
    ar.w, ar.y, ar.stats, ar.stats_ld = w.data_ptr(), y.data_ptr(), slab.data_ptr(), Cout
    tf = bench(lambda: L.check(lib.addk_conv_fwd(C.byref(ar), st), 'fwd'))
    ar.stats = 0
    tf0 = bench(lambda: L.check(lib.addk_conv_fwd(C.byref(ar), st), 'fwd'))
    dy = torch.randn(P, Cout, device=dev)
    g = torch.empty(P, K, device=dev)
    da = L.ConvDgradArgs()
    da.dy, da.lddy, da.Cout = dy.data_ptr(), Cout, Cout
    da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, H, W, 1, 1, 1, 0, 1
    da.w, da.ldw, da.cin_total, da.w_choff = w.data_ptr(), K, K, 0
    da.dst = ar.src[0]
    r2 = lib.addk_conv_rows(P, K)
    dab = torch.zeros(r2, K, 2, device=dev, dtype=torch.float64)
    da.g, da.ldg, da.accumulate, da.dab = g.data_ptr(), K, 0, dab.data_ptr()
    td = bench(lambda: L.check(lib.addk_conv_dgrad(C.byref(da), st), 'dgrad'))
    da.accumulate = 1
    tda = bench(lambda: L.check(lib.addk_conv_dgrad(C.byref(da), st), 'dgrad'))
    mbf, mbd = 4e-6 * P * (K + Cout), 4e-6 * P * (Cout + 2 * K)
    print('%-16s P=%6d  fwd+stats %5.1f us (%4.2f TB/s)  fwd %5.1f us   dgrad %5.1f us (%4.2f TB/s)  dgrad+accumulate %5.1f us' % (name, P, tf, mbf / tf, tf0, td, mbd / td, tda))
