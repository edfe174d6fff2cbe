#!/usr/bin/env python
"""Micro-benchmark of one SepConv half (operations.py:51-54) through the C ABI: the fused launch of csrc/sepf.hip against the
launches it replaces (depthwise + pointwise [+ bn_finalize | + affine_sum]), at the cell shapes of config 2.  Each variant is
captured into a hipGraph of REP dependent repetitions (the output of one is the input of the next, as in the network) and
timed with HIP events.      python scripts/bench_sep.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import addk                                    # noqa: E402
import addk._lib as L                          # noqa: E402

REP = 20


def main():
    lb = L.load()
    dev = torch.device('cuda:0')
    shapes = [(2, 128, 256, 40, 3), (2, 128, 256, 40, 5), (2, 125, 253, 40, 5), (2, 64, 128, 80, 3), (2, 64, 128, 80, 5), (2, 63, 127, 80, 5),
              (1, 128, 256, 40, 5), (1, 64, 128, 80, 5)]
    torch.manual_seed(0)
    for N, H, W, Cc, k in shapes:
        P = N * H * W
        bufs = [torch.randn(P, Cc, device=dev) for _ in range(2)]
        tb = torch.empty(P, Cc, device=dev)
        a, b = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        wdw, wpw = 0.3 * torch.randn(Cc, k * k, device=dev), 0.2 * torch.randn(Cc, Cc, device=dev)
        gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        fa, fb, fm, fi = (torch.zeros(Cc, device=dev) for _ in range(4))
        counter = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
        u1 = torch.randn(P, Cc, device=dev)
        keep = []

        def sep_args(src, dst, mode):
            ar = L.SepArgs()
            ar.src.x, ar.src.a, ar.src.b, ar.src.ld, ar.src.C, ar.src.relu = src.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
            ar.N, ar.H, ar.W, ar.K, ar.Cout, ar.ldw = N, H, W, k, Cc, Cc
            ar.dw_w, ar.pw_w, ar.y, ar.ldy = wdw.data_ptr(), wpw.data_ptr(), dst.data_ptr(), Cc
            if mode in ('train', 'train_sepfin'):
                rows = max(lb.addk_conv_rows(P, Cc), lb.addk_sep_rows(C.byref(ar)))
                slab = torch.zeros(rows, Cc, 2, dtype=torch.float64, device=dev)
                ar.t, ar.ldt, ar.stats, ar.stats_ld, ar.stats_rows = tb.data_ptr(), Cc, slab.data_ptr(), Cc, rows
                ar.fin.count, ar.fin.gamma, ar.fin.beta = float(P), gam.data_ptr(), bet.data_ptr()
                ar.fin.running_mean, ar.fin.running_var, ar.fin.momentum, ar.fin.eps = rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5
                ar.fin.a, ar.fin.b, ar.fin.mean, ar.fin.invstd = fa.data_ptr(), fb.data_ptr(), fm.data_ptr(), fi.data_ptr()
                ar.fin_counter = counter.data_ptr()
                keep.append(slab)
                if mode == 'train_sepfin':            # the fused launch + a separate bn_finalize launch
                    fin = L.BnFinalizeArgs()
                    for f_ in ('count', 'gamma', 'beta', 'running_mean', 'running_var', 'momentum', 'eps', 'a', 'b', 'mean', 'invstd'):
                        setattr(fin, f_, getattr(ar.fin, f_))
                    fin.partial, fin.rows, fin.C = slab.data_ptr(), rows, Cc
                    ar.fin.a, ar.fin_counter = None, None
                    keep.append(fin)
                    assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
                    keep.append(ar)
                    return [(lb.addk_sep_fwd, (C.byref(ar),)), (lb.addk_bn_finalize, (C.byref(fin),))]
            else:
                ar.ea, ar.eb, ar.nterm = a.data_ptr(), b.data_ptr(), 1
                ar.term[0].x, ar.term[0].ld, ar.term[0].C = u1.data_ptr(), Cc, Cc
            assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
            keep.append(ar)
            return ar

        def old(src, dst, mode):
            da = L.DwArgs()
            da.src.x, da.src.a, da.src.b, da.src.ld, da.src.C, da.src.relu = src.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
            da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, H, W, k, k, 1, k // 2, 1
            da.w, da.y, da.ldy = wdw.data_ptr(), tb.data_ptr(), Cc
            ca = L.ConvArgs()
            ca.src[0].x, ca.src[0].ld, ca.src[0].C = tb.data_ptr(), Cc, Cc
            ca.nsrc, ca.N, ca.H, ca.W, ca.OH, ca.OW, ca.KH, ca.KW, ca.stride, ca.pad, ca.dil = 1, N, H, W, H, W, 1, 1, 1, 0, 1
            ca.Cout, ca.ldw, ca.cin_total, ca.ldy, ca.w, ca.y = Cc, Cc, Cc, Cc, wpw.data_ptr(), dst.data_ptr()
            calls = [(lb.addk_dw_fwd, (C.byref(da),)), (lb.addk_conv_fwd, (C.byref(ca),))]
            if mode == 'train':
                rows = lb.addk_conv_rows(P, Cc)
                slab = torch.zeros(rows, Cc, 2, dtype=torch.float64, device=dev)
                ca.stats, ca.stats_ld = slab.data_ptr(), Cc
                fin = L.BnFinalizeArgs()
                fin.partial, fin.rows, fin.C, fin.count, fin.gamma, fin.beta = slab.data_ptr(), rows, Cc, float(P), gam.data_ptr(), bet.data_ptr()
                fin.running_mean, fin.running_var, fin.momentum, fin.eps = rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5
                fin.a, fin.b, fin.mean, fin.invstd = fa.data_ptr(), fb.data_ptr(), fm.data_ptr(), fi.data_ptr()
                calls.append((lb.addk_bn_finalize, (C.byref(fin),)))
                keep.extend([slab, fin])
            else:
                tmp = torch.empty(P, Cc, device=dev)
                ca.y = tmp.data_ptr()
                sa = L.AffineSumArgs()
                sa.term[0].x, sa.term[0].a, sa.term[0].b, sa.term[0].ld, sa.term[0].C = tmp.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc
                sa.term[1].x, sa.term[1].ld, sa.term[1].C = u1.data_ptr(), Cc, Cc
                sa.nterm, sa.P, sa.C, sa.out, sa.ldo = 2, P, Cc, dst.data_ptr(), Cc
                calls.append((lb.addk_affine_sum_fwd, (C.byref(sa),)))
                keep.extend([tmp, sa])
            keep.extend([da, ca])
            return calls

        res = {}
        for variant in ('fused', 'fused_sepfin', 'old'):
            for mode in ('train', 'eval'):
                seq = []
                for r in range(REP):
                    src, dst = bufs[r % 2], bufs[(r + 1) % 2]
                    if variant == 'fused':
                        seq.append((lb.addk_sep_fwd, (C.byref(sep_args(src, dst, mode)),)))
                    elif variant == 'fused_sepfin':
                        if mode != 'train':
                            continue
                        seq.extend(sep_args(src, dst, 'train_sepfin'))
                    else:
                        seq.extend(old(src, dst, mode))
                if not seq:
                    continue
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    st = s.cuda_stream
                    for fn, args in seq:
                        L.check(fn(*args, st), 'warm')
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        for fn, args in seq:
                            L.check(fn(*args, s.cuda_stream), 'cap')
                    for _ in range(3):
                        g.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(10):
                        g.replay()
                    e1.record(s)
                    torch.cuda.synchronize()
                    res[(variant, mode)] = e0.elapsed_time(e1) * 1e3 / (10 * REP)
        mb = 2 * P * Cc * 4 / 1e6
        print('N=%d %3dx%-3d C=%-3d k=%d  (in+out %.1f MB)  train: fused %.1f us  fused+finalize launch %.1f us  dw+pw+finalize %.1f us | eval(+sum): fused %.1f us  dw+pw+sum %.1f us' % (
            N, H, W, Cc, k, mb, res[('fused', 'train')], res[('fused_sepfin', 'train')], res[('old', 'train')], res[('fused', 'eval')], res[('old', 'eval')]), flush=True)


if __name__ == '__main__':
    main()
