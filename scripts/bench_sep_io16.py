#!/usr/bin/env python
"""Storage experiment for VERDICT r02 item 3a: the fused SepConv half (csrc/sepf.hip) with its activation tensors stored as bf16
(fp32 arithmetic, fp32 BatchNorm statistics) against the fp32-storage product path, at the four cell shapes of config 2:
device time per launch (hipGraph of dependent repetitions, HIP events) and the numerical price.  python scripts/bench_sep_io16.py"""
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
    torch.manual_seed(0)
    for N, H, W, Cc, k in [(2, 128, 256, 40, 3), (2, 128, 256, 40, 5), (2, 64, 128, 80, 3), (2, 64, 128, 80, 5)]:
        P = N * H * W
        a, b = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        wdw, wpw = 0.3 * torch.randn(Cc, k * k, device=dev), 0.2 * torch.randn(Cc, Cc, device=dev)
        x32 = torch.randn(P, Cc, device=dev).bfloat16().float()      # bf16-representable inputs: both storages see the same numbers
        u32 = torch.randn(P, Cc, device=dev).bfloat16().float()
        res, outs = {}, {}
        for io16 in (0, 1):
            dt = torch.bfloat16 if io16 else torch.float32
            bufs = [x32.to(dt).clone(), torch.empty(P, Cc, device=dev, dtype=dt)]
            tb = torch.empty(P, Cc, device=dev, dtype=dt)
            u1 = u32.to(dt).clone()
            keep = []
            for mode in ('train', 'eval'):
                bufs[0].copy_(x32.to(dt))
                seq = []
                for r in range(REP):
                    src, dst = bufs[r % 2], bufs[(r + 1) % 2]
                    ar = L.SepArgs()
                    ar.src.x, ar.src.a, ar.src.b, ar.src.ld, ar.src.C, ar.src.relu = src.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
                    ar.N, ar.H, ar.W, ar.K, ar.Cout, ar.ldw = N, H, W, k, Cc, Cc
                    ar.dw_w, ar.pw_w, ar.y, ar.ldy, ar.io16 = wdw.data_ptr(), wpw.data_ptr(), dst.data_ptr(), Cc, io16
                    if mode == 'train':
                        rows = max(lb.addk_conv_rows(P, Cc), lb.addk_sep_rows(C.byref(ar)))
                        slab = torch.zeros(rows, Cc, 2, dtype=torch.float64, device=dev)
                        ar.t, ar.ldt, ar.stats, ar.stats_ld, ar.stats_rows = tb.data_ptr(), Cc, slab.data_ptr(), Cc, rows
                        keep.append(slab)
                    else:
                        ar.ea, ar.eb, ar.nterm = a.data_ptr(), b.data_ptr(), 1
                        ar.term[0].x, ar.term[0].ld, ar.term[0].C = u1.data_ptr(), Cc, Cc
                    assert lb.addk_sep_fwd_supported(C.byref(ar)) == 1
                    keep.append(ar)
                    seq.append(ar)
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    L.check(lb.addk_sep_fwd(C.byref(seq[0]), s.cuda_stream), 'first')
                    torch.cuda.synchronize()
                    outs[(io16, mode)] = bufs[1].float().clone()
                    for ar in seq:
                        L.check(lb.addk_sep_fwd(C.byref(ar), s.cuda_stream), 'warm')
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        for ar in seq:
                            L.check(lb.addk_sep_fwd(C.byref(ar), s.cuda_stream), 'cap')
                    for _ in range(3):
                        g.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(10):
                        g.replay()
                    e1.record(s)
                    torch.cuda.synchronize()
                    res[(io16, mode)] = e0.elapsed_time(e1) * 1e3 / (10 * REP)
                bufs[0].copy_(x32.to(dt))
        err = {m: float((outs[(1, m)] - outs[(0, m)]).abs().max() / outs[(0, m)].abs().max()) for m in ('train', 'eval')}
        print('N=%d %3dx%-3d C=%-3d k=%d  train (raw + depthwise output + statistics): fp32 storage %.1f us  bf16 storage %.1f us | '
              'eval (+ BatchNorm + 1 term): fp32 %.1f us  bf16 %.1f us | max difference of the outputs / max: train %.1e eval %.1e' % (
                  N, H, W, Cc, k, res[(0, 'train')], res[(1, 'train')], res[(0, 'eval')], res[(1, 'eval')], err['train'], err['eval']), flush=True)


if __name__ == '__main__':
    main()
