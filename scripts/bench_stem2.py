#!/usr/bin/env python
"""stem2 forward (3x3, stride 2, 64 -> 128 at 2x512x1024) through the C ABI: split-bf16 kernel against the generic fp32-MFMA kernel."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import addk                                    # noqa: E402
import addk._lib as L                          # noqa: E402


def main():
    lb = L.load()
    dev = torch.device('cuda:0')
    N, H, W, Ci, Co = 2, 512, 1024, 64, 128
    OH, OW = H // 2, W // 2
    P = N * OH * OW
    x = torch.randn(N * H * W, Ci, device=dev)
    a, b = torch.rand(Ci, device=dev) + 0.5, 0.1 * torch.randn(Ci, device=dev)
    w = 0.1 * torch.randn(Co, 9 * Ci, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for tag, fast in (('split', 31), ('generic', 0)):
        lb.addk_set_fast_paths(fast)
        ar = L.ConvArgs()
        ar.src[0].x, ar.src[0].a, ar.src[0].b, ar.src[0].ld, ar.src[0].C, ar.src[0].relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), Ci, Ci, 1
        ar.nsrc = 1
        ar.N, ar.H, ar.W, ar.OH, ar.OW, ar.KH, ar.KW, ar.stride, ar.pad, ar.dil, ar.Cout = N, H, W, OH, OW, 3, 3, 2, 1, 1, Co
        ar.ldw, ar.cin_total, ar.w_choff, ar.ldy = 9 * Ci, Ci, 0, Co
        y = torch.empty(P, Co, device=dev)
        rows = lb.addk_conv_rows(P, Co)
        slab = torch.zeros(rows, Co, 2, device=dev, dtype=torch.float64)
        ar.w, ar.y, ar.stats, ar.stats_ld = w.data_ptr(), y.data_ptr(), slab.data_ptr(), Co
        npk = int(lb.addk_conv_fwd_pack_floats(C.byref(ar)))
        wp = torch.empty(max(npk, 1), device=dev)
        if npk:
            ar.wpack, ar.wpack_floats = wp.data_ptr(), npk
        for _ in range(3):
            L.check(lb.addk_conv_fwd(C.byref(ar), st), 'conv_fwd')
        if npk:
            ar.wpack_ready = 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(lb.addk_conv_fwd(C.byref(ar), st), 'conv_fwd')
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%-8s forward        pack floats %8d  %.3f ms  %.1f TFLOP/s (38.7 GF)' % (tag, npk, ms, 2 * P * Co * 9 * Ci / ms / 1e9))
        da = L.ConvDgradArgs()
        dy = torch.randn(P, Co, device=dev)
        da.dy, da.lddy, da.Cout = dy.data_ptr(), Co, Co
        da.N, da.H, da.W, da.OH, da.OW, da.KH, da.KW, da.stride, da.pad, da.dil = N, H, W, OH, OW, 3, 3, 2, 1, 1
        da.w, da.ldw, da.cin_total, da.w_choff = w.data_ptr(), 9 * Ci, Ci, 0
        da.dst = ar.src[0]
        g = torch.empty(N * H * W, Ci, device=dev)
        r2 = lb.addk_conv_rows(N * H * W, Ci)
        dab = torch.zeros(r2, Ci, 2, device=dev, dtype=torch.float64)
        da.g, da.ldg, da.accumulate, da.dab = g.data_ptr(), Ci, 0, dab.data_ptr()
        npk = int(lb.addk_conv_dgrad_pack_floats(C.byref(da)))
        dp = torch.empty(max(npk, 1), device=dev)
        if npk:
            da.wpack, da.wpack_floats = dp.data_ptr(), npk
        for _ in range(3):
            L.check(lb.addk_conv_dgrad(C.byref(da), st), 'conv_dgrad')
        if npk:
            da.wpack_ready = 1
        e0.record()
        for _ in range(20):
            L.check(lb.addk_conv_dgrad(C.byref(da), st), 'conv_dgrad')
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%-8s data gradient  pack floats %8d  %.3f ms  %.1f TFLOP/s (rows %d)' % (tag, npk, ms, 2 * P * Co * 9 * Ci / ms / 1e9, r2))


if __name__ == '__main__':
    main()
