#!/usr/bin/env python
"""Where the time of one fused SepConv-half launch goes (csrc/sepf.hip, reference operations.py:51-54): a DIAGNOSTIC build of the library
(-DADDK_SEPF_DIAG, built by scripts/sepf_phases.sh into /tmp) stamps s_memrealtime at the phase boundaries of every workgroup; this script
runs single launches of the config-2 cell shapes (inference form with the block-sum epilogue, and the training form), reads the mean phase
lengths and the in-kernel span (earliest workgroup start to latest end) and sets them against the launch's wall time (HIP events over a
chain of dependent launches).   ADDK_LIB=/tmp/addk_diag/libaddk.so python scripts/sepf_phases.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import addk._lib as L                          # noqa: E402

if os.environ.get('ADDK_LIB'):
    L.LIB_PATH = os.environ['ADDK_LIB']


def main():
    lb = L.load()
    raw = C.CDLL(L.LIB_PATH)
    diag = getattr(raw, 'addk_sepf_diag', None)
    dev = torch.device('cuda:0')
    shapes = [(2, 125, 253, 40, 3), (2, 125, 253, 40, 5), (2, 63, 127, 80, 3), (2, 63, 127, 80, 5), (1, 63, 127, 80, 5), (1, 125, 253, 40, 3)]
    torch.manual_seed(0)
    out8 = (C.c_ulonglong * 8)()
    for N, H, W, Cc, k in shapes:
        P = N * H * W
        bufs = [torch.randn(P, Cc, device=dev) for _ in range(2)]
        tb = torch.empty(P, Cc, device=dev)
        a, b = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        wdw, wpw = 0.3 * torch.randn(Cc, k * k, device=dev), 0.2 * torch.randn(Cc, Cc, device=dev)
        u1 = torch.randn(P, Cc, device=dev)
        keep = []

        def args(src, dst, mode):
            ar = L.SepArgs()
            ar.src.x, ar.src.a, ar.src.b, ar.src.ld, ar.src.C, ar.src.relu = src.data_ptr(), a.data_ptr(), b.data_ptr(), Cc, Cc, 1
            ar.N, ar.H, ar.W, ar.K, ar.Cout, ar.ldw = N, H, W, k, Cc, Cc
            ar.dw_w, ar.pw_w, ar.y, ar.ldy = wdw.data_ptr(), wpw.data_ptr(), dst.data_ptr(), Cc
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
            return ar
        for mode in ('eval', 'train'):
            seq = [args(bufs[r % 2], bufs[(r + 1) % 2], mode) for r in range(20)]
            st = torch.cuda.current_stream().cuda_stream
            for ar in seq:
                L.check(lb.addk_sep_fwd(C.byref(ar), st), 'warm')
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for ar in seq:
                    L.check(lb.addk_sep_fwd(C.byref(ar), torch.cuda.current_stream().cuda_stream), 'cap')
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) * 1e3 / 200
            line = 'N=%d %3dx%-3d C=%-3d k=%d %-5s wall %5.1f us/launch' % (N, H, W, Cc, k, mode, wall)
            if diag is not None:
                diag(out8)                                   # reset
                L.check(lb.addk_sep_fwd(C.byref(seq[0]), st), 'one')
                torch.cuda.synchronize()
                diag(out8)
                n = max(1, out8[5])
                ph = [out8[i] / n * 0.01 for i in range(5)]
                line += ' | %d workgroups, in-kernel span %.1f us; mean per workgroup: weights %.2f  patch load+store %.2f  barrier %.2f  compute %.2f  epilogue issue %.2f us (sum %.2f)' % (
                    n, out8[6] * 0.01, ph[0], ph[1], ph[2], ph[3], ph[4], sum(ph))
            print(line, flush=True)
            chain = getattr(raw, 'addk_sepf_chain', None)
            if chain is not None:
                # [r5] the same 20 dependent launches, ONE graph replay, every launch's own clock readings: where the wall time between two
                # dependent launches goes — boundary (last workgroup end -> next launch's first workgroup start), ramp (first -> last
                # workgroup START: the dispatcher placing 512 workgroups of 65 KB LDS at two per CU), body, drain (first -> last END)
                buf = (C.c_ulonglong * 256)()
                nl = C.c_uint()
                chain(buf, C.byref(nl))                      # reset
                g.replay()
                torch.cuda.synchronize()
                chain(buf, C.byref(nl))
                rec = [[buf[4 * i + j] for j in range(4)] for i in range(min(int(nl.value), 20))]
                if len(rec) >= 3:
                    gap = [(rec[i + 1][0] - rec[i][3]) * 0.01 for i in range(1, len(rec) - 1)]
                    ramp = [(r[1] - r[0]) * 0.01 for r in rec[1:]]
                    drain = [(r[3] - r[2]) * 0.01 for r in rec[1:]]
                    span = [(r[3] - r[0]) * 0.01 for r in rec[1:]]
                    first = [(r[2] - r[0]) * 0.01 for r in rec[1:]]
                    period = [(rec[i + 1][0] - rec[i][0]) * 0.01 for i in range(1, len(rec) - 1)]
                    med = lambda v: sorted(v)[len(v) // 2]
                    print('      in the chain (median of %d launches): period %.1f us = span %.1f (ramp: first -> last workgroup start %.1f; first workgroup start -> first end %.1f; '
                          'drain: first -> last end %.1f) + boundary (last end -> next first start) %.1f us' % (len(period), med(period), med(span), med(ramp), med(first), med(drain), med(gap)), flush=True)


if __name__ == '__main__':
    main()
