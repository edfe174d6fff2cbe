// Depthwise k x k convolution (the depthwise halves of SepConv, operations.py:52,56), NHWC fp32.
// HBM-bound: every thread owns one channel quad (float4) for the whole kernel and walks pixels,
// so global accesses are 16 B per lane, contiguous across the lanes of a pixel, and per-channel
// reductions of the backward pass (BN-backward sums, weight gradients) stay in registers until a
// single fixed-order block reduction.  Tap weights sit in LDS transposed to [tap][channel].
#include "common.h"

namespace {

struct DwK {
  addk_src src;
  int N, H, W, OH, OW, KH, KW, stride, pad, dil;
  const float* w;
  float* y; int ldy;
  // backward
  const float* dy; int lddy;
  float* g; int ldg; int accumulate;
  double* dab; float* ws;
  int nq, npl; long P; int vec;
};

constexpr int MAXT = 25;

__global__ void __launch_bounds__(256) dw_fwd_kernel(const DwK p) {
  extern __shared__ float wl[];               // [taps][C4] with C4 = nq*4
  const int taps = p.KH * p.KW, C = p.src.C, C4 = p.nq * 4;
  for (int i = threadIdx.x; i < taps * C4; i += 256) {
    int tp = i / C4, c = i - tp * C4;
    wl[i] = c < C ? p.w[(long)c * taps + tp] : 0.f;
  }
  __syncthreads();
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  if (pl >= p.npl) return;
  const int c = 4 * q, nrem = C - c;
  const int ohw = p.OH * p.OW;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  if (p.src.a) { av = ld4g(p.src.a + c, nrem, p.vec); bv = ld4g(p.src.b + c, nrem, p.vec); }
  const bool relu = p.src.relu != 0;
  const int P = (int)p.P;
  for (int pp = blockIdx.x * p.npl + pl; pp < P; pp += gridDim.x * p.npl) {
    int n = pp / ohw; int rem = pp - n * ohw;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    float4 acc = zero4();
    for (int kh = 0; kh < p.KH; ++kh) {
      int ih = oh * p.stride - p.pad + kh * p.dil;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        int iw = ow * p.stride - p.pad + kw * p.dil;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        float4 v = ld4g(p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + c, nrem, p.vec);
        v.x = fmaf(av.x, v.x, bv.x); v.y = fmaf(av.y, v.y, bv.y); v.z = fmaf(av.z, v.z, bv.z); v.w = fmaf(av.w, v.w, bv.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const float* wt = &wl[(kh * p.KW + kw) * C4 + c];
        acc.x = fmaf(wt[0], v.x, acc.x); acc.y = fmaf(wt[1], v.y, acc.y);
        acc.z = fmaf(wt[2], v.z, acc.z); acc.w = fmaf(wt[3], v.w, acc.w);
      }
    }
    st4g(p.y + (long)pp * p.ldy + c, acc, nrem, p.vec);
  }
}

// LDS-tiled forward for the vector-aligned case.  The kernel above issues its k*k loads one dependent round trip at a
// time (25 for a 5x5) and is latency-bound at ~10 % of HBM speed.  Here a block owns a TH x 16 output tile of one
// channel group (<= 10 channel quads): it stages the haloed input patch in LDS once — all loads of a thread are
// independent and issued back to back, BatchNorm/ReLU applied on the way in, zero padding after it — and every tap
// then reads LDS.
struct DwT {
  DwK k;
  int nqb, ngrp;          // channel quads per block, channel groups
  int TH, PH, PW;         // output rows per tile, patch rows / columns
  int tiles_x, tiles_y;
  int gx, ntiles, rows;   // grid.x of this launch, tiles (backward: walked with stride gx), workspace rows (backward)
  unsigned shbytes;       // dynamic LDS bytes
};
constexpr int DW_TW = 16;

template <int KS>
__device__ __forceinline__ void dw_fwd_tile_body(const DwT& t, float* sm) {
  const DwK& p = t.k;
  const int C4b = t.nqb * 4;
  float* wl = sm;                              // [KS*KS][C4b]
  float* patch = sm + KS * KS * C4b;           // [PH][PW][C4b]
  const int grp = blockIdx.y, cg0 = grp * C4b;
  const int C = p.src.C;
  int b = blockIdx.x;
  const int tx = b % t.tiles_x; b /= t.tiles_x;
  const int ty = b % t.tiles_y; const int n = b / t.tiles_y;
  const int oh0 = ty * t.TH, ow0 = tx * DW_TW;
  const int ih0 = oh0 * p.stride - p.pad, iw0 = ow0 * p.stride - p.pad;
  for (int i = threadIdx.x; i < KS * KS * C4b; i += 256) {
    const int tp = i / C4b, c = cg0 + i - tp * C4b;
    wl[i] = c < C ? p.w[(long)c * (KS * KS) + tp] : 0.f;
  }
  const int npl = 256 / t.nqb;
  const int q = threadIdx.x % t.nqb, pl = threadIdx.x / t.nqb;
  const int c = cg0 + 4 * q;
  const bool cact = pl < npl && c < C;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  if (p.src.a && cact) { av = ld4(p.src.a + c); bv = ld4(p.src.b + c); }
  const bool relu = p.src.relu != 0;
  if (pl < npl) {
    const int npix = t.PH * t.PW;
    const float* xb = p.src.x + (cact ? c : 0);
#pragma unroll 4
    for (int pix = pl; pix < npix; pix += npl) {
      const int pr = pix / t.PW, pc = pix - pr * t.PW;
      const int ih = ih0 + pr, iw = iw0 + pc;
      const bool ok = cact && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      float4 v = ld4(xb + (ok ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld : 0));
      v.x = fmaf(av.x, v.x, bv.x); v.y = fmaf(av.y, v.y, bv.y); v.z = fmaf(av.z, v.z, bv.z); v.w = fmaf(av.w, v.w, bv.w);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      lds_st4(&patch[pix * C4b + 4 * q], v);
    }
  }
  __syncthreads();
  if (!cact) return;
  float4 wr[KS * KS];
#pragma unroll
  for (int tp = 0; tp < KS * KS; ++tp) wr[tp] = lds_ld4(&wl[tp * C4b + 4 * q]);
  const int nout = t.TH * DW_TW;
  for (int o = pl; o < nout; o += npl) {
    const int orow = o / DW_TW, ocol = o - orow * DW_TW;
    const int oh = oh0 + orow, ow = ow0 + ocol;
    if (oh >= p.OH || ow >= p.OW) continue;
    const float* pb = &patch[((orow * p.stride) * t.PW + ocol * p.stride) * C4b + 4 * q];
    float4 acc = zero4();
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const float4 v = lds_ld4(pb + ((kh * p.dil) * t.PW + kw * p.dil) * C4b);
        const float4 w = wr[kh * KS + kw];
        acc.x = fmaf(w.x, v.x, acc.x); acc.y = fmaf(w.y, v.y, acc.y); acc.z = fmaf(w.z, v.z, acc.z); acc.w = fmaf(w.w, v.w, acc.w);
      }
    st4(p.y + ((long)(n * p.OH + oh) * p.OW + ow) * p.ldy + c, acc);
  }
}

template <int KS>
__global__ void __launch_bounds__(256) dw_fwd_tile_kernel(const DwT t) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  dw_fwd_tile_body<KS>(t, sm);
}
// several independent depthwise convs of one dependency level in one launch: block (x, y, z) runs descriptor z
template <int KS>
__global__ void __launch_bounds__(256) dw_fwd_tile_batch_kernel(const DwT* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const DwT t = tab[blockIdx.z];
  if ((int)blockIdx.x >= t.gx || (int)blockIdx.y >= t.ngrp) return;
  dw_fwd_tile_body<KS>(t, sm);
}

// Backward, organised by INPUT pixel: for input pixel p and tap t the output pixel o(p,t) that read p
// through t contributes  w[t]*dy[o]  to dz[p]  and  dy[o]*z[p]  to dW[t]  — one pass yields both.
template <int NT>
__global__ void __launch_bounds__(256) dw_bwd_kernel(const DwK p) {
  extern __shared__ float sm[];               // weights [NT][C4], then reduction tile
  const int C = p.src.C, C4 = p.nq * 4;
  float* wl = sm;
  float* redt = sm + NT * C4;                 // [C4][NT] weight-gradient tile, then [C4][2] dab tile
  for (int i = threadIdx.x; i < NT * C4; i += 256) {
    int tp = i / C4, c = i - tp * C4;
    wl[i] = c < C ? p.w[(long)c * NT + tp] : 0.f;
  }
  __syncthreads();
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q, nrem = C - c;
  const int hw = p.H * p.W;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  if (active && p.src.a) { av = ld4g(p.src.a + c, nrem, p.vec); bv = ld4g(p.src.b + c, nrem, p.vec); }
  const int P = (int)p.P;
  float4 dwacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) dwacc[i] = zero4();
  double sA[4] = {0.0, 0.0, 0.0, 0.0}, sB[4] = {0.0, 0.0, 0.0, 0.0};
  if (active) {
    for (int pp = blockIdx.x * p.npl + pl; pp < P; pp += gridDim.x * p.npl) {
      int n = pp / hw; int rem = pp - n * hw;
      int ih = rem / p.W, iw = rem - ih * p.W;
      float4 x = ld4g(p.src.x + (long)pp * p.src.ld + c, nrem, p.vec);
      float4 zp = make_float4(fmaf(av.x, x.x, bv.x), fmaf(av.y, x.y, bv.y), fmaf(av.z, x.z, bv.z), fmaf(av.w, x.w, bv.w));
      const bool r = p.src.relu != 0;
      bool m0 = !r || zp.x > 0.f, m1 = !r || zp.y > 0.f, m2 = !r || zp.z > 0.f, m3 = !r || zp.w > 0.f;
      float4 z = make_float4(m0 ? zp.x : 0.f, m1 ? zp.y : 0.f, m2 ? zp.z : 0.f, m3 ? zp.w : 0.f);
      float4 dz = zero4();
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) {
        int kh = tp / p.KW, kw = tp - kh * p.KW;
        int th = ih + p.pad - kh * p.dil, tw = iw + p.pad - kw * p.dil;
        bool ok = th >= 0 && tw >= 0;
        int oh = th, ow = tw;
        if (p.stride > 1) { ok = ok && (th % p.stride == 0) && (tw % p.stride == 0); oh = th / p.stride; ow = tw / p.stride; }
        ok = ok && oh < p.OH && ow < p.OW;
        if (ok) {
          float4 d = ld4g(p.dy + ((long)(n * p.OH + oh) * p.OW + ow) * p.lddy + c, nrem, p.vec);
          const float* wt = &wl[tp * C4 + c];
          dz.x = fmaf(wt[0], d.x, dz.x); dz.y = fmaf(wt[1], d.y, dz.y);
          dz.z = fmaf(wt[2], d.z, dz.z); dz.w = fmaf(wt[3], d.w, dz.w);
          dwacc[tp].x = fmaf(d.x, z.x, dwacc[tp].x); dwacc[tp].y = fmaf(d.y, z.y, dwacc[tp].y);
          dwacc[tp].z = fmaf(d.z, z.z, dwacc[tp].z); dwacc[tp].w = fmaf(d.w, z.w, dwacc[tp].w);
        }
      }
      float4 gm = make_float4(m0 ? dz.x : 0.f, m1 ? dz.y : 0.f, m2 ? dz.z : 0.f, m3 ? dz.w : 0.f);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sA[e] += (double)get4(gm, e) * (double)get4(x, e); sB[e] += (double)get4(gm, e); }
      if (p.g) {
        float4 gv = make_float4(gm.x * av.x, gm.y * av.y, gm.z * av.z, gm.w * av.w);
        float* gp = p.g + (long)pp * p.ldg + c;
        if (p.accumulate) { float4 o = ld4g(gp, nrem, p.vec); gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
        st4g(gp, gv, nrem, p.vec);
      }
    }
  }
  // Block reduction over pixel lanes, deterministic and cheap: TG taps at a time every thread drops its float4s into an
  // [TG][npl][C4] LDS panel, then one thread per (tap, channel) adds the npl rows in fixed order (2 barriers per group).
  constexpr int TG = NT == 9 ? 3 : 5;
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TG) {
#pragma unroll
    for (int u = 0; u < TG; ++u)
      if (active) *reinterpret_cast<float4*>(&redt[(u * p.npl + pl) * C4 + c]) = dwacc[t0 + u];
    __syncthreads();
    for (int i = threadIdx.x; i < TG * C; i += 256) {
      const int u = i / C, ch = i - u * C;
      float s = 0.f;
      for (int r = 0; r < p.npl; ++r) s += redt[(u * p.npl + r) * C4 + ch];
      p.ws[((long)blockIdx.x * C + ch) * NT + t0 + u] = s;
    }
    __syncthreads();
  }
  if (p.dab) {
    double* redd = reinterpret_cast<double*>(redt);      // [npl][C4][2] doubles
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { redd[((pl * C4) + c + e) * 2] = sA[e]; redd[((pl * C4) + c + e) * 2 + 1] = sB[e]; }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      double a = 0.0, b = 0.0;
      for (int r = 0; r < p.npl; ++r) { a += redd[(r * C4 + ch) * 2]; b += redd[(r * C4 + ch) * 2 + 1]; }
      p.dab[((long)blockIdx.x * C + ch) * 2] = a;
      p.dab[((long)blockIdx.x * C + ch) * 2 + 1] = b;
    }
  }
}

// LDS-tiled backward (stride 1): a block walks TH x 16 INPUT-pixel tiles of one channel group.  Per tile it stages the
// dy patch those pixels touch ([TH+(k-1)d][16+(k-1)d][channels], zero outside the output map) with independent loads, then
// every tap reads LDS; the weight-gradient and (dA,dB) sums stay in registers across all tiles of the block and are
// reduced once, in the same fixed order as the kernel above.  Workspace rows beyond gridDim.x are zero-filled.
template <int KS>
__device__ __forceinline__ void dw_bwd_tile_body(const DwT& t, float* sm) {
  constexpr int NT = KS * KS;
  const int ntiles = t.ntiles, rows = t.rows, gxs = t.gx;
  const DwK& p = t.k;
  const int C4b = t.nqb * 4, C = p.src.C;
  float* wl = sm;                              // [NT][C4b]
  float* patch = sm + NT * C4b;                // [PH][PW][C4b]; reused as the reduction panel at the end
  const int grp = blockIdx.y, cg0 = grp * C4b;
  for (int i = threadIdx.x; i < NT * C4b; i += 256) {
    const int tp = i / C4b, c = cg0 + i - tp * C4b;
    wl[i] = c < C ? p.w[(long)c * NT + tp] : 0.f;
  }
  const int npl = 256 / t.nqb;
  const int q = threadIdx.x % t.nqb, pl = threadIdx.x / t.nqb;
  const int c = cg0 + 4 * q;
  const bool cact = pl < npl && c < C;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  if (p.src.a && cact) { av = ld4(p.src.a + c); bv = ld4(p.src.b + c); }
  const bool relu = p.src.relu != 0;
  float4 dwacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) dwacc[i] = zero4();
  double sA[4] = {0.0, 0.0, 0.0, 0.0}, sB[4] = {0.0, 0.0, 0.0, 0.0};
  const int halo = (KS - 1) * p.dil;
  for (int tile = blockIdx.x; tile < ntiles; tile += gxs) {
    int b = tile;
    const int tx = b % t.tiles_x; b /= t.tiles_x;
    const int ty = b % t.tiles_y; const int n = b / t.tiles_y;
    const int ih0 = ty * t.TH, iw0 = tx * DW_TW;
    const int oh0 = ih0 + p.pad - halo, ow0 = iw0 + p.pad - halo;     // output position of patch (0, 0)
    __syncthreads();                                                 // previous tile's readers are done (and wl is ready)
    if (pl < npl) {
      const int npix = t.PH * t.PW;
      const float* yb = p.dy + (cact ? c : 0);
#pragma unroll 4
      for (int pix = pl; pix < npix; pix += npl) {
        const int pr = pix / t.PW, pc = pix - pr * t.PW;
        const int oh = oh0 + pr, ow = ow0 + pc;
        const bool ok = cact && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
        float4 v = ld4(yb + (ok ? ((long)(n * p.OH + oh) * p.OW + ow) * p.lddy : 0));
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        lds_st4(&patch[pix * C4b + 4 * q], v);
      }
    }
    __syncthreads();
    if (cact) {
      const int nin = t.TH * DW_TW;
      for (int o = pl; o < nin; o += npl) {
        const int r = o / DW_TW, cc = o - r * DW_TW;
        const int ih = ih0 + r, iw = iw0 + cc;
        if (ih >= p.H || iw >= p.W) continue;
        const long pp = (long)(n * p.H + ih) * p.W + iw;
        const float4 x = ld4(p.src.x + pp * p.src.ld + c);
        const float4 zp = make_float4(fmaf(av.x, x.x, bv.x), fmaf(av.y, x.y, bv.y), fmaf(av.z, x.z, bv.z), fmaf(av.w, x.w, bv.w));
        const bool m0 = !relu || zp.x > 0.f, m1 = !relu || zp.y > 0.f, m2 = !relu || zp.z > 0.f, m3 = !relu || zp.w > 0.f;
        const float4 z = make_float4(m0 ? zp.x : 0.f, m1 ? zp.y : 0.f, m2 ? zp.z : 0.f, m3 ? zp.w : 0.f);
        const float* pb = &patch[(r * t.PW + cc) * C4b + 4 * q];
        float4 dz = zero4();
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const int tp = kh * KS + kw;
            const float4 d = lds_ld4(pb + (((KS - 1 - kh) * p.dil) * t.PW + (KS - 1 - kw) * p.dil) * C4b);
            const float4 w = lds_ld4(&wl[tp * C4b + 4 * q]);
            dz.x = fmaf(w.x, d.x, dz.x); dz.y = fmaf(w.y, d.y, dz.y); dz.z = fmaf(w.z, d.z, dz.z); dz.w = fmaf(w.w, d.w, dz.w);
            dwacc[tp].x = fmaf(d.x, z.x, dwacc[tp].x); dwacc[tp].y = fmaf(d.y, z.y, dwacc[tp].y);
            dwacc[tp].z = fmaf(d.z, z.z, dwacc[tp].z); dwacc[tp].w = fmaf(d.w, z.w, dwacc[tp].w);
          }
        const float4 gm = make_float4(m0 ? dz.x : 0.f, m1 ? dz.y : 0.f, m2 ? dz.z : 0.f, m3 ? dz.w : 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sA[e] += (double)get4(gm, e) * (double)get4(x, e); sB[e] += (double)get4(gm, e); }
        if (p.g) {
          float4 gv = make_float4(gm.x * av.x, gm.y * av.y, gm.z * av.z, gm.w * av.w);
          float* gp = p.g + pp * p.ldg + c;
          if (p.accumulate) { const float4 o4 = ld4(gp); gv.x += o4.x; gv.y += o4.y; gv.z += o4.z; gv.w += o4.w; }
          st4(gp, gv);
        }
      }
    }
  }
  // fixed-order block reduction over the pixel lanes (same scheme as dw_bwd_kernel), channels of this group only
  float* redt = patch;
  const int cgn = (C - cg0 < C4b) ? C - cg0 : C4b;            // valid channels of this group
  constexpr int TG = NT == 9 ? 3 : 5;
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TG) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TG; ++u)
      if (pl < npl) *reinterpret_cast<float4*>(&redt[(u * npl + pl) * C4b + 4 * q]) = cact ? dwacc[t0 + u] : zero4();
    __syncthreads();
    for (int i = threadIdx.x; i < TG * cgn; i += 256) {
      const int u = i / cgn, ch = i - u * cgn;
      float sacc = 0.f;
      for (int r = 0; r < npl; ++r) sacc += redt[(u * npl + r) * C4b + ch];
      p.ws[((long)blockIdx.x * C + cg0 + ch) * NT + t0 + u] = sacc;
      for (int rr = blockIdx.x + gxs; rr < rows; rr += gxs) p.ws[((long)rr * C + cg0 + ch) * NT + t0 + u] = 0.f;
    }
  }
  if (p.dab) {
    __syncthreads();
    double* redd = reinterpret_cast<double*>(redt);      // [npl][C4b][2] doubles
    if (pl < npl) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { redd[((pl * C4b) + 4 * q + e) * 2] = cact ? sA[e] : 0.0; redd[((pl * C4b) + 4 * q + e) * 2 + 1] = cact ? sB[e] : 0.0; }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < cgn; ch += 256) {
      double a = 0.0, b2 = 0.0;
      for (int r = 0; r < npl; ++r) { a += redd[(r * C4b + ch) * 2]; b2 += redd[(r * C4b + ch) * 2 + 1]; }
      p.dab[((long)blockIdx.x * C + cg0 + ch) * 2] = a;
      p.dab[((long)blockIdx.x * C + cg0 + ch) * 2 + 1] = b2;
      for (int rr = blockIdx.x + gxs; rr < rows; rr += gxs) { p.dab[((long)rr * C + cg0 + ch) * 2] = 0.0; p.dab[((long)rr * C + cg0 + ch) * 2 + 1] = 0.0; }
    }
  }
}

template <int KS>
__global__ void __launch_bounds__(256) dw_bwd_tile_kernel(const DwT t) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  dw_bwd_tile_body<KS>(t, sm);
}
template <int KS>
__global__ void __launch_bounds__(256) dw_bwd_tile_batch_kernel(const DwT* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const DwT t = tab[blockIdx.z];
  if ((int)blockIdx.x >= t.gx || (int)blockIdx.y >= t.ngrp) return;
  dw_bwd_tile_body<KS>(t, sm);
}

// one 64-lane wave per weight element: lanes stride over the partial rows, then a fixed-order butterfly
__global__ void __launch_bounds__(256) dw_wreduce_kernel(const float* ws, int rows, int n, float* dw, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  float s = 0.f;
  for (int r = lane; r < rows; r += 64) s += ws[(long)r * n + i];
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) dw[i] = accumulate ? dw[i] + s : s;
}

// batched form: block b = (item, 4 weight elements); the items' tables live in device memory
__global__ void __launch_bounds__(256) dw_wreduce_batch_kernel(const addk_dw_wreduce_item* __restrict__ items, int max_blocks_per_item) {
  // 64 consecutive weight elements x 4 row groups per block: every load is a coalesced 256-byte row segment (lanes along the
  // elements; the wave-per-element form strides its lanes over the rows, 64 cache lines per load), fixed-order combine in LDS
  __shared__ float part[4][64];
  const addk_dw_wreduce_item it = items[blockIdx.y];
  const gfloat* ws = (const gfloat*)it.ws;
  gfloat* dwo = (gfloat*)it.dw;
  const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int i0 = blockIdx.x * 64; i0 < it.n; i0 += max_blocks_per_item * 64) {
    const int i = i0 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < it.n) {
      int r = rg;
      for (; r + 12 < it.rows; r += 16) {
        s0 += ws[(long)r * it.n + i]; s1 += ws[(long)(r + 4) * it.n + i];
        s2 += ws[(long)(r + 8) * it.n + i]; s3 += ws[(long)(r + 12) * it.n + i];
      }
      for (; r < it.rows; r += 4) s0 += ws[(long)r * it.n + i];
    }
    part[rg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && i < it.n) {
      const float s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
      dwo[i] = it.accumulate ? dwo[i] + s : s;
    }
    __syncthreads();
  }
}

int dw_rows(long P, int C) {
  EwMap m = ew_map(C);
  long r = P / ((long)m.npl * 2);      // >= 4 resident blocks per CU on the level-1 maps
  if (r < 1) r = 1;
  if (r > 1024) r = 1024;
  return (int)r;
}

}  // namespace

extern "C" int addk_dw_rows(int64_t P, int32_t C) { return dw_rows(P, C); }

static int dw_fill(DwK& k, const addk_src& src, int N, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int dil) {
  ADDK_REQUIRE(src.x && src.C > 0 && src.C <= 1024 && src.ld >= src.C, "dw: bad source");
  ADDK_REQUIRE((src.a == nullptr) == (src.b == nullptr), "dw: a/b must come together");
  ADDK_REQUIRE(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && KH > 0 && KW > 0 && KH * KW <= MAXT && stride > 0 && dil > 0, "dw: bad geometry");
  ADDK_REQUIRE((long)N * H * W < (1L << 30) && (long)N * OH * OW < (1L << 30), "dw: tensor too large for 32-bit pixel indexing");
  k.src = src; k.N = N; k.H = H; k.W = W; k.OH = OH; k.OW = OW; k.KH = KH; k.KW = KW; k.stride = stride; k.pad = pad; k.dil = dil;
  EwMap m = ew_map(src.C); k.nq = m.nq; k.npl = m.npl;
  return 0;
}


// tile geometry of the LDS-tiled kernels; false: the launch is not covered (generic kernel)
static bool dw_tile_fwd(const addk_dw_args* a, const DwK& k, DwT& t) {
  if (!(k.vec && a->KH == a->KW && (a->KH == 3 || a->KH == 5) && (addk_get_fast_paths() & ADDK_FAST_DWTILE) && k.P >= 2048)) return false;
  t.k = k;
  t.nqb = k.nq <= 10 ? k.nq : (k.nq % 10 == 0 ? 10 : 8);
  t.ngrp = cdiv(k.nq, t.nqb);
  const int span = (a->KH - 1) * a->dil + 1;
  t.PW = (DW_TW - 1) * a->stride + span;
  const long row_bytes = (long)t.PW * t.nqb * 16;
  const int ph = (int)((48 * 1024) / row_bytes);                       // patch rows that fit the LDS budget
  int th = (ph - span) / a->stride + 1;
  if (th > 16) th = 16;
  if (th > a->OH) th = a->OH;
  if (th < 2) return false;
  t.TH = th; t.PH = (th - 1) * a->stride + span;
  t.tiles_x = cdiv(a->OW, DW_TW); t.tiles_y = cdiv(a->OH, th);
  t.ntiles = a->N * t.tiles_y * t.tiles_x; t.gx = t.ntiles; t.rows = 0;
  t.shbytes = (unsigned)(((size_t)a->KH * a->KW * t.nqb * 4 + (size_t)t.PH * t.PW * t.nqb * 4) * sizeof(float));
  return true;
}
static bool dw_tile_bwd(const addk_dw_bwd_args* a, const DwK& k, int rows, DwT& t) {
  if (!(k.vec && a->stride == 1 && a->KH == a->KW && (a->KH == 3 || a->KH == 5) && (addk_get_fast_paths() & ADDK_FAST_DWTILE) && k.P >= 2048)) return false;
  t.k = k;
  t.nqb = k.nq <= 10 ? k.nq : (k.nq % 10 == 0 ? 10 : 8);
  t.ngrp = cdiv(k.nq, t.nqb);
  const int halo = (a->KH - 1) * a->dil, taps = a->KH * a->KW;
  t.PW = DW_TW + halo;
  const long row_bytes = (long)t.PW * t.nqb * 16;
  int th = (int)((40 * 1024) / row_bytes) - halo;
  if (th > 16) th = 16;
  if (th > a->H) th = a->H;
  if (th < 2) return false;
  const int npl = 256 / t.nqb;
  const size_t panel = (size_t)5 * npl * t.nqb * 4 * sizeof(float);          // [TG<=5][npl][C4b] floats >= [npl][C4b][2] doubles
  t.TH = th; t.PH = th + halo;
  t.tiles_x = cdiv(a->W, DW_TW); t.tiles_y = cdiv(a->H, th);
  t.ntiles = a->N * t.tiles_y * t.tiles_x;
  t.rows = rows; t.gx = t.ntiles < rows ? t.ntiles : rows;
  size_t pbytes = (size_t)t.PH * t.PW * t.nqb * 16;
  if (pbytes < panel) pbytes = panel;
  t.shbytes = (unsigned)((size_t)taps * t.nqb * 16 + pbytes);
  return true;
}

extern "C" int addk_dw_fwd(const addk_dw_args* a, void* stream) {
  ADDK_REQUIRE(a && a->w && a->y && a->ldy >= a->src.C, "dw_fwd: null/short output");
  DwK k{};
  int rc = dw_fill(k, a->src, a->N, a->H, a->W, a->OH, a->OW, a->KH, a->KW, a->stride, a->pad, a->dil);
  if (rc) return rc;
  k.w = a->w; k.y = a->y; k.ldy = a->ldy;
  k.P = (long)a->N * a->OH * a->OW;
  k.vec = src_vec_ok(a->src) && aligned16(a->y) && a->ldy % 4 == 0;
  DwT t;
  if (dw_tile_fwd(a, k, t)) {
    dim3 grid((unsigned)t.gx, (unsigned)t.ngrp);
    if (a->KH == 3) hipLaunchKernelGGL(dw_fwd_tile_kernel<3>, grid, dim3(256), t.shbytes, (hipStream_t)stream, t);
    else            hipLaunchKernelGGL(dw_fwd_tile_kernel<5>, grid, dim3(256), t.shbytes, (hipStream_t)stream, t);
    return addk_check_launch("dw_fwd_tile");
  }
  long blocks = cdiv(k.P, k.npl); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
  size_t sh = (size_t)a->KH * a->KW * k.nq * 4 * sizeof(float);
  hipLaunchKernelGGL(dw_fwd_kernel, dim3((unsigned)blocks), dim3(256), sh, (hipStream_t)stream, k);
  return addk_check_launch("dw_fwd");
}

extern "C" int addk_dw_bwd(const addk_dw_bwd_args* a, void* stream) {
  ADDK_REQUIRE(a && a->dy && a->w && a->dw && a->ws && a->lddy >= a->src.C, "dw_bwd: null pointer");
  ADDK_REQUIRE(!a->g || a->ldg >= a->src.C, "dw_bwd: short ldg");
  DwK k{};
  int rc = dw_fill(k, a->src, a->N, a->H, a->W, a->OH, a->OW, a->KH, a->KW, a->stride, a->pad, a->dil);
  if (rc) return rc;
  k.w = a->w; k.dy = a->dy; k.lddy = a->lddy; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate;
  k.dab = (double*)a->dab; k.ws = a->ws;
  k.P = (long)a->N * a->H * a->W;
  k.vec = src_vec_ok(a->src) && aligned16(a->dy) && a->lddy % 4 == 0 && (!a->g || (aligned16(a->g) && a->ldg % 4 == 0));
  const int taps = a->KH * a->KW, C4 = k.nq * 4;
  const int rows = dw_rows(k.P, a->src.C);
  size_t sh = (size_t)(taps * C4 + k.npl * C4 * 5) * sizeof(float);     // tap weights + [TG<=5][npl][C4] reduction panel (>= the fp64 (dA,dB) panel)
  hipStream_t st = (hipStream_t)stream;
  bool tiled = false;
  DwT t;
  if (dw_tile_bwd(a, k, rows, t)) {
    dim3 grid((unsigned)t.gx, (unsigned)t.ngrp);
    if (taps == 9) hipLaunchKernelGGL(dw_bwd_tile_kernel<3>, grid, dim3(256), t.shbytes, st, t);
    else           hipLaunchKernelGGL(dw_bwd_tile_kernel<5>, grid, dim3(256), t.shbytes, st, t);
    tiled = true;
  }
  if (tiled) {}
  else if (taps == 9) hipLaunchKernelGGL(dw_bwd_kernel<9>, dim3(rows), dim3(256), sh, st, k);
  else if (taps == 25) hipLaunchKernelGGL(dw_bwd_kernel<25>, dim3(rows), dim3(256), sh, st, k);
  else { addk_set_error("dw_bwd: only 3x3 and 5x5 depthwise kernels are built"); return ADDK_ERR_UNSUPPORTED; }
  rc = addk_check_launch("dw_bwd");
  if (rc) return rc;
  if (a->defer_wreduce) return ADDK_OK;
  int n = a->src.C * taps;
  hipLaunchKernelGGL(dw_wreduce_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, a->ws, rows, n, a->dw, a->dw_accumulate);
  return addk_check_launch("dw_wreduce");
}

extern "C" int addk_dw_wreduce_batch(const addk_dw_wreduce_item* dev_items, int32_t n_items, void* stream) {
  ADDK_REQUIRE(dev_items && n_items > 0, "dw_wreduce_batch: bad args");
  const int per = 64;           // blocks per item: 4096 weight elements per pass (C*taps <= 160*25 = 4000)
  hipLaunchKernelGGL(dw_wreduce_batch_kernel, dim3(per, n_items), dim3(256), 0, (hipStream_t)stream, dev_items, per);
  return addk_check_launch("dw_wreduce_batch");
}

// ---- batched form of the LDS-tiled kernels: the independent depthwise convs of one dependency level in one launch --------
static bool dw_fill_fwd_k(const addk_dw_args* a, DwK& k) {
  if (!a || !a->w || !a->y || a->ldy < a->src.C) return false;
  if (dw_fill(k, a->src, a->N, a->H, a->W, a->OH, a->OW, a->KH, a->KW, a->stride, a->pad, a->dil)) return false;
  k.w = a->w; k.y = a->y; k.ldy = a->ldy;
  k.P = (long)a->N * a->OH * a->OW;
  k.vec = src_vec_ok(a->src) && aligned16(a->y) && a->ldy % 4 == 0;
  return true;
}
static bool dw_fill_bwd_k(const addk_dw_bwd_args* a, DwK& k) {
  if (!a || !a->dy || !a->w || !a->dw || !a->ws || a->lddy < a->src.C || (a->g && a->ldg < a->src.C)) return false;
  if (dw_fill(k, a->src, a->N, a->H, a->W, a->OH, a->OW, a->KH, a->KW, a->stride, a->pad, a->dil)) return false;
  k.w = a->w; k.dy = a->dy; k.lddy = a->lddy; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate;
  k.dab = (double*)a->dab; k.ws = a->ws;
  k.P = (long)a->N * a->H * a->W;
  k.vec = src_vec_ok(a->src) && aligned16(a->dy) && a->lddy % 4 == 0 && (!a->g || (aligned16(a->g) && a->ldg % 4 == 0));
  return true;
}
// key >= 0 (= kernel size | backward << 4): the launch runs on the tiled kernel and can share a batch with equal keys
extern "C" int addk_dw_fwd_batch_key(const addk_dw_args* a) {
  DwK k{}; DwT t;
  return (dw_fill_fwd_k(a, k) && dw_tile_fwd(a, k, t)) ? a->KH : -1;
}
extern "C" int addk_dw_bwd_batch_key(const addk_dw_bwd_args* a) {
  DwK k{}; DwT t;
  if (!a || !a->defer_wreduce) return -1;               // the batched launch has no per-conv weight reduction
  return (dw_fill_bwd_k(a, k) && dw_tile_bwd(a, k, dw_rows(k.P, a->src.C), t)) ? (a->KH | 16) : -1;
}
template <typename Args, typename Setup>
static int64_t dw_batch_prepare(const Args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta, int bwd, Setup setup) {
  if (!a || n <= 0 || !meta) { addk_set_error("dw_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  const int64_t total = (int64_t)n * sizeof(DwT);
  if (host_blob && blob_bytes < total) { addk_set_error("dw_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  int ks = 0, gx = 0, gy = 0; unsigned sh = 0;
  for (int i = 0; i < n; ++i) {
    DwT t;
    if (!setup(&a[i], t)) { addk_set_error("dw_batch_prepare: launch %d is not a tiled-kernel shape", i); return ADDK_ERR_INVALID; }
    if (i == 0) ks = a[i].KH;
    if (a[i].KH != ks) { addk_set_error("dw_batch_prepare: mixed kernel sizes"); return ADDK_ERR_INVALID; }
    if (t.gx > gx) gx = t.gx;
    if (t.ngrp > gy) gy = t.ngrp;
    if (t.shbytes > sh) sh = t.shbytes;
    if (host_blob) reinterpret_cast<DwT*>(host_blob)[i] = t;
  }
  meta[0] = ks | (bwd << 4); meta[1] = n; meta[2] = gx; meta[3] = gy; meta[4] = sh;
  return total;
}
extern "C" int64_t addk_dw_fwd_batch_prepare(const addk_dw_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  return dw_batch_prepare(a, n, host_blob, blob_bytes, meta, 0, [](const addk_dw_args* x, DwT& t) { DwK k{}; return dw_fill_fwd_k(x, k) && dw_tile_fwd(x, k, t); });
}
extern "C" int64_t addk_dw_bwd_batch_prepare(const addk_dw_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  return dw_batch_prepare(a, n, host_blob, blob_bytes, meta, 1, [](const addk_dw_bwd_args* x, DwT& t) {
    DwK k{}; return x->defer_wreduce && dw_fill_bwd_k(x, k) && dw_tile_bwd(x, k, dw_rows(k.P, x->src.C), t); });
}
extern "C" int addk_dw_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[1] > 0 && meta[2] > 0 && meta[3] > 0, "dw_batch_run: bad args");
  const int ks = (int)meta[0] & 15, bwd = ((int)meta[0] >> 4) & 1;
  const DwT* tab = reinterpret_cast<const DwT*>(dev_blob);
  dim3 grid((unsigned)meta[2], (unsigned)meta[3], (unsigned)meta[1]);
  const size_t sh = (size_t)meta[4];
  hipStream_t st = (hipStream_t)stream;
  if (!bwd) {
    if (ks == 3) hipLaunchKernelGGL(dw_fwd_tile_batch_kernel<3>, grid, dim3(256), sh, st, tab);
    else         hipLaunchKernelGGL(dw_fwd_tile_batch_kernel<5>, grid, dim3(256), sh, st, tab);
  } else {
    if (ks == 3) hipLaunchKernelGGL(dw_bwd_tile_batch_kernel<3>, grid, dim3(256), sh, st, tab);
    else         hipLaunchKernelGGL(dw_bwd_tile_batch_kernel<5>, grid, dim3(256), sh, st, tab);
  }
  return addk_check_launch("dw_batch");
}
