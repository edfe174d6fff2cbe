// Bilinear resize, align_corners=False, no antialias — F.interpolate(mode='bilinear') as called at
// ADD.py:76-77,84,89,317 and decoder.py:24,28 (used for both up- and down-sampling).
// Index arithmetic follows ATen's area_pixel_compute_source_index in fp32:
//     scale = in/out;  src = max(0, scale*(dst+0.5)-0.5);  i0 = floor(src); i1 = min(i0+1, in-1); l1 = src-i0.
// Forward: one thread per (output pixel, channel quad), 16 B per lane.
// Backward is a GATHER (deterministic, no atomics): an input pixel enumerates the few output rows/cols
// whose i0/i1 hit it, recomputing exactly the forward weights.
#include "common.h"

namespace {

// candidate output range [lo, hi] whose taps may touch input index i
__device__ __forceinline__ void out_range(int i, float scale, int out, int& lo, int& hi) {
  float inv = 1.f / scale;
  float a = ((float)i - 1.f + 0.5f) * inv - 0.5f;
  float b = ((float)i + 1.f + 0.5f) * inv - 0.5f;
  lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// weight with which output index o reads input index i (0 when it does not)
__device__ __forceinline__ float tap_weight(int o, int i, float scale, int in) {
  int i0, i1; float l0, l1;
  src_index(o, scale, in, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

struct RsK {
  addk_src src; int N, H, W, OH, OW;
  float* y; int ldy; int nchw;
  const float* dy; int lddy; const float* dy_scale;
  float* g; int ldg; int accumulate; double* dab;
  int nq, npl, vec; long P;
};

__global__ void __launch_bounds__(256) resize_fwd_kernel(const RsK p) {
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  if (pl >= p.npl) return;
  const int c = 4 * q, nrem = p.src.C - c;
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  const long ohw = (long)p.OH * p.OW;
  const bool relu = p.src.relu != 0;
  for (long pp = (long)blockIdx.x * p.npl + pl; pp < p.P; pp += (long)gridDim.x * p.npl) {
    int n = (int)(pp / ohw); int rem = (int)(pp - (long)n * ohw);
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    int h0, h1, w0, w1; float lh0, lh1, lw0, lw1;
    src_index(oh, sh, p.H, h0, h1, lh0, lh1);
    src_index(ow, sw, p.W, w0, w1, lw0, lw1);
    const float* b = p.src.x + (long)n * p.H * p.W * p.src.ld + c;
    float4 v00 = prologue4(ld4g(b + ((long)h0 * p.W + w0) * p.src.ld, nrem, p.vec), p.src.a, p.src.b, c, nrem, relu, p.vec);
    float4 v01 = prologue4(ld4g(b + ((long)h0 * p.W + w1) * p.src.ld, nrem, p.vec), p.src.a, p.src.b, c, nrem, relu, p.vec);
    float4 v10 = prologue4(ld4g(b + ((long)h1 * p.W + w0) * p.src.ld, nrem, p.vec), p.src.a, p.src.b, c, nrem, relu, p.vec);
    float4 v11 = prologue4(ld4g(b + ((long)h1 * p.W + w1) * p.src.ld, nrem, p.vec), p.src.a, p.src.b, c, nrem, relu, p.vec);
    const float4 o = lerp4(v00, v01, v10, v11, lh0, lh1, lw0, lw1);
    st4g(p.y + pp * p.ldy + c, o, nrem, p.vec);
  }
}

// NCHW destination (final logits, decoder.py:28): one thread per output pixel, loop over channels so that
// stores are contiguous along W for every channel plane.
__global__ void __launch_bounds__(256) resize_fwd_nchw_kernel(const RsK p) {
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  const long ohw = (long)p.OH * p.OW;
  const int C = p.src.C;
  for (long pp = (long)blockIdx.x * blockDim.x + threadIdx.x; pp < p.P; pp += (long)gridDim.x * blockDim.x) {
    int n = (int)(pp / ohw); long rem = pp - (long)n * ohw;
    int oh = (int)(rem / p.OW), ow = (int)(rem - (long)oh * p.OW);
    int h0, h1, w0, w1; float lh0, lh1, lw0, lw1;
    src_index(oh, sh, p.H, h0, h1, lh0, lh1);
    src_index(ow, sw, p.W, w0, w1, lw0, lw1);
    const float* b = p.src.x + (long)n * p.H * p.W * p.src.ld;
    const float* p00 = b + ((long)h0 * p.W + w0) * p.src.ld;
    const float* p01 = b + ((long)h0 * p.W + w1) * p.src.ld;
    const float* p10 = b + ((long)h1 * p.W + w0) * p.src.ld;
    const float* p11 = b + ((long)h1 * p.W + w1) * p.src.ld;
    float* yo = p.y + (long)n * C * ohw + rem;
    for (int c = 0; c < C; ++c) {
      float a = p.src.a ? p.src.a[c] : 1.f, bb = p.src.b ? p.src.b[c] : 0.f;
      float v00 = fmaf(a, p00[c], bb), v01 = fmaf(a, p01[c], bb), v10 = fmaf(a, p10[c], bb), v11 = fmaf(a, p11[c], bb);
      if (p.src.relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
      yo[(long)c * ohw] = lh0 * (lw0 * v00 + lw1 * v01) + lh1 * (lw0 * v10 + lw1 * v11);
    }
  }
}


// Backward of the final logits up-sampling (NCHW gradient in, NHWC gradient out; decoder.py:28), LDS-tiled and SEPARABLE:
// a block owns an 8x8 tile of low-resolution pixels of one channel plane, stages the block of the high-resolution
// gradient those pixels receive from (<= RB_PM x RB_PM, contiguous row segments, independent loads), reduces it along W
// with the column weights (t1[row][j]) and then along H with the row weights: 2*T taps per pixel instead of T*T
// (T = 21 at the x8 up-sampling of config 2).  The thread-per-pixel kernel below walks its T*T x 19 taps as dependent
// global loads (1.17 ms per exit at 1024x2048).
constexpr int RB_T = 8, RB_PM = 88, RB_MAXT = 22;

__global__ void __launch_bounds__(256) resize_bwd_nchw_tile_kernel(const RsK p, int tiles_x, int tiles_y) {
  __shared__ float patch[RB_PM][RB_PM + 1];
  __shared__ float t1[RB_PM][RB_T];
  __shared__ float wtw[RB_T][RB_MAXT], wth[RB_T][RB_MAXT];
  __shared__ int lo_w[RB_T], lo_h[RB_T], n_w[RB_T], n_h[RB_T];
  const int C = p.src.C;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; const int n = b / tiles_y;
  const int c = blockIdx.y;
  const int ih0 = ty * RB_T, iw0 = tx * RB_T;
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  int rlo, rhi, clo, chi, tmp;
  out_range(ih0, sh, p.OH, rlo, tmp);
  out_range(min(ih0 + RB_T - 1, p.H - 1), sh, p.OH, tmp, rhi);
  out_range(iw0, sw, p.OW, clo, tmp);
  out_range(min(iw0 + RB_T - 1, p.W - 1), sw, p.OW, tmp, chi);
  const int PH = rhi - rlo + 1, PW = chi - clo + 1;          // host guarantees <= RB_PM
  const float* plane = p.dy + ((long)n * C + c) * ((long)p.OH * p.OW) + (long)rlo * p.OW + clo;
  const int per = PH * PW;
#pragma unroll 4
  for (int i = threadIdx.x; i < per; i += 256) {
    const int r = i / PW, cc = i - r * PW;
    patch[r][cc] = plane[(long)r * p.OW + cc];
  }
  // per-axis weight tables of the tile's 8 columns / 8 rows
  if (threadIdx.x < 2 * RB_T) {
    const bool isw = threadIdx.x < RB_T;
    const int j = threadIdx.x & (RB_T - 1);
    const int i = (isw ? iw0 : ih0) + j;
    const int lim = isw ? p.W : p.H;
    int lo = 0, hi = -1;
    if (i < lim) out_range(i, isw ? sw : sh, isw ? p.OW : p.OH, lo, hi);
    if (isw) { lo_w[j] = lo; n_w[j] = hi - lo + 1; } else { lo_h[j] = lo; n_h[j] = hi - lo + 1; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * RB_T * RB_MAXT; i += 256) {
    const bool isw = i < RB_T * RB_MAXT;
    const int r = isw ? i : i - RB_T * RB_MAXT;
    const int j = r / RB_MAXT, k = r - j * RB_MAXT;
    if (isw) wtw[j][k] = k < n_w[j] ? tap_weight(lo_w[j] + k, iw0 + j, sw, p.W) : 0.f;
    else     wth[j][k] = k < n_h[j] ? tap_weight(lo_h[j] + k, ih0 + j, sh, p.H) : 0.f;
  }
  __syncthreads();
  // pass 1: along W
  for (int i = threadIdx.x; i < PH * RB_T; i += 256) {
    const int r = i / RB_T, j = i - r * RB_T;
    const float* row = &patch[r][lo_w[j] - clo];
    float acc = 0.f;
    const int nw = n_w[j];
#pragma unroll
    for (int k = 0; k < RB_MAXT; ++k) if (k < nw) acc = fmaf(wtw[j][k], row[k], acc);
    t1[r][j] = acc;
  }
  __syncthreads();
  // pass 2: along H, one thread per pixel of the tile
  if (threadIdx.x < RB_T * RB_T) {
    const int i = threadIdx.x / RB_T, j = threadIdx.x - i * RB_T;
    const int ih = ih0 + i, iw = iw0 + j;
    if (ih < p.H && iw < p.W) {
      const int nh = n_h[i], r0 = lo_h[i] - rlo;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < RB_MAXT; ++k) if (k < nh) acc = fmaf(wth[i][k], t1[r0 + k][j], acc);
      const float gs = p.dy_scale ? *p.dy_scale : 1.f;
      const float a = p.src.a ? p.src.a[c] : 1.f;
      float* gp = p.g + ((long)(n * p.H + ih) * p.W + iw) * p.ldg + c;
      const float gv = acc * gs * a;
      *gp = p.accumulate ? *gp + gv : gv;
    }
  }
}

template <bool NCHW>
__global__ void __launch_bounds__(256) resize_bwd_kernel(const RsK p) {
  extern __shared__ double redt[];       // [C4][2]
  const int C = p.src.C;
  const int q = NCHW ? 0 : threadIdx.x % p.nq, pl = NCHW ? threadIdx.x : threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q, nrem = C - c;
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  const long hw = (long)p.H * p.W, ohw = (long)p.OH * p.OW;
  const float gs = p.dy_scale ? *p.dy_scale : 1.f;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  double sA[4] = {0.0, 0.0, 0.0, 0.0}, sB[4] = {0.0, 0.0, 0.0, 0.0};
  if (!NCHW && active && p.src.a) { av = ld4g(p.src.a + c, nrem, p.vec); bv = ld4g(p.src.b + c, nrem, p.vec); }
  if (active) {
    for (long pp = (long)blockIdx.x * p.npl + pl; pp < p.P; pp += (long)gridDim.x * p.npl) {
      int n = (int)(pp / hw); int rem = (int)(pp - (long)n * hw);
      int ih = rem / p.W, iw = rem - ih * p.W;
      int hlo, hhi, wlo, whi;
      out_range(ih, sh, p.OH, hlo, hhi);
      out_range(iw, sw, p.OW, wlo, whi);
      if (NCHW) {
        // channel loop outside: dy planes are contiguous along W; no prologue on the logits source
        for (int cc = 0; cc < C; ++cc) {
          const float* d = p.dy + ((long)n * C + cc) * ohw;
          float s = 0.f;
          for (int oh = hlo; oh <= hhi; ++oh) {
            float wh = tap_weight(oh, ih, sh, p.H);
            if (wh == 0.f) continue;
            float r = 0.f;
            for (int ow = wlo; ow <= whi; ++ow) {
              float ww = tap_weight(ow, iw, sw, p.W);
              if (ww != 0.f) r = fmaf(ww, d[(long)oh * p.OW + ow], r);
            }
            s = fmaf(wh, r, s);
          }
          s *= gs;
          float a = p.src.a ? p.src.a[cc] : 1.f;
          float* gp = p.g + pp * p.ldg + cc;
          float gv = s * a;
          *gp = p.accumulate ? *gp + gv : gv;
        }
      } else {
        float4 dz = zero4();
        // trim the conservative candidate ranges to the taps that really read this pixel (weights > 0 form one run)
        while (hlo < hhi && tap_weight(hlo, ih, sh, p.H) == 0.f) ++hlo;
        while (hhi > hlo && tap_weight(hhi, ih, sh, p.H) == 0.f) --hhi;
        while (wlo < whi && tap_weight(wlo, iw, sw, p.W) == 0.f) ++wlo;
        while (whi > wlo && tap_weight(whi, iw, sw, p.W) == 0.f) --whi;
        const int nh = hhi - hlo + 1, nw = whi - wlo + 1;
        if (nh <= 5 && nw <= 5 && p.vec && nrem >= 4) {
          // up to x2 up-sampling: <= 5x5 taps, every load of a row unconditional (clamped column, zero weight) and independent
          float ww5[5]; int wo5[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            ww5[k] = k < nw ? tap_weight(wlo + k, iw, sw, p.W) : 0.f;
            wo5[k] = (k < nw ? wlo + k : wlo) * p.lddy;
          }
          for (int a = 0; a < nh; ++a) {
            const float wh = tap_weight(hlo + a, ih, sh, p.H);
            const float* rowp = p.dy + (long)(n * p.OH + hlo + a) * p.OW * p.lddy + c;
            float4 d5[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) d5[b] = ld4(rowp + wo5[b]);
#pragma unroll
            for (int b = 0; b < 5; ++b) {
              const float k = wh * ww5[b];
              dz.x = fmaf(k, d5[b].x, dz.x); dz.y = fmaf(k, d5[b].y, dz.y); dz.z = fmaf(k, d5[b].z, dz.z); dz.w = fmaf(k, d5[b].w, dz.w);
            }
          }
        } else if (nh <= 8 && nw <= 8) {
          // per-axis weights in registers (static indexing), then at most 8x8 taps
          float wh8[8], ww8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            wh8[k] = k < nh ? tap_weight(hlo + k, ih, sh, p.H) : 0.f;
            ww8[k] = k < nw ? tap_weight(wlo + k, iw, sw, p.W) : 0.f;
          }
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            if (wh8[a] != 0.f) {
#pragma unroll
              for (int b = 0; b < 8; ++b) {
                if (ww8[b] != 0.f) {
                  float4 d = ld4g(p.dy + ((long)(n * p.OH + hlo + a) * p.OW + wlo + b) * p.lddy + c, nrem, p.vec);
                  float k = wh8[a] * ww8[b];
                  dz.x = fmaf(k, d.x, dz.x); dz.y = fmaf(k, d.y, dz.y); dz.z = fmaf(k, d.z, dz.z); dz.w = fmaf(k, d.w, dz.w);
                }
              }
            }
          }
        } else {
          for (int oh = hlo; oh <= hhi; ++oh) {
            float wh = tap_weight(oh, ih, sh, p.H);
            if (wh == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
              float ww = tap_weight(ow, iw, sw, p.W);
              if (ww == 0.f) continue;
              float4 d = ld4g(p.dy + ((long)(n * p.OH + oh) * p.OW + ow) * p.lddy + c, nrem, p.vec);
              float k = wh * ww;
              dz.x = fmaf(k, d.x, dz.x); dz.y = fmaf(k, d.y, dz.y); dz.z = fmaf(k, d.z, dz.z); dz.w = fmaf(k, d.w, dz.w);
            }
          }
        }
        dz.x *= gs; dz.y *= gs; dz.z *= gs; dz.w *= gs;
        if (p.src.relu || p.dab) {
          float4 x = ld4g(p.src.x + pp * p.src.ld + c, nrem, p.vec);
          if (p.src.relu) {
            if (!(fmaf(av.x, x.x, bv.x) > 0.f)) dz.x = 0.f;
            if (!(fmaf(av.y, x.y, bv.y) > 0.f)) dz.y = 0.f;
            if (!(fmaf(av.z, x.z, bv.z) > 0.f)) dz.z = 0.f;
            if (!(fmaf(av.w, x.w, bv.w) > 0.f)) dz.w = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { sA[e] += (double)get4(dz, e) * (double)get4(x, e); sB[e] += (double)get4(dz, e); }
        }
        float4 gv = make_float4(dz.x * av.x, dz.y * av.y, dz.z * av.z, dz.w * av.w);
        float* gp = p.g + pp * p.ldg + c;
        if (p.accumulate) { float4 o = ld4g(gp, nrem, p.vec); gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
        st4g(gp, gv, nrem, p.vec);
      }
    }
  }
  if (!NCHW && p.dab) {
    // [npl][C4][2] panel, one barrier, fixed-order column sums (see affine_sum_bwd_kernel)
    const int C4 = p.nq * 4;
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { redt[((pl * C4) + c + e) * 2] = sA[e]; redt[((pl * C4) + c + e) * 2 + 1] = sB[e]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * 2; k += 256) {
      const int ch = k >> 1, ab = k & 1;
      double acc = 0.0;
      for (int r = 0; r < p.npl; ++r) acc += redt[((r * C4) + ch) * 2 + ab];
      p.dab[(long)blockIdx.x * C * 2 + k] = acc;
    }
  }
}

// Table-driven backward for resizes of at most x2 up-sampling (every in-network resize of the path: /4, /2, x2 and the
// 63 <-> 64 / 125 <-> 128 fits of even-sized inputs).  The kernel above re-derives, in EVERY (pixel, channel quad) thread, the
// candidate output range of its input pixel and trims it with up to ten tap_weight evaluations per axis before it can issue a
// load: 28 us per launch at 1.1 TB/s.  Here a workgroup walks row segments of `tw` input pixels; per segment it builds the
// per-axis tables (first output index, tap count <= 5, weights — the very same fp32 expressions, so results are bit-identical)
// ONCE in LDS, tw + 1 threads busy for a few dozen cycles, and every thread then issues its nh*nw independent 16-byte loads
// straight from the table.  Statistics ((dA, dB) of a ReLU'd lazy source) as above: registers across segments, one row per workgroup.
constexpr int RT_MAXW = 128;
// MT: taps per axis an input pixel can receive from — 5 up to x2 up-sampling, 9 up to x4, 17 up to x8 (the resizes in front of ASPP)
template <int MT>
__device__ __forceinline__ void resize_bwd_tab_body(const RsK& p, int tiles_per_row, int ntiles, int tw, const int bx, const int gx) {      // workgroup bx of gx
  extern __shared__ double redt[];       // [npl][C4][2]
  __shared__ float wt_w[RT_MAXW][MT], wt_h[MT];
  __shared__ int lo_w[RT_MAXW], n_w[RT_MAXW], lo_h, n_h;
  const int C = p.src.C;
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q;
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  const float gs = p.dy_scale ? *p.dy_scale : 1.f;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  double sA[4] = {0.0, 0.0, 0.0, 0.0}, sB[4] = {0.0, 0.0, 0.0, 0.0};
  if (active && p.src.a) { av = ld4(p.src.a + c); bv = ld4(p.src.b + c); }
  for (int tile = bx; tile < ntiles; tile += gx) {
    const int rowid = tile / tiles_per_row, seg = tile - rowid * tiles_per_row;
    const int n = rowid / p.H, ih = rowid - n * p.H, iw0 = seg * tw;
    __syncthreads();                                   // the previous segment's readers are done with the tables
    if ((int)threadIdx.x <= tw) {
      const bool isw = (int)threadIdx.x < tw;
      const int i = isw ? iw0 + (int)threadIdx.x : ih, lim = isw ? p.W : p.H, in = lim, out = isw ? p.OW : p.OH;
      const float sc = isw ? sw : sh;
      int lo = 0, cnt = 0; float wv[MT];
#pragma unroll
      for (int k = 0; k < MT; ++k) wv[k] = 0.f;
      if (i < lim) {
        int hi; out_range(i, sc, out, lo, hi);
        while (lo < hi && tap_weight(lo, i, sc, in) == 0.f) ++lo;
        while (hi > lo && tap_weight(hi, i, sc, in) == 0.f) --hi;
        cnt = hi - lo + 1;
        if (cnt > MT) cnt = MT;                           // (cannot happen for the scales the host sends here)
#pragma unroll
        for (int k = 0; k < MT; ++k) wv[k] = k < cnt ? tap_weight(lo + k, i, sc, in) : 0.f;
      }
      if (isw) { lo_w[threadIdx.x] = lo; n_w[threadIdx.x] = cnt;
#pragma unroll
        for (int k = 0; k < MT; ++k) wt_w[threadIdx.x][k] = wv[k]; }
      else { lo_h = lo; n_h = cnt;
#pragma unroll
        for (int k = 0; k < MT; ++k) wt_h[k] = wv[k]; }
    }
    __syncthreads();
    if (!active) continue;
    const int nh = n_h, hlo = lo_h;
    for (int px = pl; px < tw; px += p.npl) {
      const int iw = iw0 + px;
      if (iw >= p.W) break;
      const int nw = n_w[px], wlo = lo_w[px];
      const long pp = (long)rowid * p.W + iw;
      float4 dz = zero4();
      float ww5[MT]; int wo5[MT];
#pragma unroll
      for (int k = 0; k < MT; ++k) { ww5[k] = wt_w[px][k]; wo5[k] = (k < nw ? wlo + k : wlo) * p.lddy; }
      for (int a = 0; a < nh; ++a) {
        const float wh = wt_h[a];
        const float* rowp = p.dy + (long)(n * p.OH + hlo + a) * p.OW * p.lddy + c;
        float4 d5[MT];
#pragma unroll
        for (int b = 0; b < MT; ++b) d5[b] = ld4(rowp + wo5[b]);
#pragma unroll
        for (int b = 0; b < MT; ++b) {
          const float k = wh * ww5[b];
          dz.x = fmaf(k, d5[b].x, dz.x); dz.y = fmaf(k, d5[b].y, dz.y); dz.z = fmaf(k, d5[b].z, dz.z); dz.w = fmaf(k, d5[b].w, dz.w);
        }
      }
      dz.x *= gs; dz.y *= gs; dz.z *= gs; dz.w *= gs;
      if (p.src.relu || p.dab) {
        const float4 x = ld4(p.src.x + pp * p.src.ld + c);
        if (p.src.relu) {
          if (!(fmaf(av.x, x.x, bv.x) > 0.f)) dz.x = 0.f;
          if (!(fmaf(av.y, x.y, bv.y) > 0.f)) dz.y = 0.f;
          if (!(fmaf(av.z, x.z, bv.z) > 0.f)) dz.z = 0.f;
          if (!(fmaf(av.w, x.w, bv.w) > 0.f)) dz.w = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sA[e] += (double)get4(dz, e) * (double)get4(x, e); sB[e] += (double)get4(dz, e); }
      }
      float4 gv = make_float4(dz.x * av.x, dz.y * av.y, dz.z * av.z, dz.w * av.w);
      float* gp = p.g + pp * p.ldg + c;
      if (p.accumulate) { const float4 o = ld4(gp); gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
      st4(gp, gv);
    }
  }
  if (p.dab) {
    const int C4 = p.nq * 4;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { redt[((pl * C4) + c + e) * 2] = sA[e]; redt[((pl * C4) + c + e) * 2 + 1] = sB[e]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C * 2; k += 256) {
      const int ch = k >> 1, ab = k & 1;
      double acc = 0.0;
      for (int r = 0; r < p.npl; ++r) acc += redt[((r * C4) + ch) * 2 + ab];
      p.dab[(long)bx * C * 2 + k] = acc;
    }
  }
}

template <int MT>
__global__ void __launch_bounds__(256) resize_bwd_tab_kernel(const RsK p, int tiles_per_row, int ntiles, int tw) {
  resize_bwd_tab_body<MT>(p, tiles_per_row, ntiles, tw, blockIdx.x, gridDim.x);
}
// [r4] the mutually independent resize backwards of one dependency level (a dense-in cell receives up to ten resized feature maps,
// ADD.py:84-90: ten gathers of 15-18 us, each little more than its launch floor) in ONE launch: block (x, z) = workgroup x of table entry z
struct RsTabItem { RsK p; int tpr, ntiles, tw, rows; };
template <int MT>
__global__ void __launch_bounds__(256) resize_bwd_tab_batch_kernel(const RsTabItem* __restrict__ tab) {
  const RsTabItem& e = tab[blockIdx.z];
  if ((int)blockIdx.x >= e.rows) return;
  resize_bwd_tab_body<MT>(e.p, e.tpr, e.ntiles, e.tw, blockIdx.x, e.rows);
}

int rs_rows(long P, int C) {
  EwMap m = ew_map(C);
  long r = P / ((long)m.npl * 2);
  if (r < 1) r = 1;
  if (r > 1024) r = 1024;
  return (int)r;
}

// table-driven backward: 0 = not covered, else the taps-per-axis variant (5 / 9 / 17); fills the launch geometry
int rs_tab_config(const addk_resize_bwd_args* a, RsK& k, RsTabItem& it, size_t& sh) {
  if (!a || a->nchw_in || !a->dy || !a->g || a->src.C <= 0 || a->src.C > 1024 || a->ldg < a->src.C || a->lddy < a->src.C) return 0;
  if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->OH <= 0 || a->OW <= 0) return 0;
  if ((a->src.relu || a->dab) && !(a->src.x && a->src.ld >= a->src.C)) return 0;
  if ((a->src.a == nullptr) != (a->src.b == nullptr)) return 0;
  k = RsK{};
  k.src = a->src; k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.dy = a->dy; k.lddy = a->lddy; k.dy_scale = a->dy_scale; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate; k.dab = (double*)a->dab;
  k.P = (long)a->N * a->H * a->W;
  EwMap m = ew_map(a->src.C); k.nq = m.nq; k.npl = m.npl;
  k.vec = aligned16(a->dy) && a->lddy % 4 == 0 && aligned16(a->g) && a->ldg % 4 == 0 && a->src.C % 4 == 0 && (!a->src.x || src_vec_ok(a->src));
  const bool up2 = a->OH <= 2 * a->H + 1 && a->OW <= 2 * a->W + 1, up4 = a->OH <= 4 * a->H + 1 && a->OW <= 4 * a->W + 1,
             up8 = a->OH <= 8 * a->H + 1 && a->OW <= 8 * a->W + 1;
  if (!(k.vec && a->src.C == m.nq * 4 && m.npl >= 2 && (up2 || up4 || up8) && (addk_get_fast_paths() & ADDK_FAST_DWTILE) && k.P < (1L << 30))) return 0;
  int tw = m.npl >= 4 ? m.npl * 4 : m.npl * 8; if (tw > RT_MAXW) tw = RT_MAXW; if (tw > a->W) tw = a->W;      // (2-3 pixel lanes: 260-512 channels)
  it.tpr = cdiv(a->W, tw); it.ntiles = a->N * a->H * it.tpr; it.tw = tw; it.rows = rs_rows(k.P, a->src.C);
  it.p = k;
  sh = (size_t)m.npl * m.nq * 4 * 2 * sizeof(double);
  return up2 ? 5 : up4 ? 9 : 17;
}

}  // namespace

// batched form: key >= 0 (the kernel variant) when the launch runs on the table-driven kernel and may share a batch
extern "C" int addk_resize_bwd_batch_key(const addk_resize_bwd_args* a) {
  RsK k; RsTabItem it; size_t sh;
  return rs_tab_config(a, k, it, sh) > 0 ? rs_tab_config(a, k, it, sh) : -1;
}
// host_blob = NULL: returns the blob size in bytes.  meta[0..3] = variant, n, grid x, dynamic LDS bytes
extern "C" int64_t addk_resize_bwd_batch_prepare(const addk_resize_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  if (!a || n <= 0 || !meta) { addk_set_error("resize_bwd_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  const int64_t total = (int64_t)n * sizeof(RsTabItem);
  if (host_blob && blob_bytes < total) { addk_set_error("resize_bwd_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  int key0 = -1, gx = 0; size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    RsK k; RsTabItem it; size_t sh;
    const int key = rs_tab_config(&a[i], k, it, sh);
    if (key <= 0) { addk_set_error("resize_bwd_batch_prepare: launch %d is not a table-driven shape", i); return ADDK_ERR_INVALID; }
    if (i == 0) key0 = key;
    if (key != key0) { addk_set_error("resize_bwd_batch_prepare: mixed kernel variants"); return ADDK_ERR_INVALID; }
    if (it.rows > gx) gx = it.rows;
    if (sh > lds) lds = sh;
    if (host_blob) reinterpret_cast<RsTabItem*>(host_blob)[i] = it;
  }
  meta[0] = key0; meta[1] = n; meta[2] = gx; meta[3] = (int64_t)lds;
  return total;
}
extern "C" int addk_resize_bwd_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[1] > 0 && meta[2] > 0, "resize_bwd_batch_run: bad args");
  const RsTabItem* tab = reinterpret_cast<const RsTabItem*>(dev_blob);
  const dim3 grid((unsigned)meta[2], 1, (unsigned)meta[1]);
  hipStream_t st = (hipStream_t)stream;
  if (meta[0] == 5) hipLaunchKernelGGL(resize_bwd_tab_batch_kernel<5>, grid, dim3(256), (size_t)meta[3], st, tab);
  else if (meta[0] == 9) hipLaunchKernelGGL(resize_bwd_tab_batch_kernel<9>, grid, dim3(256), (size_t)meta[3], st, tab);
  else if (meta[0] == 17) hipLaunchKernelGGL(resize_bwd_tab_batch_kernel<17>, grid, dim3(256), (size_t)meta[3], st, tab);
  else { addk_set_error("resize_bwd_batch_run: unknown variant %d", (int)meta[0]); return ADDK_ERR_INVALID; }
  return addk_check_launch("resize_bwd_batch");
}

extern "C" int addk_resize_fwd(const addk_resize_args* a, void* stream) {
  if (a && a->src.C > 1024 && !a->nchw_out) {      // wide concat buffers (F=40, level 3: 1600 channels): 1024-channel slices
    for (int c0 = 0; c0 < a->src.C; c0 += 1024) {
      addk_resize_args b = *a;
      b.src.x = a->src.x + c0; b.src.C = a->src.C - c0 < 1024 ? a->src.C - c0 : 1024;
      if (a->src.a) { b.src.a = a->src.a + c0; b.src.b = a->src.b + c0; }
      b.y = a->y + c0;
      int rc = addk_resize_fwd(&b, stream);
      if (rc) return rc;
    }
    return ADDK_OK;
  }
  ADDK_REQUIRE(a && a->src.x && a->y && a->src.C > 0 && a->src.C <= 1024 && a->src.ld >= a->src.C, "resize_fwd: bad args");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "resize_fwd: empty shape");
  ADDK_REQUIRE((a->src.a == nullptr) == (a->src.b == nullptr), "resize_fwd: a/b must come together");
  RsK k{};
  k.src = a->src; k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW; k.y = a->y; k.ldy = a->ldy; k.nchw = a->nchw_out;
  k.P = (long)a->N * a->OH * a->OW;
  hipStream_t st = (hipStream_t)stream;
  if (a->nchw_out) {
    long b = cdiv(k.P, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(resize_fwd_nchw_kernel, dim3((unsigned)b), dim3(256), 0, st, k);
  } else {
    ADDK_REQUIRE(a->ldy >= a->src.C, "resize_fwd: short ldy");
    EwMap m = ew_map(a->src.C); k.nq = m.nq; k.npl = m.npl;
    k.vec = src_vec_ok(a->src) && aligned16(a->y) && a->ldy % 4 == 0;
    long b = cdiv(k.P, (long)m.npl); if (b > 8192) b = 8192; if (b < 1) b = 1;
    hipLaunchKernelGGL(resize_fwd_kernel, dim3((unsigned)b), dim3(256), 0, st, k);
  }
  return addk_check_launch("resize_fwd");
}

extern "C" int addk_resize_bwd(const addk_resize_bwd_args* a, void* stream) {
  if (a && a->src.C > 1024 && !a->nchw_in && !a->dab) {
    for (int c0 = 0; c0 < a->src.C; c0 += 1024) {
      addk_resize_bwd_args b = *a;
      b.src.C = a->src.C - c0 < 1024 ? a->src.C - c0 : 1024;
      if (a->src.x) b.src.x = a->src.x + c0;
      if (a->src.a) { b.src.a = a->src.a + c0; b.src.b = a->src.b + c0; }
      b.dy = a->dy + c0; b.g = a->g + c0;
      int rc = addk_resize_bwd(&b, stream);
      if (rc) return rc;
    }
    return ADDK_OK;
  }
  ADDK_REQUIRE(a && a->dy && a->g && a->src.C > 0 && a->src.C <= 1024 && a->ldg >= a->src.C, "resize_bwd: bad args");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "resize_bwd: empty shape");
  ADDK_REQUIRE(!(a->src.relu || a->dab) || (a->src.x && a->src.ld >= a->src.C), "resize_bwd: prologue needs the forward input");
  ADDK_REQUIRE((a->src.a == nullptr) == (a->src.b == nullptr), "resize_bwd: a/b must come together");
  RsK k{};
  k.src = a->src; k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.dy = a->dy; k.lddy = a->lddy; k.dy_scale = a->dy_scale; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate; k.dab = (double*)a->dab;
  k.P = (long)a->N * a->H * a->W;
  hipStream_t st = (hipStream_t)stream;
  if (a->nchw_in) {
    ADDK_REQUIRE(!a->src.relu && !a->dab, "resize_bwd: NCHW gradient input has no prologue support");
    k.nq = 1; k.npl = 256;
    // receptive block of an 8x8 tile: (8 + 2) / scale + 4 output rows / columns (out_range's margins), at most RB_PM
    const float invh = (float)a->OH / (float)a->H, invw = (float)a->OW / (float)a->W;
    const bool fits = (RB_T + 2) * invh + 5.f <= (float)RB_PM && (RB_T + 2) * invw + 5.f <= (float)RB_PM &&
                      2.f * invh + 5.f <= (float)RB_MAXT && 2.f * invw + 5.f <= (float)RB_MAXT && invh >= 1.f && invw >= 1.f;
    if (fits && (addk_get_fast_paths() & ADDK_FAST_DWTILE)) {
      const int txs = cdiv(a->W, RB_T), tys = cdiv(a->H, RB_T);
      dim3 grid((unsigned)(a->N * tys * txs), (unsigned)a->src.C);
      hipLaunchKernelGGL(resize_bwd_nchw_tile_kernel, grid, dim3(256), 0, st, k, txs, tys);
      return addk_check_launch("resize_bwd_nchw_tile");
    }
    long b = cdiv(k.P, 256); if (b > 8192) b = 8192;
    hipLaunchKernelGGL(resize_bwd_kernel<true>, dim3((unsigned)b), dim3(256), 0, st, k);
  } else {
    ADDK_REQUIRE(a->lddy >= a->src.C, "resize_bwd: short lddy");
    EwMap m = ew_map(a->src.C); k.nq = m.nq; k.npl = m.npl;
    k.vec = aligned16(a->dy) && a->lddy % 4 == 0 && aligned16(a->g) && a->ldg % 4 == 0 && a->src.C % 4 == 0 &&
            (!a->src.x || src_vec_ok(a->src));
    int rows = rs_rows(k.P, a->src.C);
    size_t sh = (size_t)m.npl * m.nq * 4 * 2 * sizeof(double);
    // at most x2 up-sampling (<= 5 taps per axis), vector-aligned, a few pixel lanes per workgroup: the table-driven kernel
    const bool up2 = a->OH <= 2 * a->H + 1 && a->OW <= 2 * a->W + 1, up4 = a->OH <= 4 * a->H + 1 && a->OW <= 4 * a->W + 1,
               up8 = a->OH <= 8 * a->H + 1 && a->OW <= 8 * a->W + 1;
    if (k.vec && a->src.C == m.nq * 4 && m.npl >= 2 && (up2 || up4 || up8) && (addk_get_fast_paths() & ADDK_FAST_DWTILE) &&
        (long)a->N * a->H * a->W < (1L << 30)) {
      int tw = m.npl >= 4 ? m.npl * 4 : m.npl * 8; if (tw > RT_MAXW) tw = RT_MAXW; if (tw > a->W) tw = a->W;      // (2-3 pixel lanes: 260-512 channels)
      const int tpr = cdiv(a->W, tw), ntiles = a->N * a->H * tpr;
      if (up2) hipLaunchKernelGGL(resize_bwd_tab_kernel<5>, dim3(rows), dim3(256), sh, st, k, tpr, ntiles, tw);
      else if (up4) hipLaunchKernelGGL(resize_bwd_tab_kernel<9>, dim3(rows), dim3(256), sh, st, k, tpr, ntiles, tw);
      else hipLaunchKernelGGL(resize_bwd_tab_kernel<17>, dim3(rows), dim3(256), sh, st, k, tpr, ntiles, tw);
      return addk_check_launch("resize_bwd_tab");
    }
    hipLaunchKernelGGL(resize_bwd_kernel<false>, dim3(rows), dim3(256), sh, st, k);
  }
  return addk_check_launch("resize_bwd");
}
