// Pointwise (1x1, stride 1) convolution with register-stationary weights — the pointwise halves of SepConv and the
// 40/80/160-channel glue convs, forward and data gradient.  These launches are far too small to amortise the LDS
// staging + two barriers per 32-channel chunk of the general implicit-GEMM kernel (measured: 25-30 us each against an
// HBM-ideal 5-10 us).  Here there is NO LDS and NO barrier in the main loop:
//   * a wave keeps its whole weight panel as MFMA A-fragments in registers (K <= 160, <= 3 column tiles of 16),
//   * it walks 16-pixel tiles; the MFMA B-fragment of a tile is loaded straight from global memory, 16 B per lane
//     (lane (pixel li, quarter kq) holds k = 16g + 4kq + {0..3}; the k order inside a 16-channel group is a permutation
//     that weights and pixels share, so the sum is unchanged), two tiles in flight per wave,
//   * the lazy BatchNorm/ReLU prologue is applied to the fragment in registers,
//   * the epilogue stores 16 B per lane and keeps the BN statistics / (dA,dB) sums in per-lane fp64 registers for the
//     whole kernel; one butterfly + cross-wave LDS reduction at the very end writes the block's slab row.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

enum { PW_FWD = 0, PW_DGRAD = 1 };

struct PwK {
  addk_src src;            // fwd: input (x,a,b,relu); dgrad: dy (no prologue)
  int K;                   // reduction length (fwd: Cin, dgrad: Cout_fwd)
  int Cn;                  // GEMM N (fwd: Cout, dgrad: channels of dst)
  const float* w; int ldw; int w_off;     // fwd: w[(n)*ldw + w_off + k]; dgrad: w[(k)*ldw + w_off + n]
  float* y; int ldy;
  const float* bias;
  double* slab; int slab_ld;
  addk_src dst; int accumulate;          // dgrad epilogue
  int P; int ntiles16; int rows;       // rows: slab rows the caller allocated (>= gx; the extra rows are zero-filled)
  int gx, gy;                          // grid of this launch (a batched launch runs several descriptors on one larger grid)
  // fused SepConv half (SEP = 3 / 5): depthwise K x K in front of the pointwise conv, computed while the B fragment is built
  const float* dww; int H, W;          // depthwise weights [C][K*K]; image geometry (stride 1, dilation 1, pad K/2)
  float* t; int ldt;                   // optional copy of the depthwise output (training: the backward pass reads it)
  const float* ea; const float* eb;    // inference epilogue: y = ea*acc + eb + sum of terms
  int nterm; addk_src term[ADDK_MAX_TERMS];
  // RS (forward only): src.x is an [N, RH, RW] map sampled bilinearly onto this launch's [N, H, W] pixel grid (addk_src.rs_hw);
  // rs_y: optional materialised copy of the interpolated input (training: the backward pass reads it)
  int RH, RW; float* rs_y; int rs_ldy;
};

// These launches are latency chains (kernel arguments -> weight panel -> one or two pixel tiles -> store -> statistics)
// run by a few thousand waves: what sets their duration is how many of those chains are resident at once, so the kernel
// is kept under 128 (KG=3) / 168 (KG=5) registers for 4 / 3 waves per SIMD: no prefetch buffer (occupancy hides the
// latency) and, for maps of >= 4096 pixels, per-lane statistics in fp32 (a lane sums at most a handful of values; the
// cross-lane / cross-wave / cross-block sums stay fp64).
// SEP > 0 (forward only): the B operand is not x but depthwise_SEPxSEP(relu?(a*x+b)) — every lane builds its fragment
// (pixel li, channels 16g + 4kq + {0..3}) from the SEP*SEP neighbouring pixels, all loads unconditional and independent (L1/L2 hits:
// neighbouring lanes share them), zero padding after the prologue as in the reference; tap weights sit in LDS as [tap][channel].
// No barrier in the main loop and no round trip of the depthwise output through HBM (operations.py:51-53: ReLU, dw, pw in one launch).
// RS: the B operand is the bilinear interpolation (align_corners = False) of a map of another size, taken while the fragment is loaded —
// four 16-byte loads per channel group instead of one (neighbouring lanes share them: L1 / L2 hits), the resize.hip expressions
// (common.h: src_index, lerp4) in the same order, so the result is bit-identical to addk_resize_fwd followed by the plain launch
// (ADD.py:76-77,84-90: F.interpolate in front of `preprocess` / `pre_preprocess`; the resized tensor is never written at inference).
template <int CT, int KG, int MODE, bool RED32, int SEP = 0, bool RS = false>     // CT column tiles of 16, KG groups of 16 reduction channels
__device__ __forceinline__ void pw_body(const PwK& p, double (*red)[CT * 16][2], const float* dwl = nullptr, float* patch = nullptr, const float* abl = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.y * (CT * 16);

  // ---- weight panel -> registers (A operand: row = output channel li of tile i, k = 16g + 4kq + e) ----
  // Every load below is unconditional (masked lanes read a safe address and are zeroed afterwards) so that the whole
  // panel, the prologue coefficients and the first pixel tile are ONE round trip to memory, not a chain of them.
  float4 wf[CT][KG];
  auto load_panel = [&]() {
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const int n = n0 + i * 16 + li, k = 16 * g + 4 * kq;
        const bool ok = n < p.Cn && k < p.K;          // K % 4 == 0: a quad is valid as a whole
        float4 v;
        if (MODE == PW_FWD) {
          v = ld4(ok ? p.w + (long)n * p.ldw + p.w_off + k : p.w);
        } else {
          const gfloat* b = (const gfloat*)(ok ? p.w + (long)k * p.ldw + p.w_off + n : p.w);
          const long st = ok ? p.ldw : 0;
          v = make_float4(b[0], b[st], b[2 * st], b[3 * st]);
        }
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        wf[i][g] = v;
      }
  };
  if (!SEP) load_panel();        // fused form: the panel is fetched after the depthwise phase (its registers are needed there)
  // lazy prologue coefficients of this lane's k slots
  const bool relu = p.src.relu != 0;
  const bool pro = relu || p.src.a != nullptr;
  float4 pa[KG], pb[KG];
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    const int k = 16 * g + 4 * kq;
    pa[g] = make_float4(1.f, 1.f, 1.f, 1.f); pb[g] = zero4();
    if (p.src.a) {                                  // wave-uniform
      const bool ok = k < p.K;
      const float4 av = ld4(p.src.a + (ok ? k : 0)), bv = ld4(p.src.b + (ok ? k : 0));
      pa[g].x = ok ? av.x : 1.f; pa[g].y = ok ? av.y : 1.f; pa[g].z = ok ? av.z : 1.f; pa[g].w = ok ? av.w : 1.f;
      pb[g].x = ok ? bv.x : 0.f; pb[g].y = ok ? bv.y : 0.f; pb[g].z = ok ? bv.z : 0.f; pb[g].w = ok ? bv.w : 0.f;
    }
  }

  typedef typename std::conditional<RED32, float, double>::type red_t;
  red_t s1[CT][4], s2[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[i][e] = 0; s2[i][e] = 0; }

  const int wstride = p.gx * 4;
  float4 xf[KG];
  const float rs_sh = RS ? (float)p.RH / (float)p.H : 0.f, rs_sw = RS ? (float)p.RW / (float)p.W : 0.f;
  auto load_tile = [&](int tile, float4 (&x)[KG]) {
    const int pp = tile * 16 + li;
    if (RS) {
      const int hw = p.H * p.W, pc = pp < p.P ? pp : 0;
      const int n = pc / hw, rem = pc - n * hw, oh = rem / p.W, ow = rem - oh * p.W;
      int h0, h1, w0, w1; float lh0, lh1, lw0, lw1;
      src_index(oh, rs_sh, p.RH, h0, h1, lh0, lh1);
      src_index(ow, rs_sw, p.RW, w0, w1, lw0, lw1);
      const float* b = p.src.x + (long)n * p.RH * p.RW * p.src.ld;
      const long o00 = ((long)h0 * p.RW + w0) * p.src.ld, o01 = ((long)h0 * p.RW + w1) * p.src.ld;
      const long o10 = ((long)h1 * p.RW + w0) * p.src.ld, o11 = ((long)h1 * p.RW + w1) * p.src.ld;
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const int k = 16 * g + 4 * kq;
        const bool ok = pp < p.P && k < p.K;
        const float* bk = ok ? b + k : p.src.x;
        const float4 v00 = ld4(bk + (ok ? o00 : 0)), v01 = ld4(bk + (ok ? o01 : 0)), v10 = ld4(bk + (ok ? o10 : 0)), v11 = ld4(bk + (ok ? o11 : 0));
        float4 v = lerp4(v00, v01, v10, v11, lh0, lh1, lw0, lw1);
        if (p.rs_y && ok && blockIdx.y == 0) st4(p.rs_y + (long)pp * p.rs_ldy + k, v);
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        x[g] = v;
      }
      return;
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int k = 16 * g + 4 * kq;
      const bool ok = pp < p.P && k < p.K;
      float4 v = ld4(ok ? p.src.x + (long)pp * p.src.ld + k : p.src.x);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      x[g] = v;
    }
  };
  // Fused SepConv half, LDS-tiled: the block owns 64 consecutive pixels of one image row; it stages the haloed input patch
  // [SEP rows][64 + SEP - 1 px][channels] once (all loads independent, BatchNorm/ReLU applied on the way in, zero padding
  // after it), and every lane then builds its B fragment — the depthwise output of (pixel 16 wave + li, channels 16g + 4kq ..)
  // — from SEP*SEP 16-byte LDS reads per group.  The pixel stride KP = 16 KG + 4 floats makes the 16 pixel lanes of a
  // fragment read hit 16 distinct 4-bank groups (52 and 84 are = 4 * odd mod 64).
  constexpr int SEPX = SEP ? SEP : 1, PWX = 64 + SEPX - 1, KP = KG * 16 + 4, KQ = KG * 4;
  constexpr int NSLOT = (SEPX * PWX * KQ + 255) / 256;
  auto stage_patch = [&](int n, int oh, int ow0) {
    constexpr int HK = SEPX / 2;
    constexpr int NB = 8;                                               // independent loads per thread in flight
#pragma unroll 1
    for (int b0 = 0; b0 < NSLOT; b0 += NB) {
      float4 rv[NB];
      unsigned okm = 0;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int slot = t + 256 * (b0 + u);
        const int r = slot / (PWX * KQ), rem = slot - r * (PWX * KQ), px = rem / KQ, q = rem - px * KQ;
        const int ih = oh - HK + r, iw = ow0 - HK + px;
        const bool ok = b0 + u < NSLOT && r < SEPX && 4 * q < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        okm |= (ok ? 1u : 0u) << u;
        rv[u] = ld4(p.src.x + (ok ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + 4 * q : 0));
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int slot = t + 256 * (b0 + u);
        const int r = slot / (PWX * KQ), rem = slot - r * (PWX * KQ), px = rem / KQ, q = rem - px * KQ;
        float4 v = rv[u];
        if (pro) {
          const float4 av = lds_ld4(abl + 4 * q), bv = lds_ld4(abl + KG * 16 + 4 * q);
          v.x = fmaf(av.x, v.x, bv.x); v.y = fmaf(av.y, v.y, bv.y); v.z = fmaf(av.z, v.z, bv.z); v.w = fmaf(av.w, v.w, bv.w);
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        const bool ok = (okm >> u) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        if (b0 + u < NSLOT && r < SEPX) lds_st4(patch + (r * PWX + px) * KP + 4 * q, v);
      }
    }
  };
  auto dw_from_patch = [&](float4 (&x)[KG]) {
    const float* pb0 = patch + (16 * wave + li) * KP + 4 * kq;
#pragma unroll
    for (int g = 0; g < KG; ++g) x[g] = zero4();
#pragma unroll
    for (int kh = 0; kh < SEPX; ++kh)
#pragma unroll
      for (int kw = 0; kw < SEPX; ++kw)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          const float4 v = lds_ld4(pb0 + (kh * PWX + kw) * KP + 16 * g);
          const float4 w4 = lds_ld4(dwl + (kh * SEPX + kw) * (KG * 16) + 16 * g + 4 * kq);
          x[g].x = fmaf(w4.x, v.x, x[g].x); x[g].y = fmaf(w4.y, v.y, x[g].y); x[g].z = fmaf(w4.z, v.z, x[g].z); x[g].w = fmaf(w4.w, v.w, x[g].w);
        }
  };
  const int spr = SEP ? (p.W + 63) / 64 : 1;
  for (int tile = SEP ? (int)blockIdx.x : blockIdx.x * 4 + wave; tile < p.ntiles16; tile += SEP ? p.gx : wstride) {
    int pp; bool pin;
    if (SEP) {            // tile = row segment (n, oh, 64 px): block-wide staging, then per-wave fragments
      const int rowid = tile / spr, sx = tile - rowid * spr;
      const int n = rowid / p.H, oh = rowid - n * p.H, ow = sx * 64 + 16 * wave + li;
      if (tile != (int)blockIdx.x) __syncthreads();          // every wave is done with the previous segment's patch
      if (KG <= 3 && tile == (int)blockIdx.x) load_panel();  // in the same round trip as the patch
      stage_patch(n, oh, sx * 64);
      __syncthreads();
      dw_from_patch(xf);
      pin = ow < p.W;
      pp = rowid * p.W + ow;
      if (p.t && pin) {
#pragma unroll
        for (int g = 0; g < KG; ++g) { const int k = 16 * g + 4 * kq; if (k < p.K) st4(p.t + (long)pp * p.ldt + k, xf[g]); }
      }
    } else {
      load_tile(tile, xf);
      pp = tile * 16 + li;
      pin = pp < p.P;
    }
    // [r4] data gradient: the epilogue's operands (the forward input for the ReLU mask and the (dA, dB) sums, the gradient to
    // accumulate into) do not depend on the matrix product — they are requested with the tile, not after it: one dependent
    // round trip per tile instead of two or three (these launches are latency chains: 2 TB/s of their algorithmic bytes before)
    float4 dxp[CT], dop[CT];
    if (MODE == PW_DGRAD) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int c = n0 + i * 16 + kq * 4;
        const int nrem = pin ? p.Cn - c : 0;
        dxp[i] = ld4g(p.dst.x + (pin ? (long)pp * p.dst.ld + c : 0), nrem, true);
        dop[i] = zero4();
        if (p.accumulate) dop[i] = ld4g(p.y + (pin ? (long)pp * p.ldy + c : 0), nrem, true);
      }
    }
    if (pro && !SEP) {
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        float4 v = xf[g];
        v.x = fmaf(pa[g].x, v.x, pb[g].x); v.y = fmaf(pa[g].y, v.y, pb[g].y); v.z = fmaf(pa[g].z, v.z, pb[g].z); v.w = fmaf(pa[g].w, v.w, pb[g].w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = pp < p.P && 16 * g + 4 * kq < p.K;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        xf[g] = v;
      }
    }
    f32x4 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (SEP && KG > 3) {          // wide fused form: the weight fragments of one group at a time (L1 hits), not the whole 25-quad panel
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int n = n0 + i * 16 + li, k = 16 * g + 4 * kq;
          const bool ok = n < p.Cn && k < p.K;
          float4 v = ld4(ok ? p.w + (long)n * p.ldw + p.w_off + k : p.w);
          v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
          wf[i][0] = v;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < CT; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(wf[i][(SEP && KG > 3) ? 0 : g], e), get4(xf[g], e), acc[i], 0, 0, 0);
    }
    // ---- epilogue: lane holds channels n0 + i*16 + kq*4 + {0..3} of pixel pp ----
    if (pin) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int c = n0 + i * 16 + kq * 4;
        const int nrem = p.Cn - c;
        if (nrem <= 0) continue;
        float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        if (MODE == PW_FWD) {
          if (p.bias) { float4 b = ld4g(p.bias + c, nrem, false); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
          if (SEP) {            // inference epilogue: own frozen BatchNorm, then the other branches of the cell block
            if (p.ea) {
              const float4 ea = ld4g(p.ea + c, nrem, true), eb = ld4g(p.eb + c, nrem, true);
              v.x = fmaf(ea.x, v.x, eb.x); v.y = fmaf(ea.y, v.y, eb.y); v.z = fmaf(ea.z, v.z, eb.z); v.w = fmaf(ea.w, v.w, eb.w);
            }
            for (int ti = 0; ti < p.nterm; ++ti) {
              const addk_src& T = p.term[ti];
              const float4 u = prologue4(ld4g(T.x + (long)pp * T.ld + c, nrem, true), T.a, T.b, c, nrem, T.relu != 0, true);
              v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
          }
          st4g(p.y + (long)pp * p.ldy + c, v, nrem, true);
          if (p.slab) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const red_t f = e < nrem ? (red_t)get4(v, e) : (red_t)0; s1[i][e] += f; s2[i][e] += f * f; }
          }
        } else {
          const float4 x = dxp[i];
          float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
          if (p.dst.a) { av = ld4g(p.dst.a + c, nrem, true); bv = ld4g(p.dst.b + c, nrem, true); }      // (C-length vectors: L1 hits)
          float4 g4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = get4(x, e), ae = get4(av, e), be = get4(bv, e), dz = get4(v, e);
            const bool m = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
            set4(g4, e, m ? dz * ae : 0.f);
            if (p.slab && m) { s1[i][e] += (red_t)dz * (red_t)xe; s2[i][e] += (red_t)dz; }
          }
          float* gp = p.y + (long)pp * p.ldy + c;
          if (p.accumulate) { const float4 o = dop[i]; g4.x += o.x; g4.y += o.y; g4.z += o.z; g4.w += o.w; }
          st4g(gp, g4, nrem, true);
        }
      }
    }
  }

  if (p.slab) {     // once per kernel: butterfly over the 16 pixel lanes, then the four waves through LDS (fixed order)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red_t a = s1[i][e], b = s2[i][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        if (li == 0) { red[wave][i * 16 + kq * 4 + e][0] = (double)a; red[wave][i * 16 + kq * 4 + e][1] = (double)b; }
      }
    __syncthreads();
    if (t < CT * 16 && n0 + t < p.Cn) {
      gdouble* o = (gdouble*)p.slab + ((long)blockIdx.x * p.slab_ld + n0 + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
      for (int r = blockIdx.x + p.gx; r < p.rows; r += p.gx) {     // rows no workgroup owns
        gdouble* z = (gdouble*)p.slab + ((long)r * p.slab_ld + n0 + t) * 2;
        z[0] = 0.0; z[1] = 0.0;
      }
    }
  }
}

template <int CT, int KG, int MODE, bool RED32, bool RS = false>
__global__ void __launch_bounds__(256, (KG <= 3 && !RS && MODE == PW_FWD ? 4 : 3)) pw_kernel(const PwK p) {
  __shared__ double red[4][CT * 16][2];
  pw_body<CT, KG, MODE, RED32, 0, RS>(p, red);
}
// several independent pointwise convs of one dependency level in ONE launch: block (x, y, z) runs descriptor z
template <int CT, int KG, int MODE, bool RED32, bool RS = false>
__global__ void __launch_bounds__(256, (KG <= 3 && !RS && MODE == PW_FWD ? 4 : 3)) pw_batch_kernel(const PwK* __restrict__ tab) {
  __shared__ double red[4][CT * 16][2];
  const PwK p = tab[blockIdx.z];
  if ((int)blockIdx.x >= p.gx || (int)blockIdx.y >= p.gy) return;
  pw_body<CT, KG, MODE, RED32, 0, RS>(p, red);
}

// Streaming-K pointwise forward for the 1x1 convs the register-stationary kernel does not take: many input channels
// (the dense-connection preprocess convs, K = 200..800) and virtual concatenations (up to 12 sources).  Same lane
// layout and arithmetic as pw_kernel, but the weight fragments are streamed per 16-channel group (every wave reads the
// same panel: L1/L2 hits) instead of being held for the whole kernel, and the loads of group g+1 are in flight while the
// 4*CT MFMAs of group g issue — no LDS, no barrier.  On the generic implicit-GEMM kernel these launches paid two
// barriers and an exposed global-load round trip per 32 input channels for 16 MFMAs per wave (8 TF/s).
struct PwkK {
  addk_src src[ADDK_MAX_SRC]; int nsrc;
  int Cn; const float* w; int ldw; int w_off;
  float* y; int ldy; const float* bias;
  double* slab; int slab_ld;
  int P, ntiles16, rows, gx;
  int H, W; float* rs_y; int rs_ldy;       // RS: this launch's pixel grid; sources with rs_hw != 0 are sampled onto it (pw_body's RS form)
};

template <int CT, bool RED32, bool RS = false>
__global__ void __launch_bounds__(256, RS ? 2 : 3) pwk_kernel(const PwkK p) {
  __shared__ double red[4][CT * 16][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.y * (CT * 16);
  typedef typename std::conditional<RED32, float, double>::type red_t;
  red_t s1[CT][4], s2[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[i][e] = 0; s2[i][e] = 0; }
  bool nok[CT]; const float* wrow[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int n = n0 + i * 16 + li;
    nok[i] = n < p.Cn;
    wrow[i] = p.w + (long)(nok[i] ? n : 0) * p.ldw + p.w_off;
  }
  const int wstride = p.gx * 4;
  for (int tile = blockIdx.x * 4 + wave; tile < p.ntiles16; tile += wstride) {
    const int pp = tile * 16 + li;
    const bool pok = pp < p.P;
    f32x4 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one group = 16 input channels of one source: x quad, prologue coefficients, CT weight quads
    struct Grp { float4 x, x01, x10, x11, a, b, w[CT]; float lh0, lh1, lw0, lw1; bool ok, pro, relu, rs, copy; int k; };
    int s = 0, g = 0, choff = 0;
    int rn = 0, roh = 0, row = 0;
    if (RS) {
      const int hw = p.H * p.W, pc = pok ? pp : 0;
      rn = pc / hw; const int rem = pc - rn * hw; roh = rem / p.W; row = rem - roh * p.W;
    }
    auto load = [&](Grp& G, int s_, int g_, int choff_) {
      const addk_src S = p.src[s_];
      const int k = 16 * g_ + 4 * kq;
      const bool kok = k < S.C;
      G.ok = kok && pok; G.pro = S.a != nullptr; G.relu = S.relu != 0;
      G.rs = RS && S.rs_hw != 0;
      if (G.rs) {             // wave-uniform: the four taps of the interpolation, combined in compute() (the loads stay in flight)
        const int RH = S.rs_hw >> 16, RW = S.rs_hw & 0xffff;
        int h0, h1, w0, w1;
        src_index(roh, (float)RH / (float)p.H, RH, h0, h1, G.lh0, G.lh1);
        src_index(row, (float)RW / (float)p.W, RW, w0, w1, G.lw0, G.lw1);
        const float* b = S.x + (G.ok ? (long)rn * RH * RW * S.ld + k : 0);
        G.x = ld4(b + (G.ok ? ((long)h0 * RW + w0) * S.ld : 0));
        G.x01 = ld4(b + (G.ok ? ((long)h0 * RW + w1) * S.ld : 0));
        G.x10 = ld4(b + (G.ok ? ((long)h1 * RW + w0) * S.ld : 0));
        G.x11 = ld4(b + (G.ok ? ((long)h1 * RW + w1) * S.ld : 0));
        G.copy = s_ == 0 && p.rs_y != nullptr && blockIdx.y == 0; G.k = k;
      } else
      G.x = ld4(S.x + (G.ok ? (long)pp * S.ld + k : 0));
      G.a = make_float4(1.f, 1.f, 1.f, 1.f); G.b = zero4();
      if (S.a) { G.a = ld4(S.a + (kok ? k : 0)); G.b = ld4(S.b + (kok ? k : 0)); }
#pragma unroll
      for (int i = 0; i < CT; ++i) G.w[i] = ld4(wrow[i] + (kok ? choff_ + k : 0));
    };
    auto advance = [&](int& s_, int& g_, int& choff_) {
      ++g_;
      if (16 * g_ >= p.src[s_].C) { choff_ += p.src[s_].C; g_ = 0; ++s_; }
    };
    auto compute = [&](const Grp& G) {
      float4 v = G.x;
      if (RS && G.rs) {
        v = lerp4(G.x, G.x01, G.x10, G.x11, G.lh0, G.lh1, G.lw0, G.lw1);
        if (G.copy && G.ok) st4(p.rs_y + (long)pp * p.rs_ldy + G.k, v);
      }
      v.x = fmaf(G.a.x, v.x, G.b.x); v.y = fmaf(G.a.y, v.y, G.b.y); v.z = fmaf(G.a.z, v.z, G.b.z); v.w = fmaf(G.a.w, v.w, G.b.w);
      if (G.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      v.x = G.ok ? v.x : 0.f; v.y = G.ok ? v.y : 0.f; v.z = G.ok ? v.z : 0.f; v.w = G.ok ? v.w : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          // masked k slots carry x = 0, rows beyond Cout are never stored: the weight operand needs no masking
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(G.w[i], e), get4(v, e), acc[i], 0, 0, 0);
        }
    };
    Grp A, B;
    load(A, s, g, choff);
    while (true) {
      int s2_ = s, g2 = g, c2 = choff;
      advance(s2_, g2, c2);
      const bool more = s2_ < p.nsrc;
      if (more) load(B, s2_, g2, c2);
      compute(A);
      if (!more) break;
      s = s2_; g = g2; choff = c2;
      int s3 = s, g3 = g, c3 = choff;
      advance(s3, g3, c3);
      const bool more2 = s3 < p.nsrc;
      if (more2) load(A, s3, g3, c3);
      compute(B);
      if (!more2) break;
      s = s3; g = g3; choff = c3;
    }
    if (pok) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int c = n0 + i * 16 + kq * 4;
        const int nrem = p.Cn - c;
        if (nrem <= 0) continue;
        float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        if (p.bias) { float4 b = ld4g(p.bias + c, nrem, false); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        st4g(p.y + (long)pp * p.ldy + c, v, nrem, true);
        if (p.slab) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const red_t f = e < nrem ? (red_t)get4(v, e) : (red_t)0; s1[i][e] += f; s2[i][e] += f * f; }
        }
      }
    }
  }
  if (p.slab) {
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red_t a = s1[i][e], b = s2[i][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        if (li == 0) { red[wave][i * 16 + kq * 4 + e][0] = (double)a; red[wave][i * 16 + kq * 4 + e][1] = (double)b; }
      }
    __syncthreads();
    if (t < CT * 16 && n0 + t < p.Cn) {
      gdouble* o = (gdouble*)p.slab + ((long)blockIdx.x * p.slab_ld + n0 + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
      for (int r = blockIdx.x + p.gx; r < p.rows; r += p.gx) {
        gdouble* z = (gdouble*)p.slab + ((long)r * p.slab_ld + n0 + t) * 2;
        z[0] = 0.0; z[1] = 0.0;
      }
    }
  }
}

struct PwCfg { int ct, kg, red32, gx, gy, rs; };
static bool pw_config(PwK& k, int rows, PwCfg& c) {
  c.rs = k.RH != 0;
  c.kg = cdiv(k.K, 16);
  if (!(c.kg == 3 || c.kg == 5)) return false;
  c.ct = c.kg == 5 ? 2 : 3;
  if (cdiv(k.Cn, 16) < c.ct) c.ct = cdiv(k.Cn, 16);
  if (c.ct < 1) c.ct = 1;
  k.rows = rows;
  c.gx = rows;
  // [r4] two 16-pixel tiles per wave: a wave's fixed cost (the weight panel — 36 strided scalar loads per lane in the data gradient —, the
  // prologue coefficients, the statistics reduction) was paid for ONE tile; with two the level-batched launches (up to ~9000 workgroups
  // queueing on 256 CUs) do half of it: step 35.0 -> 34.5-34.7 ms, inference segment unchanged (ADDK_PW_TILES=1 restores one tile, 4 measured 34.65)
  const int tpw = 2;
  if (c.gx > cdiv(k.ntiles16, 4 * tpw)) c.gx = cdiv(k.ntiles16, 4 * tpw);
  if (c.gx < 1) c.gx = 1;
  c.gy = cdiv(k.Cn, 16 * c.ct);
  c.red32 = k.P >= 4096;
  k.gx = c.gx; k.gy = c.gy;
  return true;
}

template <int MODE>
int pw_launch(PwK& k, int rows, hipStream_t st) {
  PwCfg c;
  if (!pw_config(k, rows, c)) return 1;                  // no instantiation: caller falls back to the general kernel
  dim3 grid(c.gx, c.gy);          // (round 1 measured far fewer, fatter workgroups slower, 14 -> 36 us: the grid keeps >= ~500 workgroups on the large maps)
#define ADDK_PW(CT_, KG_) \
  if (c.ct == CT_ && c.kg == KG_) { \
    if (MODE == PW_FWD && c.rs) { \
      if (c.red32) hipLaunchKernelGGL((pw_kernel<CT_, KG_, PW_FWD, true, true>), grid, dim3(256), 0, st, k); \
      else hipLaunchKernelGGL((pw_kernel<CT_, KG_, PW_FWD, false, true>), grid, dim3(256), 0, st, k); \
    } else if (c.red32) hipLaunchKernelGGL((pw_kernel<CT_, KG_, MODE, true>), grid, dim3(256), 0, st, k); \
    else hipLaunchKernelGGL((pw_kernel<CT_, KG_, MODE, false>), grid, dim3(256), 0, st, k); \
    return addk_check_launch("pw_conv"); }
  ADDK_PW(1, 3) ADDK_PW(2, 3) ADDK_PW(3, 3)
  ADDK_PW(1, 5) ADDK_PW(2, 5)
#undef ADDK_PW
  return 1;
}

template <int MODE>
int pw_batch_launch(const PwK* tab, int n, int ct, int kg, int red32, int rs, int gx, int gy, hipStream_t st) {
  dim3 grid(gx, gy, n);
#define ADDK_PW(CT_, KG_) \
  if (ct == CT_ && kg == KG_) { \
    if (MODE == PW_FWD && rs) { \
      if (red32) hipLaunchKernelGGL((pw_batch_kernel<CT_, KG_, PW_FWD, true, true>), grid, dim3(256), 0, st, tab); \
      else hipLaunchKernelGGL((pw_batch_kernel<CT_, KG_, PW_FWD, false, true>), grid, dim3(256), 0, st, tab); \
    } else if (red32) hipLaunchKernelGGL((pw_batch_kernel<CT_, KG_, MODE, true>), grid, dim3(256), 0, st, tab); \
    else hipLaunchKernelGGL((pw_batch_kernel<CT_, KG_, MODE, false>), grid, dim3(256), 0, st, tab); \
    return addk_check_launch("pw_conv_batch"); }
  ADDK_PW(1, 3) ADDK_PW(2, 3) ADDK_PW(3, 3)
  ADDK_PW(1, 5) ADDK_PW(2, 5)
#undef ADDK_PW
  addk_set_error("pw_batch: no instantiation");
  return ADDK_ERR_UNSUPPORTED;
}

bool pw_fill_fwd(const addk_conv_args* a, PwK& k) {
  if (a->nsrc != 1 || a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->bias_n) return false;
  const addk_src& s = a->src[0];
  const int kg = cdiv(s.C, 16);
  if (!(kg == 3 || kg == 5) || !src_vec_ok(s) || !aligned16(a->y) || a->ldy % 4 || a->Cout % 4 || a->H != a->OH || a->W != a->OW) return false;
  if (!aligned16(a->w) || a->ldw % 4 || a->w_choff % 4) return false;
  k = PwK{};
  k.src = s; k.K = s.C; k.Cn = a->Cout; k.w = a->w; k.ldw = a->ldw; k.w_off = a->w_choff;
  k.y = a->y; k.ldy = a->ldy; k.bias = a->bias;
  k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.P = a->N * a->OH * a->OW; k.ntiles16 = cdiv(k.P, 16);
  k.H = a->OH; k.W = a->OW;
  if (s.rs_hw) {                      // the input is sampled from an [N, RH, RW] map (addk_src.rs_hw)
    k.RH = s.rs_hw >> 16; k.RW = s.rs_hw & 0xffff;
    if (k.RH < 1 || k.RW < 1) return false;
    k.rs_y = a->rs_y; k.rs_ldy = a->rs_ldy;
    if (k.rs_y && (!aligned16(k.rs_y) || k.rs_ldy % 4 || k.rs_ldy < s.C)) return false;
  }
  return true;
}
bool pw_fill_dgrad(const addk_conv_dgrad_args* a, PwK& k) {
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->H != a->OH || a->W != a->OW) return false;
  const int kg = cdiv(a->Cout, 16);
  addk_src dy{a->dy, nullptr, nullptr, a->lddy, a->Cout, 0, 0};
  if (!(kg == 3 || kg == 5) || !src_vec_ok(dy) || !src_vec_ok(a->dst) || !aligned16(a->g) || a->ldg % 4) return false;
  k = PwK{};
  k.src = dy; k.K = a->Cout; k.Cn = a->dst.C; k.w = a->w; k.ldw = a->ldw; k.w_off = a->w_choff;
  k.y = a->g; k.ldy = a->ldg; k.slab = (double*)a->dab; k.slab_ld = a->dst.C;
  k.dst = a->dst; k.accumulate = a->accumulate;
  k.P = a->N * a->H * a->W; k.ntiles16 = cdiv(k.P, 16);
  return true;
}
inline int pw_key(const PwCfg& c, int mode) { return (mode << 12) | (c.ct << 8) | (c.kg << 4) | (c.rs << 1) | c.red32; }


// stem0 (ADD.py:153-157): 3x3 stride-2 convolution of the 3-channel image into 64 channels.  On the generic implicit-GEMM
// kernel every tap was a 32-channel chunk with 3 live k slots (9 chunks, 18 barriers, 8x the matrix work: 0.38 ms = 0.85 TB/s on a
// launch that moves 320 MB).  Here a tap is ONE k-step of v_mfma_f32_16x16x4_f32 — k = the pixel's 3 channels + a zero —
// with all 9 x 4 weight fragments register-stationary: lane (li, kq) loads channel kq of its pixel's nine neighbours (independent
// 4-byte loads, L1 hits: an input pixel is read by 2.25 outputs) and stores 16 bytes per output tile.  No LDS, no barrier.
struct StemK {
  const float* x; int ld; int N, H, W, OH, OW;
  const float* w; int ldw;
  float* y; int ldy;
  double* slab; int slab_ld;
  int P, ntiles16, rows, gx;
};
template <int CT, bool RED32>
__global__ void __launch_bounds__(256) stem0_kernel(const StemK p) {
  __shared__ double red[4][CT * 16][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  float wf[CT][9];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wf[i][tap] = kq < 3 ? ((const gfloat*)p.w)[(long)(i * 16 + li) * p.ldw + tap * 3 + kq] : 0.f;
  typedef typename std::conditional<RED32, float, double>::type red_t;
  red_t s1[CT][4], s2[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[i][e] = 0; s2[i][e] = 0; }
  const int ohw = p.OH * p.OW;
  const gfloat* xg = (const gfloat*)p.x;
  for (int tile = blockIdx.x * 4 + wave; tile < p.ntiles16; tile += p.gx * 4) {
    const int pp = tile * 16 + li;
    const bool pok = pp < p.P;
    const int n = pp / ohw, rem = pp - n * ohw, oh = rem / p.OW, ow = rem - oh * p.OW;
    float xv[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ih = 2 * oh - 1 + tap / 3, iw = 2 * ow - 1 + tap % 3;
      const bool ok = pok && kq < 3 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const float v = xg[ok ? ((long)(n * p.H + ih) * p.W + iw) * p.ld + kq : 0];
      xv[tap] = ok ? v : 0.f;
    }
    f32x4 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][tap], xv[tap], acc[i], 0, 0, 0);
    if (pok) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        st4(p.y + (long)pp * p.ldy + i * 16 + kq * 4, v);
        if (p.slab) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const red_t f = (red_t)get4(v, e); s1[i][e] += f; s2[i][e] += f * f; }
        }
      }
    }
  }
  if (p.slab) {
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red_t a = s1[i][e], b = s2[i][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        if (li == 0) { red[wave][i * 16 + kq * 4 + e][0] = (double)a; red[wave][i * 16 + kq * 4 + e][1] = (double)b; }
      }
    __syncthreads();
    if (t < CT * 16) {
      gdouble* o = (gdouble*)p.slab + ((long)blockIdx.x * p.slab_ld + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
      for (int r = blockIdx.x + p.gx; r < p.rows; r += p.gx) {     // rows no workgroup owns
        gdouble* z = (gdouble*)p.slab + ((long)r * p.slab_ld + t) * 2;
        z[0] = 0.0; z[1] = 0.0;
      }
    }
  }
}

}  // namespace

// Returns 0 when the launch was taken, 1 when the shape is not covered (caller falls back), <0 on error.
static bool pwk_covers(const addk_conv_args* a, bool& rs) {
  rs = false;
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->bias_n || a->H != a->OH || a->W != a->OW) return false;
  if (!aligned16(a->y) || a->ldy % 4 || a->Cout % 4 || !aligned16(a->w) || a->ldw % 4 || a->w_choff % 4) return false;
  int ktot = 0;
  for (int i = 0; i < a->nsrc; ++i) {
    if (!src_vec_ok(a->src[i])) return false;
    ktot += a->src[i].C;
    if (a->src[i].rs_hw) { rs = true; if ((a->src[i].rs_hw >> 16) < 1 || (a->src[i].rs_hw & 0xffff) < 1) return false; }
  }
  if (a->rs_y && (!a->src[0].rs_hw || !aligned16(a->rs_y) || a->rs_ldy % 4 || a->rs_ldy < a->src[0].C)) return false;
  // narrow outputs only: with 256 output channels (ASPP 1x1 1280->256) every wave would stream the whole 1.3 MB weight panel
  // and the LDS-staged kernel's operand reuse wins (63 vs 42 TF/s)
  if (ktot < 64 || a->Cout > 160 || (long)a->N * a->OH * a->OW < 1024) return false;
  return true;
}
static int pwk_try_fwd(const addk_conv_args* a, int rows, hipStream_t st) {
  bool rs;
  if (!pwk_covers(a, rs)) return 1;
  PwkK k{};
  for (int i = 0; i < a->nsrc; ++i) k.src[i] = a->src[i];
  k.nsrc = a->nsrc; k.Cn = a->Cout; k.w = a->w; k.ldw = a->ldw; k.w_off = a->w_choff;
  k.y = a->y; k.ldy = a->ldy; k.bias = a->bias;
  k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.P = a->N * a->OH * a->OW; k.ntiles16 = cdiv(k.P, 16); k.rows = rows;
  k.H = a->OH; k.W = a->OW; k.rs_y = a->rs_y; k.rs_ldy = a->rs_ldy;
  int ct = cdiv(a->Cout, 16); if (ct > 3) ct = 3;
  k.gx = rows; if (k.gx > cdiv(k.ntiles16, 4)) k.gx = cdiv(k.ntiles16, 4); if (k.gx < 1) k.gx = 1;
  dim3 grid(k.gx, cdiv(a->Cout, 16 * ct));
  const bool red32 = k.P >= 4096;
#define ADDK_PWK(CT_) \
  if (ct == CT_) { \
    if (rs) { \
      if (red32) hipLaunchKernelGGL((pwk_kernel<CT_, true, true>), grid, dim3(256), 0, st, k); \
      else hipLaunchKernelGGL((pwk_kernel<CT_, false, true>), grid, dim3(256), 0, st, k); \
    } else if (red32) hipLaunchKernelGGL((pwk_kernel<CT_, true>), grid, dim3(256), 0, st, k); \
    else hipLaunchKernelGGL((pwk_kernel<CT_, false>), grid, dim3(256), 0, st, k); \
    return addk_check_launch("pwk_conv"); }
  ADDK_PWK(1) ADDK_PWK(2) ADDK_PWK(3)
#undef ADDK_PWK
  return 1;
}


// stem0: 3 input channels (pixel stride 4), 3x3, stride 2, pad 1, 64 output channels, no prologue / bias
static int stem0_try_fwd(const addk_conv_args* a, int rows, hipStream_t st) {
  if (a->nsrc != 1 || a->KH != 3 || a->KW != 3 || a->stride != 2 || a->pad != 1 || a->dil != 1 || a->bias || a->bias_n) return 1;
  const addk_src& s = a->src[0];
  if (s.C != 3 || s.ld < 3 || s.a || s.b || s.relu || a->Cout != 64 || a->cin_total != 3 || a->w_choff != 0) return 1;
  if (!aligned16(a->y) || a->ldy % 4 || a->ldy < 64) return 1;
  if (a->OH != (a->H + 2 - 3) / 2 + 1 || a->OW != (a->W + 2 - 3) / 2 + 1) return 1;
  StemK k{};
  k.x = s.x; k.ld = s.ld; k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.w = a->w; k.ldw = a->ldw; k.y = a->y; k.ldy = a->ldy;
  k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.P = a->N * a->OH * a->OW; k.ntiles16 = cdiv(k.P, 16); k.rows = rows;
  k.gx = rows; if (k.gx > cdiv(k.ntiles16, 4)) k.gx = cdiv(k.ntiles16, 4); if (k.gx < 1) k.gx = 1;
  if (k.P >= 4096) hipLaunchKernelGGL((stem0_kernel<4, true>), dim3(k.gx), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((stem0_kernel<4, false>), dim3(k.gx), dim3(256), 0, st, k);
  return addk_check_launch("stem0");
}

int addk_pw_try_fwd(const addk_conv_args* a, int rows, void* stream) {
  { const int r = stem0_try_fwd(a, rows, (hipStream_t)stream); if (r <= 0) return r; }
  PwK k;
  if (pw_fill_fwd(a, k)) {
    const int r = pw_launch<PW_FWD>(k, rows, (hipStream_t)stream);
    if (r <= 0) return r;
  }
  return pwk_try_fwd(a, rows, (hipStream_t)stream);       // many input channels / several sources: streaming-K kernel
}
extern "C" int addk_conv_fwd_resample_ok(const addk_conv_args* a) {
  if (!a || a->nsrc < 1 || a->nsrc > ADDK_MAX_SRC || !(addk_get_fast_paths() & ADDK_FAST_PW)) return 0;
  bool any = false;
  for (int i = 0; i < a->nsrc; ++i) any = any || a->src[i].rs_hw != 0;
  if (!any || a->wpack) return 0;
  PwK k; PwCfg c;
  if (pw_fill_fwd(a, k) && pw_config(k, addk_conv_rows((long)a->N * a->OH * a->OW, a->Cout), c)) return 1;
  bool rs;
  return pwk_covers(a, rs) ? 1 : 0;
}
// Data gradient of a 1x1 convolution with FEW output channels into a wide input — the classifier (decoder.py last_conv, 256 -> 19 classes):
// 0.6 GF against 2 x 67 MB at config 2.  On the generic implicit-GEMM kernel (LDS tiles, 32-channel chunks with 19 live k slots) it took 110 us;
// here a lane owns four input channels (64 lanes = 256 channels), keeps its K x 4 weights in registers, and a wave walks pixels: the K values of
// dy of a pixel are loaded by K lanes and broadcast by v_readlane, the product is K x 4 fp32 FMAs per lane, the ReLU mask / (dA, dB) sums / store
// follow as in pw_kernel's data gradient.  Two pixels per trip (all loads of both before any use).
struct K1sK {
  const float* dy; int lddy, K;
  const float* w; int ldw, w_off;
  addk_src dst; float* g; int ldg, accumulate;
  double* slab; int P, rows, gx, Cn;
};
template <int KMAX>
__global__ void __launch_bounds__(256) k1s_dgrad_kernel(const K1sK p) {
  __shared__ double red[4][256][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = 4 * lane;
  const bool cok = c < p.Cn;
  float4 wq[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) wq[k] = (k < p.K && cok) ? ld4(p.w + (long)k * p.ldw + p.w_off + c) : zero4();
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  if (p.dst.a && cok) { av = ld4(p.dst.a + c); bv = ld4(p.dst.b + c); }
  const bool relu = p.dst.relu != 0, acc = p.accumulate != 0;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  const gfloat* dyg = (const gfloat*)p.dy;
  auto finish = [&](int pp, const float4 dz, const float4 x, const float4 old) {
    float4 g4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xe = get4(x, e), ae = get4(av, e), be = get4(bv, e), d = get4(dz, e);
      const bool m = !relu || fmaf(ae, xe, be) > 0.f;
      set4(g4, e, m ? d * ae : 0.f);
      if (m) { s1[e] += (double)d * (double)xe; s2[e] += (double)d; }
    }
    if (acc) { g4.x += old.x; g4.y += old.y; g4.z += old.z; g4.w += old.w; }
    st4(p.g + (long)pp * p.ldg + c, g4);
  };
  const int nw = p.gx * 4;
  for (int p0 = blockIdx.x * 4 + wave; p0 < p.P; p0 += 2 * nw) {
    const int p1 = p0 + nw;
    const bool ok1 = p1 < p.P;
    const float d0 = lane < p.K ? dyg[(long)p0 * p.lddy + lane] : 0.f;
    const float d1 = (ok1 && lane < p.K) ? dyg[(long)p1 * p.lddy + lane] : 0.f;
    float4 x0 = zero4(), x1 = zero4(), o0 = zero4(), o1 = zero4();
    if (cok) {
      x0 = ld4(p.dst.x + (long)p0 * p.dst.ld + c);
      if (ok1) x1 = ld4(p.dst.x + (long)p1 * p.dst.ld + c);
      if (acc) { o0 = ld4(p.g + (long)p0 * p.ldg + c); if (ok1) o1 = ld4(p.g + (long)p1 * p.ldg + c); }
    }
    float4 a0 = zero4(), a1 = zero4();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float u0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0), k)), u1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1), k));
      a0.x = fmaf(u0, wq[k].x, a0.x); a0.y = fmaf(u0, wq[k].y, a0.y); a0.z = fmaf(u0, wq[k].z, a0.z); a0.w = fmaf(u0, wq[k].w, a0.w);
      a1.x = fmaf(u1, wq[k].x, a1.x); a1.y = fmaf(u1, wq[k].y, a1.y); a1.z = fmaf(u1, wq[k].z, a1.z); a1.w = fmaf(u1, wq[k].w, a1.w);
    }
    if (cok) { finish(p0, a0, x0, o0); if (ok1) finish(p1, a1, x1, o1); }
  }
  if (p.slab) {      // the four waves through LDS in a fixed order; one slab row per workgroup, rows no workgroup owns zeroed
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][c + e][0] = s1[e]; red[wave][c + e][1] = s2[e]; }
    __syncthreads();
    if (t < p.Cn) {
      gdouble* o = (gdouble*)p.slab + ((long)blockIdx.x * p.Cn + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
      for (int r = blockIdx.x + p.gx; r < p.rows; r += p.gx) {
        gdouble* z = (gdouble*)p.slab + ((long)r * p.Cn + t) * 2;
        z[0] = 0.0; z[1] = 0.0;
      }
    }
  }
}
static int k1s_try_dgrad(const addk_conv_dgrad_args* a, int rows, hipStream_t st) {
  const int en = addk_env("ADDK_K1S", 1);
  if (!en || a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->H != a->OH || a->W != a->OW) return 1;
  if (a->Cout > 32 || a->dst.C < 128 || a->dst.C > 256 || a->dst.C % 4) return 1;
  if (!src_vec_ok(a->dst) || !aligned16(a->g) || a->ldg % 4 || !aligned16(a->w) || a->ldw % 4 || a->w_choff % 4) return 1;
  if (a->dst.a && (!aligned16(a->dst.a) || !aligned16(a->dst.b))) return 1;
  K1sK k{};
  k.dy = a->dy; k.lddy = a->lddy; k.K = a->Cout; k.w = a->w; k.ldw = a->ldw; k.w_off = a->w_choff;
  k.dst = a->dst; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate; k.slab = a->dab;
  k.P = a->N * a->H * a->W; k.rows = rows; k.gx = rows; k.Cn = a->dst.C;
  if (k.gx > cdiv(k.P, 8)) k.gx = cdiv(k.P, 8);
  if (k.gx < 1) k.gx = 1;
  if (a->Cout <= 20) hipLaunchKernelGGL((k1s_dgrad_kernel<20>), dim3(k.gx), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((k1s_dgrad_kernel<32>), dim3(k.gx), dim3(256), 0, st, k);
  return addk_check_launch("conv_dgrad (1x1, few output channels)");
}
int addk_pw_try_dgrad(const addk_conv_dgrad_args* a, int rows, void* stream) {
  PwK k;
  if (!pw_fill_dgrad(a, k)) return k1s_try_dgrad(a, rows, (hipStream_t)stream);
  return pw_launch<PW_DGRAD>(k, rows, (hipStream_t)stream);
}

// ---- batched form: mutually independent pointwise convs (one dependency level of the cell DAG) in one launch -------
// key >= 0: the launch runs on pw_kernel with that template variant (launches with equal keys can share a batch); -1: not
extern "C" int addk_conv_fwd_batch_key(const addk_conv_args* a) {
  if (!a || !(addk_get_fast_paths() & ADDK_FAST_PW)) return -1;
  PwK k; PwCfg c;
  if (!pw_fill_fwd(a, k) || !pw_config(k, addk_conv_rows((long)a->N * a->OH * a->OW, a->Cout), c)) return -1;
  return pw_key(c, PW_FWD);
}
extern "C" int addk_conv_dgrad_batch_key(const addk_conv_dgrad_args* a) {
  if (!a || !(addk_get_fast_paths() & ADDK_FAST_PW)) return -1;
  PwK k; PwCfg c;
  if (!pw_fill_dgrad(a, k) || !pw_config(k, addk_conv_rows((long)a->N * a->H * a->W, a->dst.C), c)) return -1;
  return pw_key(c, PW_DGRAD);
}
// host_blob = NULL: returns the blob size in bytes.  meta[0..5] = key, n, gx, gy, reserved
template <typename Args, typename Fill, typename Rows>
static int64_t pw_batch_prepare(const Args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta, int mode, Fill fill, Rows rows_of) {
  if (!a || n <= 0 || !meta) { addk_set_error("conv_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  const int64_t total = (int64_t)n * sizeof(PwK);
  if (host_blob && blob_bytes < total) { addk_set_error("conv_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  int key0 = -1, gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    PwK k; PwCfg c;
    if (!fill(&a[i], k) || !pw_config(k, rows_of(&a[i]), c)) { addk_set_error("conv_batch_prepare: launch %d is not a pointwise-kernel shape", i); return ADDK_ERR_INVALID; }
    const int key = pw_key(c, mode);
    if (i == 0) key0 = key;
    if (key != key0) { addk_set_error("conv_batch_prepare: mixed kernel variants"); return ADDK_ERR_INVALID; }
    if (c.gx > gx) gx = c.gx;
    if (c.gy > gy) gy = c.gy;
    if (host_blob) reinterpret_cast<PwK*>(host_blob)[i] = k;
  }
  meta[0] = key0; meta[1] = n; meta[2] = gx; meta[3] = gy;
  return total;
}
extern "C" int64_t addk_conv_fwd_batch_prepare(const addk_conv_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  return pw_batch_prepare(a, n, host_blob, blob_bytes, meta, PW_FWD, pw_fill_fwd,
                          [](const addk_conv_args* x) { return addk_conv_rows((long)x->N * x->OH * x->OW, x->Cout); });
}
extern "C" int64_t addk_conv_dgrad_batch_prepare(const addk_conv_dgrad_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  return pw_batch_prepare(a, n, host_blob, blob_bytes, meta, PW_DGRAD, pw_fill_dgrad,
                          [](const addk_conv_dgrad_args* x) { return addk_conv_rows((long)x->N * x->H * x->W, x->dst.C); });
}
extern "C" int addk_conv_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[1] > 0 && meta[2] > 0 && meta[3] > 0, "conv_batch_run: bad args");
  const int key = (int)meta[0], mode = key >> 12, ct = (key >> 8) & 15, kg = (key >> 4) & 15, red32 = key & 1, rs = (key >> 1) & 1;
  const PwK* tab = reinterpret_cast<const PwK*>(dev_blob);
  if (mode == PW_FWD) return pw_batch_launch<PW_FWD>(tab, (int)meta[1], ct, kg, red32, rs, (int)meta[2], (int)meta[3], (hipStream_t)stream);
  return pw_batch_launch<PW_DGRAD>(tab, (int)meta[1], ct, kg, red32, 0, (int)meta[2], (int)meta[3], (hipStream_t)stream);
}
