// Halo-patch form of the wide 3x3, stride-1, 'same'-padded convolutions (decoder, ASPP dilated branches, stem1),
// forward and data gradient, on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// The generic implicit-GEMM kernel (conv.hip) walks K as (tap, 32-channel chunk) and re-stages the pixel tile for every
// tap: 9 global loads + 9 BatchNorm/ReLU prologues per input element, two barriers per 128 MFMAs, and the staging phase
// does not overlap the MFMA phase (measured: 454 us of staging + 714 us of MFMAs ~ the 1072 us of the decoder conv).
// Here a block owns 128 consecutive pixels of ONE output row and 64/128 output channels, and walks K in 16-channel
// chunks.  Per chunk it stages the three input rows oh-d, oh, oh+d ([3][128+2d px][16 ch], prologue applied once, zero
// padding after it) in LDS and runs all nine taps from that patch: one global load + one prologue per input element
// and 2 barriers per 576 MFMAs per wave.  The weights never touch LDS: a small pack kernel rewrites them per launch
// into MFMA-fragment order ([column block][chunk*9+tap][16-col tile][lane][4 k-slots]) so that a wave fetches the
// fragments of one tap with CT coalesced 16-byte loads (L1/L2 hits: every block reads the same stream), one tap ahead
// of their use.  The data gradient is the same kernel over dy with the packed weights transposed and tap-mirrored.
//
// Reference call sites: decoder.py:17-27 (3x3 conv+BN+ReLU x2), aspp_train.py:20-41 (dilated 3x3 branches),
// ADD.py:220-232 (stem1); autograd of nn.Conv2d for the data gradient.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
constexpr int C3_BP = 128;                 // pixels per tile (one row segment)
constexpr int C3_BK = 16;                  // channels per chunk
constexpr int C3_RS = 20;                  // floats per patch pixel in LDS (16 + 4: ds_read_b64 fragment reads hit 64 distinct banks per half-wave)
constexpr int C3_MAXCH = 64;               // chunks per launch (K <= 1024 channels)
// widest patch row: 3x3 up to dilation 18 (ASPP), 5x5 up to dilation 2 (the cells' dil_conv_5x5)
constexpr int c3_pwmax(int ks) { return ks == 3 ? C3_BP + 2 * 18 : C3_BP + 4 * 2; }
constexpr int c3_maxdil(int ks) { return ks == 3 ? 18 : 2; }

struct C3K {
  addk_src src[ADDK_MAX_SRC];
  int nsrc;
  int N, H, W, dil;       // H, W: the OUTPUT map (tiles, epilogue)
  int HT;                   // tile rows in the walk: H, or ceil(H / 2) for two-row tiles
  int IH, IW;               // the input map (= H, W for the stride-1 launches)
  int Cn, ldy;
  float* y;
  const float* wp;          // packed weights of this launch
  long wp_blk;              // floats per column block in wp
  int nT;                   // chunks * taps
  const float* bias; const float* bias_n;
  double* slab; int slab_ld;
  addk_src dst; int accumulate;
  int vecY, red32;
  long P; int ntiles, spr;
  int st;                   // stride (host-side dispatch; 2 = the de-interleaved 3x3 forward of conv3b_kernel)
  int om, oro, oco, OHo, OWo;   // output pixel of tile-grid position (oh, ow): (om oh + oro, om ow + oco) in an OHo x OWo map (om = 1: the grid itself)
};

struct PackK {
  const float* w; int ldw, cin_total;
  int mode;                 // fwd: rows = co, K = input channels; dgrad: rows = input channels, K = co
  int Cn;                   // GEMM N extent (rows of the packed tiles)
  int w_choff;              // dgrad: channel offset of the destination source inside a tap
  int nchunks, bct, taps;
  int cbase[C3_MAXCH];      // first K index of the chunk (fwd: absolute channel inside a tap; dgrad: co)
  int cvalid[C3_MAXCH];     // valid K entries in the chunk (<= 16)
  float* out;
  int planes;               // 0: fp32 fragments for conv3_kernel; 2 / 3: bf16 planes (h, m[, l]) in 32-row fragment order for conv3b_kernel
  int dil_odd;              // conv3b checkerboard accumulation (see conv3b_kernel): 1 = odd dilation, taps with odd (kh + kw) carry a minus sign;
                            // 2 = stride 2, the taps of kernel column 2 carry it
  int s2d;                  // conv3b: 1 = the four parity-class packs of the stride-2 data gradient (c3b_pack_s2d_body)
};

// out[colblk][T = chunk*taps + tap][tile i][lane][m]  =  W(row = colblk*BC + i*16 + li, tap, k = 8*(m>>1) + 2*kq + (m&1))
__device__ __forceinline__ void c3_pack_body(const PackK& p, long first, long stride);
__global__ void __launch_bounds__(256) c3_pack_kernel(const PackK p) { c3_pack_body(p, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256); }
// all packs of a plan in one launch at the start of the step (the weights only change in the optimizer): block (x, y) works on descriptor y
__global__ void __launch_bounds__(256) c3_pack_batch_kernel(const PackK* __restrict__ descs) {
  c3_pack_body(descs[blockIdx.y], (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}
__device__ __forceinline__ void c3b_pack_body(const PackK& p, long first, long stride);
__device__ __forceinline__ void c3_pack_body(const PackK& p, long first, long stride) {
  if (p.planes) { c3b_pack_body(p, first, stride); return; }
  const int BC = 16 * p.bct;
  const int nT = p.nchunks * p.taps;
  const long per_blk = (long)nT * p.bct * 256;
  const long total = (long)((p.Cn + BC - 1) / BC) * per_blk;
  for (long idx = first; idx < total; idx += stride) {
    const int m = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long r = idx >> 8;
    const int i = (int)(r % p.bct); r /= p.bct;
    const int T = (int)(r % nT); const int blk = (int)(r / nT);
    const int chunk = T / p.taps, tap = T - chunk * p.taps;
    const int li = lane & 15, kq = lane >> 4;
    const int row = blk * BC + i * 16 + li;
    const int kk = 8 * (m >> 1) + 2 * kq + (m & 1);
    float v = 0.f;
    if (row < p.Cn && kk < p.cvalid[chunk]) {
      if (p.mode == MODE_FWD) v = p.w[(long)row * p.ldw + (long)tap * p.cin_total + p.cbase[chunk] + kk];
      else                    v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(p.taps - 1 - tap) * p.cin_total + p.w_choff + row];
    }
    p.out[idx] = v;
  }
}

template <int BCT, int KS, int MODE>
__global__ void __launch_bounds__(256, 2) conv3_kernel(const C3K p) {
  constexpr int BC = 16 * BCT;
  constexpr int TAPS = KS * KS, HK = KS / 2;
  constexpr int C3_PWMAX = c3_pwmax(KS);
  constexpr int C3_NS = (KS * C3_PWMAX * 4 + 255) / 256;   // 16-byte patch slots per thread
  constexpr int WC = (BCT == 8 || BCT == 4) ? 2 : 1;      // waves across output channels
  constexpr int WP = 4 / WC;                // waves across pixels
  constexpr int PT = 8 / WP;                // 16-pixel tiles per wave
  constexpr int CT = BCT / WC;              // 16-channel tiles per wave
  __shared__ __attribute__((aligned(16))) float Ps[KS * C3_PWMAX * C3_RS];
  __shared__ double red[WP][BC][2];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int wc = wave % WC, wpx = wave / WC;
  const int n0 = blockIdx.y * BC;
  const int d = p.dil, PW = C3_BP + (KS - 1) * d;
  double tot0 = 0.0, tot1 = 0.0;

  // fixed patch-slot geometry, packed to one register per slot: slot -> (row r | patch column pj << 3); r == KS: outside the patch
  const int q = t & 3;
  int geo[C3_NS];
#pragma unroll
  for (int k = 0; k < C3_NS; ++k) {
    const int pix = (t + 256 * k) >> 2;
    int r = pix / PW; const int pj = pix - r * PW;
    if (r > KS) r = KS;
    geo[k] = r | (pj << 3);
  }
  // fragment read bases (bytes are implied by float indexing): pixel (wpx*PT*16 + li) of patch row 0, k pair 2*kq, per kw
  int xb[KS];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) xb[kw] = (wpx * PT * 16 + li + kw * d) * C3_RS + 2 * kq;
  const float4* wpl = reinterpret_cast<const float4*>(p.wp + (long)blockIdx.y * p.wp_blk) + (long)wc * CT * 64 + lane;
  const int nT = p.nT;

  const int tpx = p.ntiles >> 3;
  const bool swz = (p.ntiles & 7) == 0 && p.ntiles >= 64;
  for (int tlin = blockIdx.x; tlin < p.ntiles; tlin += gridDim.x) {
    const int tile = swz ? (tlin & 7) * tpx + (tlin >> 3) : tlin;
    const int rowid = tile / p.spr, sx = tile - rowid * p.spr;
    const int n = rowid / p.H, oh = rowid - n * p.H;
    const int ow0 = sx * C3_BP;
    // per-tile validity of the slots (image borders)
    unsigned vmask = 0;
    const int pbase = (n * p.H + oh - HK * d) * p.W + ow0 - HK * d;      // pixel index of patch (row 0, column 0)
#pragma unroll
    for (int k = 0; k < C3_NS; ++k) {
      const int r = geo[k] & 7, pj = geo[k] >> 3;
      const int ih = oh + (r - HK) * d, iw = ow0 - HK * d + pj;
      const bool ok = r < KS && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      vmask |= (ok ? 1u : 0u) << k;
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[C3_NS];
    float4 pa = make_float4(1.f, 1.f, 1.f, 1.f), pb = zero4();
    bool prelu = false, pch = false;
    auto load_patch = [&](int s_, int c0_) {      // branch-free: masked slots read the source base and are zeroed at store time
      const addk_src S = p.src[s_];
      const int c = c0_ + 4 * q;
      pch = c < S.C;
      prelu = S.relu != 0;
      pa = make_float4(1.f, 1.f, 1.f, 1.f); pb = zero4();
      if (S.a && pch) { pa = ld4(S.a + c); pb = ld4(S.b + c); }
      const float* sb = S.x + (pch ? c : 0);
#pragma unroll
      for (int k = 0; k < C3_NS; ++k) {
        const int r = geo[k] & 7, pj = geo[k] >> 3;
        const int po = ((vmask >> k) & 1u) ? pbase + r * d * p.W + pj : 0;
        ra[k] = ld4(sb + (long)po * S.ld);
      }
    };
    auto store_patch = [&]() {
#pragma unroll
      for (int k = 0; k < C3_NS; ++k) {
        float4 v = ra[k];
        v.x = fmaf(pa.x, v.x, pb.x); v.y = fmaf(pa.y, v.y, pb.y); v.z = fmaf(pa.z, v.z, pb.z); v.w = fmaf(pa.w, v.w, pb.w);
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = pch && ((vmask >> k) & 1u);
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        const int r = geo[k] & 7, pj = geo[k] >> 3;
        if (r < KS) lds_st4(&Ps[(r * C3_PWMAX + pj) * C3_RS + 4 * q], v);
      }
    };

    // weight fragments: two register sets, the next tap's fragments are fetched while the current tap's MFMAs issue
    float4 wr[2][CT];
    auto load_w = [&](int T, float4* dst) {
      const int Tc = T < nT ? T : nT - 1;
      const float4* src = wpl + (long)Tc * (BCT * 64);
#pragma unroll
      for (int i = 0; i < CT; ++i) dst[i] = src[i * 64];
    };
    // pixel fragments: one buffer per half of the 16-channel chunk (k 0-7 / 8-15), read one half ahead
    float2 xf[2][PT];
    auto read_x = [&](int tap, int h, float2* x) {
      const int kh = tap / KS, kw = tap - kh * KS;
      const float* b = &Ps[xb[kw] + kh * C3_PWMAX * C3_RS + 8 * h];
#pragma unroll
      for (int j = 0; j < PT; ++j) x[j] = *reinterpret_cast<const float2*>(b + j * 16 * C3_RS);
    };
    auto mma = [&](const float4* w, int h, const float2* x) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(h ? w[i].z : w[i].x, x[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(h ? w[i].w : w[i].y, x[j].y, acc[i][j], 0, 0, 0);
    };

    int s = 0, c0 = 0, T0 = 0;
    load_w(0, wr[0]);
    load_patch(0, 0);
    store_patch();
    __syncthreads();
    while (true) {
      int s2 = s, c2 = c0 + C3_BK;
      if (c2 >= p.src[s].C) { c2 = 0; ++s2; }
      const bool more = s2 < p.nsrc;
      read_x(0, 0, xf[0]);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        load_w(T0 + tap + 1, wr[(tap + 1) & 1]);
        if (tap == 0 && more) load_patch(s2, c2);
        read_x(tap, 1, xf[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(wr[tap & 1], 0, xf[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < TAPS) read_x(tap + 1, 0, xf[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(wr[tap & 1], 1, xf[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      T0 += TAPS;
#pragma unroll
      for (int i = 0; i < CT; ++i) wr[0][i] = wr[1][i];      // odd tap count: the set fetched during the last tap is next chunk's tap 0
      __syncthreads();
      if (!more) break;
      s = s2; c0 = c2;
      store_patch();
      __syncthreads();
    }

    // ---- epilogue (same contract as conv.hip) ----
    const bool want_red = p.slab != nullptr;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int col = (wc * CT + i) * 16 + kq * 4;
      const int c = n0 + col;
      const int nrem = p.Cn - c;
      double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int lp = (wpx * PT + j) * 16 + li;
        const long pp = (long)rowid * p.W + ow0 + lp;
        const bool pv = ow0 + lp < p.W && nrem > 0;
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (MODE == MODE_FWD) {
          if (pv) {
            if (p.bias) { float4 b = ld4g(p.bias + c, nrem, false); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.bias_n) {
              float4 b = ld4g(p.bias_n + (long)n * p.Cn + c, nrem, false);
              v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            st4g(p.y + pp * p.ldy + c, v, nrem, p.vecY);
            if (want_red) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                double f = (e < nrem) ? (double)get4(v, e) : 0.0;
                s1[e] += f; s2v[e] += f * f;
              }
            }
          }
        } else {
          if (pv) {
            float4 x = ld4g(p.dst.x + pp * p.dst.ld + c, nrem, p.vecY);
            float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
            if (p.dst.a) { av = ld4g(p.dst.a + c, nrem, p.vecY); bv = ld4g(p.dst.b + c, nrem, p.vecY); }
            float4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float xe = get4(x, e), ae = get4(av, e), be = get4(bv, e), dz = get4(v, e);
              bool m = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
              set4(g, e, m ? dz * ae : 0.f);
              if (want_red && m) { s1[e] += (double)dz * (double)xe; s2v[e] += (double)dz; }
            }
            float* gp = p.y + pp * p.ldy + c;
            if (p.accumulate) { float4 o = ld4g(gp, nrem, p.vecY); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
            st4g(gp, g, nrem, p.vecY);
          }
        }
      }
      if (want_red) {
        if (p.red32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = (float)s1[e], b = (float)s2v[e];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
            if (li == 0) { red[wpx][col + e][0] = a; red[wpx][col + e][1] = b; }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            double a = s1[e], b = s2v[e];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
            if (li == 0) { red[wpx][col + e][0] = a; red[wpx][col + e][1] = b; }
          }
        }
      }
    }
    if (want_red) {
      __syncthreads();
      if (t < BC) {
#pragma unroll
        for (int w = 0; w < WP; ++w) { tot0 += red[w][t][0]; tot1 += red[w][t][1]; }
      }
      __syncthreads();
    }
  }
  if (p.slab && t < BC && n0 + t < p.Cn) {
    double* o = p.slab + ((long)blockIdx.x * p.slab_ld + n0 + t) * 2;
    o[0] = tot0; o[1] = tot1;
  }
}


// ======================================================================================================================
// Split-bf16 form of the same halo-patch convolution: every fp32 operand is written as x = h + m + l with h, m, l in bf16
// (3 x 8 significand bits = the 24 of fp32, exactly) and a product is evaluated on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) as the sum of its six largest bf16 x bf16 terms
//     x*w  ~=  l*wh + h*wl + m*wm + m*wh + h*wm + h*wh          (dropped: m*wl, l*wm, l*wl  ~ 2^-26 relative)
// Measured (scripts/bf16_split_probe.hip, K = 2048, random magnitudes over 6 octaves): max |err| / sum|a*b| = 3.0e-7, rms 5.6e-8 —
// the same as the exact-fp32 v_mfma_f32_16x16x4_f32 chain (3.2e-7 / 6.8e-8) — at 6 x 16 = 96 matrix-pipe cycles per
// 16x16x32 block of fp32 work instead of 256: 2.5x the fp32 MFMA rate (374 vs 149 TFLOP/s from registers).  The 3-term form
// (h*wh + m*wh + h*wm, planes = 2) is the opt-in fast mode: 656 TFLOP/s, rms error 4.7e-7.
//
// Structure.  A block owns 128 pixels of one output row x (32 * WC) output channels; WAVE w owns channels [32w, 32w+32) of ALL
// 128 pixels (four 32x32 accumulator tiles): the pixel fragments are shared through LDS (which has the bandwidth: 128 B/clk
// per CU at 8 waves), the weight fragments are private to a wave and stream from L1/L2 (32 B/clk).  K is walked in
// 16-channel chunks = one MFMA k-step; per chunk the KS input rows are staged ONCE: fp32 loads, lazy BatchNorm/ReLU prologue,
// zero padding, then the split into planes — so the ~6 VALU operations per element of the split are paid once per staged
// element, not per use (each element feeds KS*KS * 6 * WC MFMA operands).  LDS image per plane: [row][pixel][16 ch] bf16 =
// 32 B per pixel, the two 16-byte halves of a pixel swapped when bit 3 of the pixel index is set: the ds_read_b128 of a
// 32-pixel fragment then touches all 64 banks once per 16-lane group for every tap shift (no padding needed).
// ======================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CB_PT = 4;                   // 32-pixel tiles per wave (128 pixels)
constexpr int cb_pwmax(int ks, bool bigd, int bpx = C3_BP, int st = 1) { return st == 2 ? 2 * bpx + 16 : ks > 10 ? bpx + 4 : ks == 1 ? bpx : ks == 3 ? (bigd ? bpx + 2 * 18 : bpx + 2 * 2) : bpx + 4 * 2; }

__device__ __forceinline__ unsigned bf16_hi(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_f(unsigned b) { return __uint_as_float(b << 16); }
// 4 floats -> 4 bf16 per plane (8 bytes each)
template <int NP>
__device__ __forceinline__ void split4(const float4 v, uint2 (&pl)[NP]) {
  float r[4] = {v.x, v.y, v.z, v.w};
  unsigned b[NP][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = r[e];
#pragma unroll
    for (int k = 0; k < NP; ++k) { b[k][e] = bf16_hi(x); x = x - bf16_f(b[k][e]); }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) pl[k] = make_uint2(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16));
}

// packed weights: out (16-byte units) [colblk][T = chunk*taps + tap][tile (wave) i][plane][lane] = 8 bf16:
//   W(row = colblk*32*bct + i*32 + (lane & 31), tap, k = 8*(lane >> 5) + j), j = 0..7       (bct = 32-row tiles per block)
// Stride-2 data gradient (3x3, pad 1): input pixel (2a + pi, 2b + pj) collects dy(a + th, b + tw) W[kh][kw] over th <= pi, tw <= pj with
// kh = 1 for pi = 0, else {2, 0}[th] (same for columns) — 1, 2, 2, 4 of the nine taps per parity class.  Four packs back to back, class
// c = 2 pi + pj at (taps of the classes before) * unit, each [colblk][chunk * tc + tap][tile][plane][lane]; tap (th, tw) carries
// (-1)^(th + tw), its share of the checkerboard sign over the class grid (a, b).
__device__ __forceinline__ void c3b_pack_s2d_body(const PackK& p, long first, long stride) {
  const int BC = 32 * p.bct, NP = p.planes;
  const long unit = (long)((p.Cn + BC - 1) / BC) * p.nchunks * p.bct * 64;          // lanes per tap
  uint4* out = reinterpret_cast<uint4*>(p.out);
  for (long idx = first; idx < 9 * unit; idx += stride) {
    const int u = (int)(idx / unit);
    const int cls = u < 1 ? 0 : u < 3 ? 1 : u < 5 ? 2 : 3, pre = cls == 0 ? 0 : cls == 1 ? 1 : cls == 2 ? 3 : 5;
    const int pi = cls >> 1, pj = cls & 1, twn = 1 + pj, tc = (1 + pi) * twn;
    long r = idx - pre * unit;
    const int lane = (int)(r & 63); r >>= 6;
    const int i = (int)(r % p.bct); r /= p.bct;
    const int nT = p.nchunks * tc;
    const int T = (int)(r % nT); const int blk = (int)(r / nT);
    const int chunk = T / tc, tap = T - chunk * tc;
    const int th = tap / twn, tw = tap - th * twn;
    const int kh = pi ? (th ? 0 : 2) : 1, kw = pj ? (tw ? 0 : 2) : 1;
    const int row = blk * BC + i * 32 + (lane & 31), k0 = 8 * (lane >> 5);
    unsigned b[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      const int kk = k0 + j;
      if (row < p.Cn && kk < p.cvalid[chunk]) v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(kh * 3 + kw) * p.cin_total + p.w_choff + row];
      if ((th + tw) & 1) v = -v;
#pragma unroll
      for (int k = 0; k < 3; ++k) { b[k][j] = bf16_hi(v); v = v - bf16_f(b[k][j]); }
    }
    uint4* o = out + pre * unit * NP + (((long)blk * nT + T) * p.bct + i) * NP * 64 + lane;
    for (int k = 0; k < NP; ++k)
      o[k * 64] = make_uint4(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16), b[k][4] | (b[k][5] << 16), b[k][6] | (b[k][7] << 16));
  }
}

__device__ __forceinline__ void c3b_pack_body(const PackK& p, long first, long stride) {
  if (p.s2d) { c3b_pack_s2d_body(p, first, stride); return; }
  const int BC = 32 * p.bct, NP = p.planes;
  const int nT = p.nchunks * p.taps;
  const long per_blk = (long)nT * p.bct * 64;
  const long total = (long)((p.Cn + BC - 1) / BC) * per_blk;
  uint4* out = reinterpret_cast<uint4*>(p.out);
  for (long idx = first; idx < total; idx += stride) {
    const int lane = (int)(idx & 63);
    long r = idx >> 6;
    const int i = (int)(r % p.bct); r /= p.bct;
    const int T = (int)(r % nT); const int blk = (int)(r / nT);
    const int chunk = T / p.taps, tap = T - chunk * p.taps;
    const int row = blk * BC + i * 32 + (lane & 31), k0 = 8 * (lane >> 5);
    const int ks = p.taps == 9 ? 3 : 5;
    const bool flip = p.dil_odd == 1 ? (((tap / ks) + (tap % ks)) & 1) != 0      // (-1)^((kh+kw) d): the tap's share of the checkerboard sign
                    : p.dil_odd == 2 ? (tap % ks) == 2 : false;
    unsigned b[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      const int kk = k0 + j;
      if (row < p.Cn && kk < p.cvalid[chunk]) {
        if (p.mode == MODE_FWD) v = p.w[(long)row * p.ldw + (long)tap * p.cin_total + p.cbase[chunk] + kk];
        else                    v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(p.taps - 1 - tap) * p.cin_total + p.w_choff + row];
      }
      if (flip) v = -v;
#pragma unroll
      for (int k = 0; k < 3; ++k) { b[k][j] = bf16_hi(v); v = v - bf16_f(b[k][j]); }
    }
    uint4* o = out + (((long)blk * nT + T) * p.bct + i) * NP * 64 + lane;
    for (int k = 0; k < NP; ++k)
      o[k * 64] = make_uint4(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16), b[k][4] | (b[k][5] << 16), b[k][6] | (b[k][7] << 16));
  }
}

// PH = 2 (the <= 64-channel launches): the block's waves are also split over the two 64-pixel halves of the tile — wave w owns
// channels [32 (w % WC), +32) of pixels [64 (w / WC), +64), two accumulator tiles — so that 4 waves share the staging work of
// a 64-channel block (2-wave blocks staged 13-22 slots per thread and spilled).
// BPX = 64: half-width pixel tiles (two accumulator tiles per wave) for launches that would otherwise put fewer than ~1.5
// blocks on a CU (ASPP at 64x128, the 80-channel cell convs at 63x127: 126-256 blocks of 128 pixels = one wave per SIMD or less).
// ST = 2 (stem2: 3x3, stride 2, pad 1, forward): output pixel ow reads input columns 2 ow - 1 + kw.  The patch row is staged
// DE-INTERLEAVED — even patch columns pj = 2 e at LDS position e (0..BPX), odd ones pj = 2 o + 1 at position OB + o — so that
// the fragment of tap kw is again 32 CONSECUTIVE positions (kw = 0: e = lp, kw = 1: o = lp, kw = 2: e = lp + 1) and the
// swizzle / bank picture of the stride-1 kernel holds unchanged.  Checkerboard: position idx carries the sign (-1)^(oh+ow0+idx),
// which is the output pixel's for kw = 0, 1 and its negative for kw = 2 — that tap's weights are packed negated.
#ifdef ADDK_C3B_DIAG
// diagnostic build (make CXXFLAGS+=-DADDK_C3B_DIAG, scripts/c3b_clock.sh): every workgroup adds its lifetime in shader-clock ticks (s_memtime)
// and in 100 MHz reference ticks (s_memrealtime): their ratio is the clock the CUs ran at INSIDE this kernel
__device__ unsigned long long g_c3b_diag[64][4];        // 64 slots: the workgroups' atomics do not queue on one L2 line
#endif
// TR = 2 (3x3 / 5x5, dilation d <= 2): the tile is TWO output rows of BPX pixels, d rows apart — oh0 and oh0 + d, whose input rows oh0 + (r - HK) d,
// r = 0..KS, overlap in KS - 1 of KS + 1 — with the accumulator count of one row of 2 BPX pixels: the patch holds KS + 1 input rows instead
// of 2 KS for the same outputs, 33 % (3x3) / 40 % (5x5) less staging (global loads, prologue, split, LDS writes) and HBM-side traffic.  The
// tile walk counts row PAIRS (p.HT; pair q -> oh0 = (q / d) 2d + q % d) and the epilogue masks second rows beyond the map.
template <int WC, int KS, int MODE, int NP, bool BIGD, int PH = 1, int BPX = C3_BP, int ST = 1, int TR = 1>
__device__ __forceinline__ void conv3b_body(const C3K& p, const int bx, const int gx) {      // workgroup bx of gx along x (tiles, slab row)
#ifdef ADDK_C3B_DIAG
  const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  static_assert(ST == 1 || (ST == 2 && KS == 3 && !BIGD && MODE == MODE_FWD), "stride 2: 3x3 forward only");
  // KS = 12 / 21 / 22: a 1x2 / 2x1 / 2x2 tap set anchored at its first tap (no centring) — the parity classes of the stride-2 data
  // gradient (c3b_s2_dgrad below); 1, 3, 5: the centred square kernels
  constexpr int KH_ = KS > 10 ? KS / 10 : KS, KW_ = KS > 10 ? KS % 10 : KS;
  static_assert(KS < 10 || (MODE == MODE_DGRAD && !BIGD && ST == 1), "anchored tap sets: data gradient only");
  constexpr int OB = BPX + 16;                                      // ST = 2: LDS position of the first odd patch column
  static_assert(TR == 1 || (TR == 2 && (KS == 3 || KS == 5) && !BIGD && ST == 1), "two-row tiles: centred kernels at dilation 1");
  constexpr int BC = 32 * WC, NTHR = 64 * WC * PH, PT = TR * BPX / 32 / PH, TPR = BPX / 32;      // TPR: 32-pixel tiles per tile row
  constexpr int PR_ = (KS > 10 ? KS / 10 : KS) + TR - 1;                                          // patch rows
  constexpr int TAPS = KH_ * KW_, HK = KS > 10 ? 0 : KS / 2;
  constexpr int PWP = cb_pwmax(KS, BIGD, BPX, ST);                         // LDS row pitch in pixels (compile time: tap offsets are immediates)
  constexpr int NS = (PR_ * PWP * 4 + NTHR - 1) / NTHR;           // 16-byte (4-channel) patch slots per thread, enumerated over the pitch grid
  static_assert(NS <= 32, "slot mask is 32 bits");
  constexpr int PLANE = PR_ * PWP * 2;                             // uint4 units per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red = reinterpret_cast<double*>(smem);                  // [PH][BC][2] running statistics of this block (per pixel half)
  uint4* Pl = reinterpret_cast<uint4*>(smem + ((PH * BC * 16 + 15) & ~15));   // [NP][KS][PWP][2] 16-byte halves
  uint2* Pl2 = reinterpret_cast<uint2*>(Pl);

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wave = wv % WC, wpx = wv / WC, lp32 = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.y * BC;
  const int d = p.dil;
  for (int i = t; i < PH * BC * 2; i += NTHR) red[i] = 0.0;

  const int q = t & 3;
  // patch slot -> (patch row r, LDS position sp in the row, patch column pj = input column - first input column of the tile)
  auto slot_geo = [&](int k, int& r, int& sp, int& pj, bool& live) {
    const int pix = (t + NTHR * k) >> 2;
    r = pix / PWP; sp = pix - r * PWP;
    if (ST == 1) { pj = sp; live = r < PR_ && pj < BPX + (KW_ - 1) * p.dil; }
    else { const bool odd = sp >= OB; const int idx = odd ? sp - OB : sp; pj = 2 * idx + (odd ? 1 : 0); live = r < KH_ && (odd ? idx < BPX : idx <= BPX); }
  };
  // fragment read base per kernel column: pixel lane%32 + kw*d of patch row 0, the 16-byte half swizzled by bit 3 of the pixel
  // (tile j adds 32 pixels: bit 3 unchanged); patch row kh and tile j are immediate offsets
  int xb[KW_];
  // 32-pixel tile j of this wave: tile row wrow + jrow(j), column tile wcol + jcol(j) (4-wave blocks of <= 64 channels split the tiles over two wave pairs:
  // by rows when the tile has two, else by columns)
  const int wrow = (PH == 2 && TR == 2) ? wpx : 0, wcol = (PH == 2 && TR == 1) ? wpx * PT : 0;
  auto jrow = [](int j) { return PH == 1 ? j / TPR : 0; };
  auto jcol = [](int j) { return PH == 1 ? j % TPR : j; };
#pragma unroll
  for (int kw = 0; kw < KW_; ++kw) {
    const int pj = ST == 2 ? (kw == 1 ? OB + lp32 : lp32 + (kw >> 1)) : lp32 + kw * d;
    xb[kw] = pj * 2 + (hh ^ ((pj >> 3) & 1)) + (wrow * PWP + wcol * 32) * 2;
  }
  const uint4* wpl = reinterpret_cast<const uint4*>(p.wp) + ((long)blockIdx.y * p.wp_blk + (long)wave * NP * 64 + lane);
  const int nT = p.nT;
  unsigned pmask = 0;                         // bit k: parity of (r*d + pj) of this thread's patch slot k (checkerboard sign, below)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int r, sp, pj; bool live;
    slot_geo(k, r, sp, pj, live);
    pmask |= (unsigned)((ST == 2 ? sp : r * d + pj) & 1) << k;
  }

  const int tpx = p.ntiles >> 3;
  const bool swz = (p.ntiles & 7) == 0 && p.ntiles >= 64;
  for (int tlin = bx; tlin < p.ntiles; tlin += gx) {
    const int tile = swz ? (tlin & 7) * tpx + (tlin >> 3) : tlin;
    const int rowid = tile / p.spr, sx = tile - rowid * p.spr;
    const int n = rowid / p.HT, prq = rowid - n * p.HT;
    const int oh = TR == 1 ? prq : (prq / d) * (2 * d) + prq % d;
    const int ow0 = sx * BPX;
    unsigned vmask = 0;
    const int pbase = (n * p.IH + oh * ST - HK * d) * p.IW + ow0 * ST - HK * d;
    const unsigned par0 = (unsigned)(oh + ow0);                 // parity of (ih + iw) of patch element (r, pj) = par0 + r*d + pj (the -2 HK d is even)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int r, sp, pj; bool live;
      slot_geo(k, r, sp, pj, live);
      const int ih = oh * ST + (r - HK) * d, iw = ow0 * ST - HK * d + pj;
      const bool ok = live && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      vmask |= (ok ? 1u : 0u) << k;
    }

    f32x16 acc[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // Blocked accumulation (the <= 64-channel launches: stem1, the cells' 40-channel dilated convs — the layers every later layer
    // inherits its error from): the running sums are flushed into a second accumulator set after every 16-channel chunk, so a
    // rounding error grows with sqrt(MFMAs per chunk) + sqrt(chunks) instead of sqrt(all MFMAs) (stem1: 54 + 4 instead of 216) —
    // what a CPU library's blocked partial sums do (the even-size gradient deficit starts at stem1's K = 576 chain).
    constexpr bool BLK = PH == 2 && (KS == 3 || KS > 10) && !BIGD;          // (the 5x5 and wide-dilation variants have no registers left: 256 + scratch with a second set)
    f32x16 acc2[BLK ? PT : 1];
    if (BLK) {
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
    }

    float4 ra[NS];
    float4 pa = make_float4(1.f, 1.f, 1.f, 1.f), pb = zero4();
    bool prelu = false, pch = false;
    auto load_patch = [&](int s_, int c0_) {      // branch-free: masked slots read the source base and are zeroed at store time
      const addk_src S = p.src[s_];
      const int c = c0_ + 4 * q;
      pch = c < S.C;
      prelu = S.relu != 0;
      pa = make_float4(1.f, 1.f, 1.f, 1.f); pb = zero4();
      if (S.a && pch) { pa = ld4(S.a + c); pb = ld4(S.b + c); }
      const float* sb = S.x + (pch ? c : 0);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        int r, sp, pj; bool live;
        slot_geo(k, r, sp, pj, live);
        const int po = ((vmask >> k) & 1u) ? pbase + r * d * p.IW + pj : 0;
        ra[k] = ld4(sb + (long)po * S.ld);
      }
    };
    auto store_patch = [&]() {                    // prologue, zero padding, split into planes
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 v = ra[k];
        v.x = fmaf(pa.x, v.x, pb.x); v.y = fmaf(pa.y, v.y, pb.y); v.z = fmaf(pa.z, v.z, pb.z); v.w = fmaf(pa.w, v.w, pb.w);
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        // zero padding and the checkerboard sign in one factor: 0 outside the image, -1 for input pixels of odd (ih + iw), else +1
        const bool ok = pch && ((vmask >> k) & 1u);
        const float sg = ok ? ((((pmask >> k) ^ par0) & 1u) ? -1.f : 1.f) : 0.f;
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
        int r, sp, pj; bool live;
        slot_geo(k, r, sp, pj, live);
        if (r < PR_) {
          uint2 pl[NP];
          split4<NP>(v, pl);
          const int slot = (r * PWP + sp) * 2 + ((q >> 1) ^ ((sp >> 3) & 1));
#pragma unroll
          for (int m = 0; m < NP; ++m) Pl2[(m * PLANE + slot) * 2 + (q & 1)] = pl[m];
        }
      }
    };
    // weight fragments of one tap: NP planes x 16 bytes per lane, the next tap's set is fetched while this tap's MFMAs issue
    auto load_w = [&](int T, uint4* dst) {
      const int Tc = T < nT ? T : nT - 1;
      const uint4* src = wpl + (long)Tc * (WC * NP * 64);
#pragma unroll
      for (int m = 0; m < NP; ++m) dst[m] = src[m * 64];
    };
    // pixel fragment of (tap, tile j): lane reads 16 bytes of pixel (32 j + lane%32 + kw*d) in patch row kh, k half lane/32
    auto read_x = [&](int tap, int j, uint4* x) {
      const int kh = tap / KW_, kw = tap - kh * KW_;
      const uint4* b = Pl + xb[kw];
#pragma unroll
      for (int m = 0; m < NP; ++m) x[m] = b[m * PLANE + ((kh + jrow(j)) * PWP + jcol(j) * 32) * 2];        // the wave's own tile row / first column tile sit in xb[]
    };
    // The accumulation inside the bf16 MFMA is not symmetric: what falls below its internal guard bits is floored, not rounded, so
    // a result sits, on average, 0.17 rms errors BELOW the exact sum whatever the sign of the data (scripts/bf16_bias_probe.hip:
    // mean error -2e-7 at |sum| ~ 1 for K = 2304; same for 32x32x16 and 16x16x32 and any term order; the fp32 MFMA shows
    // none).  Per element that is below the rounding noise, but it is COHERENT: neighbouring pixels all err the same way, and
    // the layers behind (3x3 windows, sums over 10^5-10^6 pixels) respond to such a DC shift far more than to white noise
    // (whole-network frozen-BN gradients at 2x512x1024: 2.5x the error of the exact-fp32 kernels).  The kernel therefore
    // computes a CHECKERBOARD of signs: output pixel (oh, ow) is accumulated as (-1)^(oh+ow) * y.  Input pixels of odd
    // (ih + iw) are negated when they are staged (sign bits of the bf16 planes: exact), which puts (-1)^(oh+ow) * (-1)^((kh+kw) d)
    // on the operand of tap (kh, kw); the second factor is uniform per tap and is baked into the packed weights; the epilogue undoes the
    // pixel's sign.  The floor bias then alternates from pixel to pixel and averages out in every window and every sum.
    auto mma = [&](f32x16& c, const uint4* w, const uint4* x) {
      auto W = [&](int m) { return __builtin_bit_cast(bf16x8, w[m]); };
      auto X = [&](int m) { return __builtin_bit_cast(bf16x8, x[m]); };
      if (NP == 3) {          // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(2), X(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(1), c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(0), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(1), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(0), c, 0, 0, 0);
    };

    // Software pipeline, pinned with sched_barrier (left alone, the compiler sinks every prefetch down to its first use and the
    // wave then waits out a full L2 round trip per tap): weight fragments are fetched TWO taps ahead into a ring of three
    // register sets, pixel fragments one 32-pixel tile ahead, and the next chunk's patch is loaded to registers during tap 0.
    int s = 0, c0 = 0, T0 = 0;
    uint4 wr[3][NP], xr[2][NP];
    load_w(0, wr[0]);
    load_w(1, wr[1]);
    load_patch(0, 0);
    __syncthreads();                 // every wave is done with the previous tile's patch (and red[] is initialised)
    store_patch();
    __syncthreads();
    while (true) {
      int s2 = s, c2 = c0 + C3_BK;
      if (c2 >= p.src[s].C) { c2 = 0; ++s2; }
      const bool more = s2 < p.nsrc;
      read_x(0, 0, xr[0]);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        load_w(T0 + tap + 2, wr[(tap + 2) % 3]);
        if (tap == 0 && more) load_patch(s2, c2);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int nj = (j + 1) % PT, ntap = tap + (j + 1) / PT;
          const int cur = (tap * PT + j) & 1;                       // the two fragment buffers alternate over the (tap, tile) sequence (PT may be odd: 32-pixel tiles)
          if (ntap < TAPS) read_x(ntap, nj, xr[cur ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc[j], wr[tap % 3], xr[cur]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      T0 += TAPS;
      if (TAPS % 3) {                // the two sets fetched ahead sit in ring slots TAPS%3 and (TAPS+1)%3: rotate them to 0 and 1
        uint4 t0[NP], t1[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) { t0[m] = wr[TAPS % 3][m]; t1[m] = wr[(TAPS + 1) % 3][m]; }
#pragma unroll
        for (int m = 0; m < NP; ++m) { wr[0][m] = t0[m]; wr[1][m] = t1[m]; }
      }
      if (BLK) {
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { acc2[j][e] += acc[j][e]; acc[j][e] = 0.f; }
      }
      __syncthreads();
      if (!more) break;
      s = s2; c0 = c2;
      store_patch();
      __syncthreads();
    }
    if (BLK) {
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[j] = acc2[j];
    }

    // ---- epilogue: lane holds pixel (32 j + lane%32), channels n0 + 32 wave + 8 g + 4 (lane/32) + {0..3}, g = 0..3 ----
    const bool want_red = p.slab != nullptr;
    float s1[4][4], s2v[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[g][e] = 0.f; s2v[g][e] = 0.f; }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int jr = wrow + jrow(j);
      const int lp = (wcol + jcol(j)) * 32 + lp32;
      // output pixel: the tile grid's (oh, ow) itself, or (stride-2 data gradient) pixel (om oh + oro, om ow + oco) of the OHo x OWo map
      const long pp = p.om == 1 ? ((long)n * p.H + oh + jr * d) * p.W + ow0 + lp : ((long)n * p.OHo + oh * p.om + p.oro) * p.OWo + (long)(ow0 + lp) * p.om + p.oco;
      const bool pin = ow0 + lp < p.W && oh + jr * d < p.H;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = n0 + wave * 32 + 8 * g + 4 * hh;
        const int nrem = p.Cn - c;
        if (!pin || nrem <= 0) continue;
        float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        if ((par0 + (unsigned)(jr * d + lp32)) & 1u) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }       // undo the checkerboard sign of pixel (oh, ow0 + 32 j + lane%32)
        if (MODE == MODE_FWD) {
          if (p.bias) { float4 b = ld4g(p.bias + c, nrem, false); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
          if (p.bias_n) {
            float4 b = ld4g(p.bias_n + (long)n * p.Cn + c, nrem, false);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          st4g(p.y + pp * p.ldy + c, v, nrem, p.vecY);
          if (want_red) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float f = (e < nrem) ? get4(v, e) : 0.f; s1[g][e] += f; s2v[g][e] += f * f; }
          }
        } else {
          float4 x = ld4g(p.dst.x + pp * p.dst.ld + c, nrem, p.vecY);
          float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
          if (p.dst.a) { av = ld4g(p.dst.a + c, nrem, p.vecY); bv = ld4g(p.dst.b + c, nrem, p.vecY); }
          float4 gq;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = get4(x, e), ae = get4(av, e), be = get4(bv, e), dz = get4(v, e);
            const bool m = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
            set4(gq, e, m ? dz * ae : 0.f);
            if (want_red && m) { s1[g][e] += dz * xe; s2v[g][e] += dz; }
          }
          float* gp = p.y + pp * p.ldy + c;
          if (p.accumulate) { float4 o = ld4g(gp, nrem, p.vecY); gq.x += o.x; gq.y += o.y; gq.z += o.z; gq.w += o.w; }
          st4g(gp, gq, nrem, p.vecY);
        }
      }
    }
    if (want_red) {          // a wave owns its 32 channels alone: butterfly over the 32 pixel lanes, one lane per half adds into red[]
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = s1[g][e], b = s2v[g][e];
#pragma unroll
          for (int m = 1; m < 32; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
          if (lp32 == 0) { double* r = red + ((wpx * BC) + wave * 32 + 8 * g + 4 * hh + e) * 2; r[0] += (double)a; r[1] += (double)b; }
        }
    }
  }
#ifdef ADDK_C3B_DIAG
  if (t == 0) {
    unsigned long long* dslot = g_c3b_diag[(bx + 7 * blockIdx.y) & 63];
    atomicAdd(&dslot[0], __builtin_amdgcn_s_memtime() - diag_c0); atomicAdd(&dslot[1], __builtin_amdgcn_s_memrealtime() - diag_r0);
    atomicAdd(&dslot[2], 1ull);
  }
#endif
  if (p.slab) {
    __syncthreads();
    if (t < BC && n0 + t < p.Cn) {
      double* o = p.slab + ((long)bx * p.slab_ld + n0 + t) * 2;
      o[0] = red[2 * t] + (PH > 1 ? red[2 * (BC + t)] : 0.0); o[1] = red[2 * t + 1] + (PH > 1 ? red[2 * (BC + t) + 1] : 0.0);
    }
  }
}

template <int WC, int KS, int MODE, int NP, bool BIGD, int PH = 1, int BPX = C3_BP, int ST = 1, int TR = 1>
__global__ void __launch_bounds__(64 * WC * PH, 2) conv3b_kernel(const C3K p) {
  conv3b_body<WC, KS, MODE, NP, BIGD, PH, BPX, ST, TR>(p, blockIdx.x, gridDim.x);
}
// the four parity classes of the stride-2 data gradient in ONE launch: workgroups [row0[c], row0[c + 1]) run class c
struct C3K4 { C3K c[4]; int row0[5]; };
template <int NP>
__global__ void __launch_bounds__(256, 2) conv3b_s2d_kernel(const C3K4 q) {
  const int bx = blockIdx.x;
  if (bx < q.row0[1])      conv3b_body<2, 1,  MODE_DGRAD, NP, false, 2>(q.c[0], bx, q.row0[1]);
  else if (bx < q.row0[2]) conv3b_body<2, 12, MODE_DGRAD, NP, false, 2>(q.c[1], bx - q.row0[1], q.row0[2] - q.row0[1]);
  else if (bx < q.row0[3]) conv3b_body<2, 21, MODE_DGRAD, NP, false, 2>(q.c[2], bx - q.row0[2], q.row0[3] - q.row0[2]);
  else                     conv3b_body<2, 22, MODE_DGRAD, NP, false, 2>(q.c[3], bx - q.row0[3], q.row0[4] - q.row0[3]);
}

// 32-row tiles (= waves) per block of the split-bf16 kernel: the count in 3..5 that pads the channel count least (ties: the wider)
int c3b_wc(int Cn, long P) {
  if (Cn <= 64) return 2;                 // two channel tiles; the launch adds the two pixel halves (4 waves, conv3b_kernel PH = 2)
  int best = 4; long bc = -1;
  const int cands[3] = {5, 4, 3};
  for (int i = 0; i < 3; ++i) {
    const long cols = (long)cdiv(Cn, 32 * cands[i]) * 32 * cands[i];
    if (bc < 0 || cols < bc) { bc = cols; best = cands[i]; }
  }
  // 5-wave blocks sit badly on 4 SIMDs (one block per CU at this register count, one SIMD with two waves): 4-wave blocks win
  // up to ~10 % more padded columns (ASPP data gradient, 400 channels: 4 x 128 instead of 3 x 160, 304 -> 200 us)
  if (best == 5 && (long)cdiv(Cn, 128) * 128 * 10 <= bc * 11) best = 4;
  // small maps (160 channels at 32x64): two 3-wave column blocks on half-width tiles instead of one 5-wave block per row
  if (best == 5 && P < 8192) best = 3;
  return best;
}

bool c3_enabled() { return (addk_get_fast_paths() & ADDK_FAST_CONV3) != 0; }
inline bool c3b_pointwise_enabled() { return addk_env("ADDK_C3B_POINTWISE", 1) != 0; }
// bf16 planes of a launch (3: six product terms, 2: three), or 0 = the exact fp32 MFMA kernel.  Every halo launch takes the
// split kernel, the <= 64-channel ones (stem1, the cells' 40-channel dilated convs) on 4-wave blocks = 2 channel tiles x 2
// pixel halves (PH = 2; the first 2-wave form staged 13-22 slots per thread, spilled and was slower than fp32: 125 vs 78 us
// for a 5x5 at 40 channels, now 62).  Accuracy: the network amplifies any 1e-7 perturbation of the stems to 1e-4..1e-3 in the
// whole-network frozen-BatchNorm gradients, so one input draw cannot rank arithmetics — the draw of
// test_add_whole_net_frozen_bn_gradients that once suggested keeping the narrow launches on fp32 (4.9e-4 vs 2.5e-4 median) was
// a sample of that spread: over four more draws at 2x512x1024 split-bf16 everywhere was CLOSER to fp64 than the fp32 kernels
// in all four (median 0.40-0.89x the fp32 oracle's error against 0.84-1.22x; profiles/r02_split_threshold_study.txt).
// addk_set_split_min_channels(n) keeps launches with fewer than n output channels on the fp32 kernel.
int g_c3b_minc = -1;
inline int c3_planes(int Cn, int taps) {
  const int m = addk_get_conv_precision();
  (void)taps;
  if (g_c3b_minc < 0) g_c3b_minc = 0;
  if (m == 0 || Cn < g_c3b_minc) return 0;
  if (m == 3) return Cn >= 192 ? 2 : 3;          // tail_x3: three product terms in the exit heads (ASPP, decoder: 256 / 304 / 400 channels), six elsewhere
  return m == 2 ? 3 : 2;
}
// Column block (16-channel tiles per block).  128-channel blocks (2x2 waves) for the wide heads when that still
// yields >= 512 blocks; otherwise the narrowest of 3/4/5 tiles that pads the channel count least (cells: 40 -> 3 tiles,
// 80 and 160 -> 5 tiles; stem / small ASPP maps: 4 tiles).
int c3_bct(int Cn, long P) {
  const long tiles = (P + C3_BP - 1) / C3_BP;
  if (Cn >= 128 && Cn % 128 == 0 && tiles * (Cn / 128) >= 512) return 8;
  int best = 4; long bc = -1;
  const int cands[3] = {5, 4, 3};
  for (int i = 0; i < 3; ++i) {
    const long cols = (long)cdiv(Cn, 16 * cands[i]) * 16 * cands[i];
    if (bc < 0 || cols < bc) { bc = cols; best = cands[i]; }
  }
  return best;
}
long c3_pack_floats(int Cn, int nchunks, long P, int taps) {
  if (const int np = c3_planes(Cn, taps)) {            // bf16 planes: 1 KB per (tap, 32-row tile, plane)
    const int wc = c3b_wc(Cn, P);
    return (long)cdiv(Cn, 32 * wc) * nchunks * taps * wc * np * 256;
  }
  const int bct = c3_bct(Cn, P);
  return (long)cdiv(Cn, 16 * bct) * nchunks * taps * bct * 256;
}
bool c3_geometry_ok(int KH, int KW, int stride, int pad, int dil, int H, int W, int OH, int OW, long P, int Cn, int ktot = 0, bool fwd = false) {
  if (KH == 1 && KW == 1) {      // wide pointwise heads (ASPP 1x1, the 1280 -> 256 concat conv): the split kernel as a plain GEMM (KS = 1)
    // (the FORWARD of the cells' many-input glue convs, K = 200..800 -> 40..160, was measured on this path in round 3 and is NOT taken: 28.6 vs 29.4 us
    // per launch at 40 output channels, 41.6 vs 25.7 us at 80 — the streaming-K fp32 kernel (pwk_kernel) is memory-bound on those shapes)
    (void)ktot;
    return c3_enabled() && c3b_pointwise_enabled() && stride == 1 && dil == 1 && pad == 0 && OH == H && OW == W && Cn >= 192 && c3_planes(Cn, 1) != 0 &&
           W >= 48 && P >= 2048;
  }
  if (stride == 2) {             // [r3] stem2 (3x3, stride 2, pad 1) forward on the split kernel: 128-channel blocks of 4 waves, de-interleaved patch rows
    if (!(addk_env("ADDK_C3B_STRIDE2", 1) && c3_enabled() && KH == 3 && KW == 3 && dil == 1 && pad == 1 && OH == (H - 1) / 2 + 1 && OW == (W - 1) / 2 + 1 && c3_planes(Cn, 9) != 0 && OW >= 96)) return false;
    if (fwd) return Cn >= 96 && c3b_wc(Cn, (long)P) == 4 && P >= 8192;
    return Cn >= 32 && Cn <= 64 && P >= 32768;      // data gradient: the four parity classes as 2-tile blocks (c3b_s2_dgrad)
  }
  if (!c3_enabled() || KH != KW || !(KH == 3 || KH == 5) || stride != 1 || dil < 1 || dil > c3_maxdil(KH)) return false;
  if (!(pad == dil * (KH / 2) && OH == H && OW == W && Cn >= 32)) return false;
  // the split kernel has half-width (64-pixel) tiles and also takes the 32x64 maps of level 3 (dil_conv at 160 channels: 152 us on
  // the generic kernel); the fp32 halo kernel keeps its 128-pixel tiles and the larger maps
  if (c3_planes(Cn, KH * KW)) return W >= 48 && P >= 2048;
  return W >= 100 && P >= 8192;
}

// stride-2 data gradient: one launch per parity class of the input pixel (each a stride-1 gather over dy with 1 / 2 / 2 / 4 taps and a
// strided scatter of its outputs); the statistics-slab rows (= workgroups) are shared out in proportion to the tap counts
int c3b_s2_dgrad(C3K& k, PackK& pk, int rows, hipStream_t st, bool packed, PackK* desc_out, int np) {
  const int wc = 2;
  pk.bct = wc; pk.mode = MODE_DGRAD; pk.Cn = k.Cn; pk.planes = np; pk.dil_odd = 0; pk.s2d = 1;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }
  if (rows < 9) return 1;
  const long unit = (long)cdiv(k.Cn, 32 * wc) * pk.nchunks * wc * 64;
  if (!packed) { int pb = cdiv(9 * unit, 256); if (pb > 4096) pb = 4096; hipLaunchKernelGGL(c3_pack_kernel, dim3(pb), dim3(256), 0, st, pk); }
  C3K4 q;
  int row0 = 0;
  // share of the workgroups per class: a tile costs (staging + barriers) + taps * MFMA time, not taps alone
  const int share[4] = {3, 4, 4, 6};      // measured at stem2's shape: 0.481 ms against 0.494 for 1:2:2:4
  const int shsum = share[0] + share[1] + share[2] + share[3];
  for (int cls = 0; cls < 4; ++cls) {
    const int pi = cls >> 1, pj = cls & 1, tc = (1 + pi) * (1 + pj), pre = cls == 0 ? 0 : cls == 1 ? 1 : cls == 2 ? 3 : 5;
    const int rc = cls == 3 ? rows - row0 : (rows * share[cls]) / shsum;
    C3K& c = q.c[cls];
    c = k;
    c.H = (k.OHo - pi + 1) / 2; c.W = (k.OWo - pj + 1) / 2;
    c.om = 2; c.oro = pi; c.oco = pj;
    c.nT = pk.nchunks * tc;
    c.wp = pk.out + pre * unit * np * 4;
    c.wp_blk = (long)pk.nchunks * tc * wc * np * 64;
    c.spr = cdiv(c.W, C3_BP);
    c.HT = c.H;
    c.ntiles = c.N * c.H * c.spr;
    c.red32 = 1;
    if (c.slab) c.slab += (long)row0 * c.slab_ld * 2;
    q.row0[cls] = row0;
    row0 += rc;
  }
  q.row0[4] = rows;
  const size_t lds = (size_t)((2 * 32 * wc * 16 + 15) & ~15) + (size_t)np * 2 * cb_pwmax(22, false, C3_BP) * 32;
  dim3 grid(rows, cdiv(k.Cn, 32 * wc));
  if (np == 3) hipLaunchKernelGGL(conv3b_s2d_kernel<3>, grid, dim3(256), lds, st, q);
  else hipLaunchKernelGGL(conv3b_s2d_kernel<2>, grid, dim3(256), lds, st, q);
  return addk_check_launch("conv3b stride-2 data gradient");
}

int c3b_launch(C3K& k, PackK& pk, int mode, int rows, hipStream_t st, bool packed, PackK* desc_out, int np) {
  pk.s2d = 0;
  if (k.st == 2 && mode == MODE_DGRAD) return c3b_s2_dgrad(k, pk, rows, st, packed, desc_out, np);
  const int wc = c3b_wc(k.Cn, k.P);
  pk.bct = wc; pk.mode = mode; pk.Cn = k.Cn; pk.planes = np; pk.dil_odd = k.st == 2 ? 2 : (k.dil & 1);
  k.nT = pk.nchunks * pk.taps;
  k.wp = pk.out;
  k.wp_blk = (long)pk.nchunks * pk.taps * wc * np * 64;          // 16-byte units per column block
  const bool bigd = pk.taps == 9 && k.dil > 2;
  const int ks = pk.taps == 1 ? 1 : pk.taps == 9 ? 3 : 5;
  const int ph = wc == 2 ? 2 : 1;                                  // <= 64 channels: 4 waves = 2 channel tiles x 2 pixel halves
  // half-width tiles where 128-pixel tiles leave the chip short of blocks (instantiated for 3- and 4-wave blocks)
  const long blocks128 = (long)k.N * k.H * cdiv(k.W, C3_BP) * cdiv(k.Cn, 32 * wc);
  const bool half = k.st == 1 && ph == 1 && (wc == 3 || wc == 4) && !(wc == 3 && bigd) && !(wc == 4 && ks == 5) && !(wc == 3 && ks == 1) && blocks128 < 384;      // (rows of <= 64 pixels land here too)
  // quarter-width (32-pixel) tiles where even 64-pixel tiles leave the chip short of workgroups: the cells' dilated convs on the 32x64 maps ran as
  // 128 workgroups of 3 waves — half the CUs idle, one wave per SIMD on the others, every LDS / weight round trip exposed
  const long blocks64 = (long)k.N * k.H * cdiv(k.W, 64) * cdiv(k.Cn, 32 * wc);
  const bool quarter = half && wc == 3 && !bigd && (ks == 3 || ks == 5) && blocks64 < 192;      // measured: 160 ch @ 32x64 85 -> 63 us (5x5), 45 -> 34 (3x3); 80 ch @ 63x127 (252 blocks) is slower quartered (53 -> 66)
  const int bpx = k.st == 2 ? 64 : quarter ? 32 : half ? 64 : C3_BP;
  // two-row tiles (2 rows, d apart, of half the one-row width) for the 3x3 / 5x5 launches at dilation <= 2: KS + 1 staged rows per two output rows
  const int tworow_on = addk_env("ADDK_C3B_TWOROW", 2);      // 0: one-row tiles; 1: dilation-1 3x3 only (decoder, stem1); 2: all
  const bool tr_shape = k.st == 1 && (ks == 3 || ks == 5) && !bigd && k.om == 1 && (bpx == C3_BP || (bpx == 64 && wc == 3)) && k.H >= 2 * k.dil;
  const bool tworow = tr_shape && (tworow_on >= 2 || (tworow_on == 1 && ks == 3 && k.dil == 1 && (wc == 4 || wc == 2)));
  const int rpx = tworow ? bpx / 2 : bpx;                       // pixels per tile row
  k.HT = tworow ? cdiv(k.H, 2 * k.dil) * k.dil : k.H;
  k.spr = cdiv(k.W, rpx);
  k.ntiles = k.N * k.HT * k.spr;
  k.red32 = 1;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }
  const long total = (long)cdiv(k.Cn, 32 * wc) * k.nT * wc * 64;   // pack threads: one per (tile, lane)
  int pb = cdiv(total, 256); if (pb > 4096) pb = 4096;
  if (!packed) hipLaunchKernelGGL(c3_pack_kernel, dim3(pb), dim3(256), 0, st, pk);
  const size_t lds = (size_t)((ph * 32 * wc * 16 + 15) & ~15) + (size_t)np * (tworow ? ks + 1 : ks) * cb_pwmax(ks, bigd, rpx, k.st) * 32;
  dim3 grid(rows, cdiv(k.Cn, 32 * wc));
  bool done = false;
#define ADDK_C3B_(W_, K_, M_, P_, D_, X_) { \
    constexpr int H_ = W_ == 2 ? 2 : 1; \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3b_kernel<W_, K_, M_, P_, D_, H_, X_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL((conv3b_kernel<W_, K_, M_, P_, D_, H_, X_>), grid, dim3(64 * W_ * H_), lds, st, k); done = true; }
#define ADDK_C3BX(W_, K_, D_, X_) \
  if (!done && wc == W_ && ks == K_ && bigd == D_ && bpx == X_) { \
    if (mode == MODE_FWD) { if (np == 3) ADDK_C3B_(W_, K_, MODE_FWD, 3, D_, X_) else ADDK_C3B_(W_, K_, MODE_FWD, 2, D_, X_) } \
    else { if (np == 3) ADDK_C3B_(W_, K_, MODE_DGRAD, 3, D_, X_) else ADDK_C3B_(W_, K_, MODE_DGRAD, 2, D_, X_) } }
#define ADDK_C3B(W_, K_, D_) ADDK_C3BX(W_, K_, D_, C3_BP)
  if (tworow) {
#define ADDK_C3TR(W_, K_, X_, M_, P_) { \
      constexpr int H_ = W_ == 2 ? 2 : 1; \
      static bool attr = false; \
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3b_kernel<W_, K_, M_, P_, false, H_, X_, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
      hipLaunchKernelGGL((conv3b_kernel<W_, K_, M_, P_, false, H_, X_, 1, 2>), grid, dim3(64 * W_ * H_), lds, st, k); done = true; }
#define ADDK_C3TRW(W_, K_, X_) if (!done && wc == W_ && ks == K_ && rpx == X_) { \
      if (mode == MODE_FWD) { if (np == 3) ADDK_C3TR(W_, K_, X_, MODE_FWD, 3) else ADDK_C3TR(W_, K_, X_, MODE_FWD, 2) } \
      else { if (np == 3) ADDK_C3TR(W_, K_, X_, MODE_DGRAD, 3) else ADDK_C3TR(W_, K_, X_, MODE_DGRAD, 2) } }
    ADDK_C3TRW(4, 3, 64) ADDK_C3TRW(2, 3, 64) ADDK_C3TRW(3, 3, 64) ADDK_C3TRW(5, 3, 64)
    ADDK_C3TRW(2, 5, 64) ADDK_C3TRW(3, 5, 64) ADDK_C3TRW(4, 5, 64) ADDK_C3TRW(5, 5, 64)
    ADDK_C3TRW(3, 3, 32) ADDK_C3TRW(3, 5, 32)
#undef ADDK_C3TRW
#undef ADDK_C3TR
  } else if (k.st == 2) {               // stem2 forward
    if (wc == 4 && ks == 3 && mode == MODE_FWD) {
#define ADDK_C3S2(P_, X_) { \
      static bool attr = false; \
      auto fn = &conv3b_kernel<4, 3, MODE_FWD, P_, false, 1, X_, 2>; \
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
      hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, k); done = true; }
      if (bpx == 64) { if (np == 3) ADDK_C3S2(3, 64) else ADDK_C3S2(2, 64) }
      else { if (np == 3) ADDK_C3S2(3, C3_BP) else ADDK_C3S2(2, C3_BP) }
#undef ADDK_C3S2
    }
  } else {
  ADDK_C3B(2, 3, false) ADDK_C3B(3, 3, false) ADDK_C3B(4, 3, false) ADDK_C3B(5, 3, false)
  ADDK_C3B(3, 3, true) ADDK_C3B(4, 3, true) ADDK_C3B(5, 3, true) ADDK_C3B(2, 3, true)
  ADDK_C3B(2, 5, false) ADDK_C3B(3, 5, false) ADDK_C3B(4, 5, false) ADDK_C3B(5, 5, false)
  ADDK_C3BX(3, 3, false, 64) ADDK_C3BX(3, 5, false, 64) ADDK_C3BX(4, 3, false, 64) ADDK_C3BX(4, 3, true, 64)
  ADDK_C3BX(3, 3, false, 32) ADDK_C3BX(3, 5, false, 32)
  ADDK_C3B(4, 1, false) ADDK_C3B(3, 1, false) ADDK_C3B(5, 1, false) ADDK_C3BX(4, 1, false, 64) ADDK_C3B(2, 1, false)
  }
#undef ADDK_C3B
#undef ADDK_C3BX
#undef ADDK_C3B_
  if (!done) { addk_set_error("conv3b: no instantiation for %d waves, %d taps", wc, pk.taps); return ADDK_ERR_UNSUPPORTED; }
  return addk_check_launch("conv3b");
}

int c3_launch(C3K& k, PackK& pk, int mode, int rows, hipStream_t st, bool packed, PackK* desc_out = nullptr) {
  pk.planes = 0; pk.dil_odd = 0; pk.s2d = 0;
  if (const int np = c3_planes(k.Cn, pk.taps)) return c3b_launch(k, pk, mode, rows, st, packed, desc_out, np);
  const int bct = c3_bct(k.Cn, k.P);
  pk.bct = bct; pk.mode = mode; pk.Cn = k.Cn;
  k.nT = pk.nchunks * pk.taps;
  k.wp = pk.out;
  k.wp_blk = (long)pk.nchunks * pk.taps * bct * 256;
  k.spr = cdiv(k.W, C3_BP);
  k.ntiles = k.N * k.H * k.spr;
  k.red32 = k.P >= 4096;
  const long total = c3_pack_floats(k.Cn, pk.nchunks, k.P, pk.taps);
  int pb = cdiv(total, 256 * 4); if (pb > 4096) pb = 4096;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }              // descriptor query only (addk_conv_*_pack_desc)
  if (!packed) hipLaunchKernelGGL(c3_pack_kernel, dim3(pb), dim3(256), 0, st, pk);
  dim3 grid(rows, cdiv(k.Cn, 16 * bct));
  bool done = false;
#define ADDK_C3(B_, K_) \
  if (!done && bct == B_ && pk.taps == K_ * K_) { \
    if (mode == MODE_FWD) hipLaunchKernelGGL((conv3_kernel<B_, K_, MODE_FWD>), grid, dim3(256), 0, st, k); \
    else hipLaunchKernelGGL((conv3_kernel<B_, K_, MODE_DGRAD>), grid, dim3(256), 0, st, k); \
    done = true; }
  ADDK_C3(3, 3) ADDK_C3(4, 3) ADDK_C3(5, 3) ADDK_C3(8, 3) ADDK_C3(3, 5) ADDK_C3(4, 5) ADDK_C3(5, 5)
#undef ADDK_C3
  if (!done) { addk_set_error("conv3: no instantiation for %d column tiles, %d taps", bct, pk.taps); return ADDK_ERR_UNSUPPORTED; }
  return addk_check_launch("conv3");
}

}  // namespace

// number of floats the packed-weight workspace of this launch needs; 0 = the halo-patch kernel does not apply
extern "C" int64_t addk_conv_fwd_pack_floats(const addk_conv_args* a) {
  if (!a || a->nsrc < 1 || a->nsrc > ADDK_MAX_SRC) return 0;
  int ktot = 0;
  for (int i = 0; i < a->nsrc; ++i) ktot += a->src[i].C;
  if (!c3_geometry_ok(a->KH, a->KW, a->stride, a->pad, a->dil, a->H, a->W, a->OH, a->OW, (long)a->N * a->OH * a->OW, a->Cout, ktot, true)) return 0;
  int nch = 0;
  for (int i = 0; i < a->nsrc; ++i) { if (a->src[i].C % 4 || a->src[i].ld % 4) return 0; nch += cdiv(a->src[i].C, C3_BK); }
  if (nch > C3_MAXCH || a->ldy % 4) return 0;
  return c3_pack_floats(a->Cout, nch, (long)a->N * a->OH * a->OW, a->KH * a->KW);
}
extern "C" int64_t addk_conv_dgrad_pack_floats(const addk_conv_dgrad_args* a) {
  if (!a) return 0;
  if (!c3_geometry_ok(a->KH, a->KW, a->stride, a->pad, a->dil, a->H, a->W, a->OH, a->OW, (long)a->N * a->H * a->W, a->dst.C)) return 0;
  if (a->Cout % 4 || a->lddy % 4 || a->ldg % 4 || a->dst.ld % 4 || a->dst.C % 4) return 0;
  const int nch = cdiv(a->Cout, C3_BK);
  if (nch > C3_MAXCH) return 0;
  return c3_pack_floats(a->dst.C, nch, (long)a->N * a->H * a->W, a->KH * a->KW);
}

// 0 = launched, 1 = not covered (caller falls back to the generic kernel), <0 = error
static int c3_fwd(const addk_conv_args* a, int rows, void* stream, PackK* desc_out) {
  const int64_t need = addk_conv_fwd_pack_floats(a);
  if (need == 0 || !a->wpack || a->wpack_floats < need) return 1;
  if (!aligned16(a->y) || !aligned16(a->wpack)) return 1;
  for (int i = 0; i < a->nsrc; ++i) if (!src_vec_ok(a->src[i])) return 1;
  C3K k; PackK pk;
  k.nsrc = a->nsrc;
  int nch = 0, choff = 0;
  for (int i = 0; i < a->nsrc; ++i) {
    k.src[i] = a->src[i];
    for (int c0 = 0; c0 < a->src[i].C; c0 += C3_BK) {
      pk.cbase[nch] = a->w_choff + choff + c0;
      pk.cvalid[nch] = a->src[i].C - c0 < C3_BK ? a->src[i].C - c0 : C3_BK;
      ++nch;
    }
    choff += a->src[i].C;
  }
  pk.nchunks = nch; pk.taps = a->KH * a->KW; pk.w = a->w; pk.ldw = a->ldw; pk.cin_total = a->cin_total; pk.w_choff = 0; pk.out = a->wpack;
  k.N = a->N; k.H = a->OH; k.W = a->OW; k.IH = a->H; k.IW = a->W; k.dil = a->dil; k.st = a->stride;
  k.om = 1; k.oro = k.oco = 0; k.OHo = a->OH; k.OWo = a->OW;
  k.Cn = a->Cout; k.ldy = a->ldy; k.y = a->y; k.bias = a->bias; k.bias_n = a->bias_n;
  k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.dst = addk_src{nullptr, nullptr, nullptr, 0, 0, 0, 0}; k.accumulate = 0;
  k.vecY = 1;
  k.P = (long)a->N * a->OH * a->OW;
  return c3_launch(k, pk, MODE_FWD, rows, (hipStream_t)stream, a->wpack_ready != 0, desc_out);
}

static int c3_dgrad(const addk_conv_dgrad_args* a, int rows, void* stream, PackK* desc_out) {
  const int64_t need = addk_conv_dgrad_pack_floats(a);
  if (need == 0 || !a->wpack || a->wpack_floats < need) return 1;
  if (!aligned16(a->dy) || !aligned16(a->g) || !aligned16(a->wpack) || !src_vec_ok(a->dst)) return 1;
  C3K k; PackK pk;
  k.nsrc = 1;
  k.src[0] = addk_src{a->dy, nullptr, nullptr, a->lddy, a->Cout, 0, 0};
  int nch = 0;
  for (int c0 = 0; c0 < a->Cout; c0 += C3_BK) {
    pk.cbase[nch] = c0;
    pk.cvalid[nch] = a->Cout - c0 < C3_BK ? a->Cout - c0 : C3_BK;
    ++nch;
  }
  pk.nchunks = nch; pk.taps = a->KH * a->KW; pk.w = a->w; pk.ldw = a->ldw; pk.cin_total = a->cin_total; pk.w_choff = a->w_choff; pk.out = a->wpack;
  k.N = a->N; k.H = a->H; k.W = a->W; k.IH = a->OH; k.IW = a->OW; k.dil = a->dil; k.st = a->stride;      // the gather runs over dy (OH x OW)
  k.om = 1; k.oro = k.oco = 0; k.OHo = a->H; k.OWo = a->W;
  k.Cn = a->dst.C; k.ldy = a->ldg; k.y = a->g; k.bias = nullptr; k.bias_n = nullptr;
  k.slab = (double*)a->dab; k.slab_ld = a->dst.C;
  k.dst = a->dst; k.accumulate = a->accumulate;
  k.vecY = 1;
  k.P = (long)a->N * a->H * a->W;
  return c3_launch(k, pk, MODE_DGRAD, rows, (hipStream_t)stream, a->wpack_ready != 0, desc_out);
}

#ifdef ADDK_C3B_DIAG
// (shader ticks, reference ticks, workgroups, 0) summed since the last call; resets the counters
extern "C" int addk_c3b_diag(unsigned long long* out4) {
  unsigned long long h[64][4];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3b_diag), sizeof h) != hipSuccess) return ADDK_ERR_INVALID;
  for (int k = 0; k < 4; ++k) { out4[k] = 0; for (int i = 0; i < 64; ++i) out4[k] += h[i][k]; }
  memset(h, 0, sizeof h);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_c3b_diag), h, sizeof h) == hipSuccess ? ADDK_OK : ADDK_ERR_INVALID;
}
#endif
extern "C" int addk_set_split_min_channels(int c) { g_c3b_minc = c < 0 ? 0 : c; return ADDK_OK; }
int addk_c3_try_fwd(const addk_conv_args* a, int rows, void* stream) { return c3_fwd(a, rows, stream, nullptr); }
int addk_c3_try_dgrad(const addk_conv_dgrad_args* a, int rows, void* stream) { return c3_dgrad(a, rows, stream, nullptr); }

// Hoisting the weight packs out of the step's critical path: a plan collects one descriptor per halo-patch launch,
// uploads the table and runs ONE addk_conv_pack_batch at the start of the step; the launches then carry wpack_ready = 1.
extern "C" int64_t addk_conv_pack_desc_bytes(void) { return (int64_t)sizeof(PackK); }
extern "C" int addk_conv_fwd_pack_desc(const addk_conv_args* a, void* host_desc) {
  ADDK_REQUIRE(a && host_desc, "conv_fwd_pack_desc: null pointer");
  const int r = c3_fwd(a, 1, nullptr, reinterpret_cast<PackK*>(host_desc));
  if (r == 1) { addk_set_error("conv_fwd_pack_desc: the halo-patch kernel does not cover this launch"); return ADDK_ERR_UNSUPPORTED; }
  return r;
}
extern "C" int addk_conv_dgrad_pack_desc(const addk_conv_dgrad_args* a, void* host_desc) {
  ADDK_REQUIRE(a && host_desc, "conv_dgrad_pack_desc: null pointer");
  const int r = c3_dgrad(a, 1, nullptr, reinterpret_cast<PackK*>(host_desc));
  if (r == 1) { addk_set_error("conv_dgrad_pack_desc: the halo-patch kernel does not cover this launch"); return ADDK_ERR_UNSUPPORTED; }
  return r;
}
extern "C" int addk_conv_pack_batch(const void* dev_descs, int32_t n, void* stream) {
  ADDK_REQUIRE(dev_descs && n > 0, "conv_pack_batch: bad args");
  hipLaunchKernelGGL(c3_pack_batch_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackK*>(dev_descs));
  return addk_check_launch("conv_pack_batch");
}
