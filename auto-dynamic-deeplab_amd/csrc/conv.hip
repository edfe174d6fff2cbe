// Dense convolution forward and data-gradient as one implicit-GEMM kernel on the fp32 matrix
// cores of gfx950 (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, 64-wide wavefronts).
//
//   out[p, n] = sum_{tap, c}  Z[p @ tap, c] * Wt[n][tap][c]
//
// M = pixels (tile of 64*PT, 16*PT per wave), N = output channels (tile of 16*CT, shared by the
// four waves), K = taps x channels walked in chunks of 32 channels of one tap of one source.
// The MFMA "A" operand is the WEIGHT fragment and the "B" operand the PIXEL fragment, so a
// lane's four accumulator registers are four consecutive output channels of one pixel and the
// epilogue stores 16 bytes per lane into the NHWC destination.
// Input tiles are staged through LDS with the producer's BatchNorm/ReLU applied on the way in
// (zero padding is applied after that prologue, as in the reference where the conv pads its
// already-activated input).  Rows are padded to 36 floats: ds_read_b64 fragment reads are
// bank-conflict free (row*36 mod 64 walks the 16 multiples of 4).
//
// Reference call sites: see include/addk.h (addk_conv_fwd / addk_conv_dgrad).
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct ConvK {
  addk_src src[ADDK_MAX_SRC];
  int nsrc;
  int N, H, W;      // A-side tensor spatial size (fwd: input; dgrad: dy)
  int OH, OW;       // M-side pixel grid        (fwd: output; dgrad: input)
  int KH, KW, stride, pad, dil;
  int Cn;           // GEMM N (fwd: Cout; dgrad: channels of dst)
  int ldw, cin_total, w_choff;
  int ldy;
  const float* w;
  float* y;
  const float* bias;
  const float* bias_n;
  double* slab; int slab_ld;
  addk_src dst;     // dgrad epilogue
  int accumulate;
  int vecA, vecB, vecY;
  int red32;        // 1: per-tile lane reduction in fp32 (large P); 0: fp64 end to end (tiny batches, e.g. the 2-sample ASPP pool BN)
  long P;
  int ntiles;
  // strided dgrad, one launch per input-pixel parity class: M-side pixels are (sub*i+ph, sub*j+pw) on an MH x MW grid
  // and only the taps of `taplist` can reach them (kill: the class has no tap at all -> zero gradient)
  int sub, ph, pw, MH, MW, ntaps_l, kill;
  int taplist[9];
  // ... or ALL classes in ONE launch (ncls > 0): workgroups [gx0, next gx0) of the x grid belong to class c and use cl[c]
  // in place of the single-class fields above (P, ntiles, slab included)
  int ncls;
  struct Cls { long P; double* slab; int ph, pw, MH, MW, ntaps_l, kill, ntiles, gx0; int taplist[9]; int pad_; } cl[4];
};

constexpr int BK = 32;
constexpr int BKP = 36;

// PREC_F32: exact fp32 products on v_mfma_f32_16x16x4_f32.
// PREC_B3:  every fp32 operand x is split at LDS-staging time into bf16 hi = rn(x), lo = rn(x - hi) and a product is
//           evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped lo*lo term
//           is <= 2^-16 relative): 3 bf16 MFMAs at 16x the fp32 matrix rate = 5.3x the exact path, error ~1e-5 relative
//           per product, two orders below the 1e-3 parity bound.  Rows are 32 bf16 + 16 pad (96 B): ds_read_b128
//           fragment reads of a 16-lane group then fall on 16 distinct 16-B slots of the 256-B bank row.
enum { PREC_F32 = 0, PREC_B3 = 1 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int RS = 48;      // bf16 elements per LDS row in the split path

__device__ __forceinline__ void split4(float4 v, bf16x4& h, bf16x4& l) {
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
  l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
}

template <int PT, int CT, int MODE, int PREC>
__global__ void __launch_bounds__(256) conv_kernel(const ConvK p) {
  constexpr int BP = 64 * PT;
  constexpr int BC = 16 * CT;
  constexpr int NAJ = 2 * PT;             // float4 A slots per thread
  constexpr int NBJ = (CT + 1) / 2;       // float4 B slots per thread
  // one LDS arena: fp32 path [BP+BC][36] floats; split path hi and lo panels of [BP+BC][48] bf16 (same bytes per row pair)
  constexpr int LDS_BYTES = PREC == PREC_F32 ? (BP + BC) * BKP * 4 : (BP + BC) * RS * 2 * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  float* As = reinterpret_cast<float*>(lds_raw);
  float* Bs = As + BP * BKP;
  __bf16* Ah = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Al = Ah + BP * RS;
  __bf16* Bh = Al + BP * RS;
  __bf16* Bl = Bh + BC * RS;
  __shared__ double red[4][BC][2];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.y * BC;
  const int aq = t & 7, ar = t >> 3;
  const bool cls = MODE == MODE_DGRAD && p.sub > 1;
  // this workgroup's parity class (wave-uniform): single-class launch -> the fields of p; combined launch -> cl[c]
  int bx = blockIdx.x, gx = gridDim.x;
  int c_ph = p.ph, c_pw = p.pw, c_MH = p.MH, c_MW = p.MW, c_ntaps = p.ntaps_l, c_kill = p.kill, c_ntiles = p.ntiles, c_idx = 0;
  long c_P = p.P;
  double* c_slab = p.slab;
  if (MODE == MODE_DGRAD && p.ncls > 0) {
    int c = 0;
    while (c + 1 < p.ncls && bx >= p.cl[c + 1].gx0) ++c;
    gx = (c + 1 < p.ncls ? p.cl[c + 1].gx0 : (int)gridDim.x) - p.cl[c].gx0;
    bx -= p.cl[c].gx0;
    c_ph = p.cl[c].ph; c_pw = p.cl[c].pw; c_MH = p.cl[c].MH; c_MW = p.cl[c].MW; c_ntaps = p.cl[c].ntaps_l; c_kill = p.cl[c].kill;
    c_ntiles = p.cl[c].ntiles; c_P = p.cl[c].P; c_slab = p.cl[c].slab; c_idx = c;
  }
  const int ntaps = cls ? c_ntaps : p.KH * p.KW;
  const int ohw = cls ? c_MH * c_MW : p.OH * p.OW;
  const int mw = cls ? c_MW : p.OW;
  const int P32 = (int)c_P;
  double tot0 = 0.0, tot1 = 0.0;
  // per-slot weight offsets (fixed for the whole kernel); -1 marks a slot outside the tile
  long woff[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int slot = t + 256 * j;
    if (MODE == MODE_FWD) {
      const int row = slot >> 3, q = slot & 7, co = n0 + row;
      woff[j] = (row < BC && co < p.Cn) ? (long)co * p.ldw + 4 * q : -1;
    } else {
      const int k = slot / (4 * CT), ng = slot - k * (4 * CT), ci = n0 + 4 * ng;
      woff[j] = (k < BK && ci < p.Cn) ? (long)k * p.ldw + ci : -1;
    }
  }

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so workgroup b
  // takes tile (b % 8) * (ntiles / 8) + b / 8: every XCD walks a contiguous range of pixel tiles and the rows a k x k
  // tap re-reads (one image row above / below = a few tiles away) are served from that XCD's L2.  Speed only.
  const int tpx = c_ntiles >> 3;
  const bool swz = (c_ntiles & 7) == 0 && c_ntiles >= 64;
  for (int tlin = bx; tlin < c_ntiles; tlin += gx) {
    const int tile = swz ? (tlin & 7) * tpx + (tlin >> 3) : tlin;
    int rn[NAJ], rh[NAJ], rw[NAJ];
    long roff[NAJ];              // element offset of (n, rh, rw) in a source with pixel stride 1 (scaled by S.ld per chunk)
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      int pp = tile * BP + ar + 32 * j;
      if (pp < P32) {
        int n = pp / ohw;
        int rem = pp - n * ohw;
        int oh = rem / mw, ow = rem - oh * mw;
        if (cls) { oh = oh * p.sub + c_ph; ow = ow * p.sub + c_pw; }
        rn[j] = n;
        if (MODE == MODE_FWD) { rh[j] = oh * p.stride - p.pad; rw[j] = ow * p.stride - p.pad; }
        else                  { rh[j] = oh + p.pad;            rw[j] = ow + p.pad; }
        roff[j] = ((long)n * p.H + rh[j]) * p.W + rw[j];
      } else { rn[j] = -1; rh[j] = 0; rw[j] = 0; roff[j] = 0; }
    }

    // Blocked accumulation: `acc` is the running sum of ONE chunk (a tap x 32 channels: 32 chained fp32 FMAs per output on the
    // fp32 MFMA, which is bitwise a k-ordered fmaf chain); it is flushed into `tot` after every chunk, so the rounding error of a
    // K = taps x channels reduction grows with sqrt(32) + sqrt(chunks) instead of sqrt(K) — stem2 (3x3, 64 channels): 9.9 vs 24.
    // The CPU library the reference runs on blocks its sums the same way (even-size gradient deficit).
    f32x4 acc[CT][PT], tot[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    float4 ra[NAJ], rb[NBJ];
    unsigned amask = 0;          // bit j: ra[j] holds an in-bounds pixel (the lazy prologue applies; padding stays 0)
    int s = 0, tap = 0, c0 = 0, choff = 0;

    auto load_chunk = [&](int s_, int tapi_, int c0_, int choff_) {
      const addk_src S = p.src[s_];
      const int tap_ = cls ? (p.ncls > 0 ? p.cl[c_idx].taplist[tapi_] : p.taplist[tapi_]) : tapi_;
      const int kh = tap_ / p.KW, kw = tap_ - kh * p.KW;
      const int c = c0_ + 4 * aq;
      const int nrem = S.C - c;          // valid channels from c on
      amask = 0;
      const int dh = kh * p.dil, dw = kw * p.dil;
      // wave-uniform pixel offset of this tap relative to the row base (fwd: +tap, stride-1 dgrad: -tap)
      const long tapoff = (MODE == MODE_FWD) ? ((long)dh * p.W + dw) : -((long)dh * p.W + dw);
      const float* sbase = S.x + c;
#pragma unroll
      for (int j = 0; j < NAJ; ++j) {
        float4 v = zero4();
        if (rn[j] >= 0 && nrem > 0) {
          bool ok; long poff;
          if (MODE == MODE_FWD) {
            const int ih = rh[j] + dh, iw = rw[j] + dw;
            ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            poff = roff[j] + tapoff;
          } else if (p.stride == 1) {
            const int ih = rh[j] - dh, iw = rw[j] - dw;
            ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            poff = roff[j] + tapoff;
          } else {
            const int th = rh[j] - dh, tw = rw[j] - dw;
            ok = th >= 0 && tw >= 0 && (th % p.stride == 0) && (tw % p.stride == 0) && !(cls && c_kill);
            const int ih = th / p.stride, iw = tw / p.stride;
            ok = ok && ih < p.H && iw < p.W;
            poff = ((long)rn[j] * p.H + ih) * p.W + iw;
          }
          if (ok) {
            v = ld4g(sbase + poff * S.ld, nrem, p.vecA);
            amask |= 1u << j;
          }
        }
        ra[j] = v;
      }
      if (MODE == MODE_FWD) {
        const float* wb = p.w + ((long)tap_ * p.cin_total + p.w_choff + choff_ + c0_);      // wave-uniform
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          const int cc = c0_ + 4 * ((t + 256 * j) & 7);
          float4 v = zero4();
          if (woff[j] >= 0 && cc < S.C) v = ld4g(wb + woff[j], S.C - cc, p.vecB);
          rb[j] = v;
        }
      } else {
        const float* wb = p.w + ((long)c0_ * p.ldw + (long)tap_ * p.cin_total + p.w_choff);  // wave-uniform
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          const int slot = t + 256 * j, k = slot / (4 * CT), ng = slot - k * (4 * CT);
          float4 v = zero4();
          if (woff[j] >= 0 && c0_ + k < S.C) v = ld4g(wb + woff[j], p.Cn - (n0 + 4 * ng), p.vecB);
          rb[j] = v;
        }
      }
    };
    auto store_chunk = [&](int s_, int c0_) {
      {   // lazy prologue of the producer's BatchNorm/ReLU, applied here (after the MFMAs of the previous chunk)
        const addk_src S = p.src[s_];
        if (S.a || S.relu) {
          const int c = c0_ + 4 * aq;
          const int nrem = S.C - c;
          if (nrem > 0) {
            float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
            if (S.a) { av = ld4g(S.a + c, nrem, p.vecA); bv = ld4g(S.b + c, nrem, p.vecA); }
#pragma unroll
            for (int j = 0; j < NAJ; ++j) {
              if (amask & (1u << j)) {
                float4 v = ra[j];
                v.x = fmaf(av.x, v.x, bv.x); v.y = fmaf(av.y, v.y, bv.y); v.z = fmaf(av.z, v.z, bv.z); v.w = fmaf(av.w, v.w, bv.w);
                if (S.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (nrem < 4) { if (nrem < 2) v.y = 0.f; if (nrem < 3) v.z = 0.f; v.w = 0.f; }
                ra[j] = v;
              }
            }
          }
        }
      }
      if (PREC == PREC_B3) {
#pragma unroll
        for (int j = 0; j < NAJ; ++j) {
          bf16x4 h, l; split4(ra[j], h, l);
          *reinterpret_cast<bf16x4*>(&Ah[(ar + 32 * j) * RS + 4 * aq]) = h;
          *reinterpret_cast<bf16x4*>(&Al[(ar + 32 * j) * RS + 4 * aq]) = l;
        }
        if (MODE == MODE_FWD) {
#pragma unroll
          for (int j = 0; j < NBJ; ++j) {
            int slot = t + 256 * j, row = slot >> 3, q = slot & 7;
            if (row < BC) {
              bf16x4 h, l; split4(rb[j], h, l);
              *reinterpret_cast<bf16x4*>(&Bh[row * RS + 4 * q]) = h;
              *reinterpret_cast<bf16x4*>(&Bl[row * RS + 4 * q]) = l;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < NBJ; ++j) {
            int slot = t + 256 * j, k = slot / (4 * CT), ng = slot - k * (4 * CT);
            if (k < BK) {
              bf16x4 h, l; split4(rb[j], h, l);
#pragma unroll
              for (int e = 0; e < 4; ++e) { Bh[(4 * ng + e) * RS + k] = h[e]; Bl[(4 * ng + e) * RS + k] = l[e]; }
            }
          }
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < NAJ; ++j) lds_st4(&As[(ar + 32 * j) * BKP + 4 * aq], ra[j]);
      if (MODE == MODE_FWD) {
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          int slot = t + 256 * j, row = slot >> 3, q = slot & 7;
          if (row < BC) lds_st4(&Bs[row * BKP + 4 * q], rb[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          int slot = t + 256 * j, k = slot / (4 * CT), ng = slot - k * (4 * CT);
          if (k < BK) {
            Bs[(4 * ng + 0) * BKP + k] = rb[j].x;
            Bs[(4 * ng + 1) * BKP + k] = rb[j].y;
            Bs[(4 * ng + 2) * BKP + k] = rb[j].z;
            Bs[(4 * ng + 3) * BKP + k] = rb[j].w;
          }
        }
      }
    };

    load_chunk(s, tap, c0, choff);
    store_chunk(s, c0);
    __syncthreads();
    while (true) {
      // current chunk extent, then advance the iterator
      const int Cs = p.src[s].C;
      const int kc = min(BK, Cs - c0);
      const int nu = (kc + 7) >> 3;
      int s2 = s, tap2 = tap, c2 = c0 + BK, ch2 = choff;
      if (c2 >= Cs) { c2 = 0; ++tap2; if (tap2 == ntaps) { tap2 = 0; ch2 += Cs; ++s2; } }
      const bool more = s2 < p.nsrc;
      if (more) load_chunk(s2, tap2, c2, ch2);

      if (PREC == PREC_B3) {
        bf16x8 wh[CT], wl[CT], xh[PT], xl[PT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          wh[i] = *reinterpret_cast<const bf16x8*>(&Bh[(i * 16 + li) * RS + 8 * kq]);
          wl[i] = *reinterpret_cast<const bf16x8*>(&Bl[(i * 16 + li) * RS + 8 * kq]);
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          xh[j] = *reinterpret_cast<const bf16x8*>(&Ah[((wave * PT + j) * 16 + li) * RS + 8 * kq]);
          xl[j] = *reinterpret_cast<const bf16x8*>(&Al[((wave * PT + j) * 16 + li) * RS + 8 * kq]);
        }
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < PT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh[j], acc[i][j], 0, 0, 0);
          }
      } else
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < nu) {
          float2 wf[CT], xf[PT];
#pragma unroll
          for (int i = 0; i < CT; ++i)
            wf[i] = *reinterpret_cast<const float2*>(&Bs[(i * 16 + li) * BKP + 8 * u + 2 * kq]);
#pragma unroll
          for (int j = 0; j < PT; ++j)
            xf[j] = *reinterpret_cast<const float2*>(&As[((wave * PT + j) * 16 + li) * BKP + 8 * u + 2 * kq]);
#pragma unroll
          for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].x, xf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].y, xf[j].y, acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      __syncthreads();
      if (!more) break;
      s = s2; tap = tap2; c0 = c2; choff = ch2;
      store_chunk(s, c0);
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[i][j] = tot[i][j];

    // ---- epilogue ----
    const bool want_red = c_slab != nullptr;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int c = n0 + i * 16 + kq * 4;
      const int nrem = p.Cn - c;
      double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        int pp = tile * BP + (wave * PT + j) * 16 + li;
        const bool pv = pp < P32 && nrem > 0;
        if (cls && pv) {      // class-local pixel -> position in the full gradient map
          const int n = pp / ohw, rem = pp - n * ohw, i2 = rem / mw, j2 = rem - i2 * mw;
          pp = (n * p.OH + i2 * p.sub + c_ph) * p.OW + j2 * p.sub + c_pw;
        }
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (MODE == MODE_FWD) {
          if (pv) {
            if (p.bias) { float4 b = ld4g(p.bias + c, nrem, false); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.bias_n) {
              int n = pp / ohw;
              float4 b = ld4g(p.bias_n + (long)n * p.Cn + c, nrem, false);
              v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            st4g(p.y + (long)pp * p.ldy + c, v, nrem, p.vecY);
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
            float4 x = ld4g(p.dst.x + (long)pp * p.dst.ld + c, nrem, p.vecY);
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
            float* gp = p.y + (long)pp * p.ldy + c;
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
            if (li == 0) { red[wave][i * 16 + kq * 4 + e][0] = a; red[wave][i * 16 + kq * 4 + e][1] = b; }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            double a = s1[e], b = s2v[e];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
            if (li == 0) { red[wave][i * 16 + kq * 4 + e][0] = a; red[wave][i * 16 + kq * 4 + e][1] = b; }
          }
        }
      }
    }
    if (want_red) {
      __syncthreads();
      if (t < BC) {
        tot0 += red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
        tot1 += red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
      }
      __syncthreads();
    }
  }
  if (c_slab && t < BC && n0 + t < p.Cn) {
    double* o = c_slab + ((long)bx * p.slab_ld + n0 + t) * 2;
    o[0] = tot0; o[1] = tot1;
  }
}

int pick_ct(int Cn, long ntiles) {
  // smallest padding first, larger tiles on ties (fewer re-reads of the pixel tile) ...
  const int cands[5] = {8, 5, 4, 3, 2};
  int best = 2; long best_cost = -1;
  for (int k = 0; k < 5; ++k) {
    int ct = cands[k];
    long cols = (long)cdiv(Cn, 16 * ct) * 16 * ct;
    if (best_cost < 0 || cols < best_cost) { best_cost = cols; best = ct; }
  }
  // ... but a launch needs a few hundred workgroups to fill 256 CUs: on the small maps (levels 2-3) trade up to 25 %
  // channel padding for more, narrower column blocks (the pixel tile is re-read from L2, not HBM)
  for (int k = 0; k < 5 && ntiles * cdiv(Cn, 16 * best) < 512; ++k) {
    int ct = cands[k];
    if (ct >= best) continue;
    long cols = (long)cdiv(Cn, 16 * ct) * 16 * ct;
    if (cols * 4 <= best_cost * 5 && ntiles * cdiv(Cn, 16 * ct) > ntiles * cdiv(Cn, 16 * best)) best = ct;
  }
  return best;
}
int pick_pt(long P) { return P >= 128L * 512 ? 2 : 1; }

// Arithmetic of the wide k x k contractions (the halo-patch kernels of conv3.hip, the split weight gradients of wgrad.hip): 0 = exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32), 1 = "f16x3" (default): split-fp16, two terms per operand under an exact power-of-two scale, three product terms
// (conv3b.h), 2 = split-bf16 with six product terms (the same accuracy at twice the matrix instructions), 3 = "tail_x3": f16x3 in the exit heads
// only (launches with >= 192 output / gradient channels: ASPP and decoder forward, data and weight gradients), bf16x6 everywhere else.
// addk_set_conv_precision / ADDK_MATH=fp32|f16x3|bf16x6|tail_x3.  Every other kernel is fp32.
int g_prec = -1;
int conv_precision() {
  if (g_prec < 0) {
    g_prec = addk_env_math(ADDK_DEFAULT_PRECISION);
  }
  return g_prec;
}

int g_fast = -1;
int fast_paths() {
  if (g_fast < 0) {
    int m = ADDK_FAST_PW | ADDK_FAST_CONV3 | ADDK_FAST_WGRAD3 | ADDK_FAST_DWTILE | ADDK_FAST_WGRAD_RS;
    if (!addk_env("ADDK_PW", 1)) m &= ~ADDK_FAST_PW;
    if (!addk_env("ADDK_C3", 1)) m &= ~ADDK_FAST_CONV3;
    if (!addk_env("ADDK_WGRAD_H3", 1)) m &= ~ADDK_FAST_WGRAD3;
    if (!addk_env("ADDK_DWTILE", 1)) m &= ~ADDK_FAST_DWTILE;
    if (!addk_env("ADDK_WGRAD_RS", 1)) m &= ~ADDK_FAST_WGRAD_RS;
    g_fast = m;
  }
  return g_fast;
}
bool pw_enabled() { return (fast_paths() & ADDK_FAST_PW) != 0; }

template <int MODE>
int launch(ConvK& k, hipStream_t st, int grid_x = 0) {
  const int pt = pick_pt(k.P);
  const int BP = 64 * pt;
  k.ntiles = cdiv(k.P, BP);
  for (int c = 0; c < k.ncls; ++c) k.cl[c].ntiles = cdiv(k.cl[c].P, BP);      // combined parity classes: k.P is the largest class
  k.red32 = k.P >= 4096;
  const int ct = pick_ct(k.Cn, k.ntiles);
  dim3 grid(grid_x > 0 ? grid_x : addk_conv_rows(k.P, k.Cn), cdiv(k.Cn, 16 * ct));   // workgroups beyond ntiles only write their (zero) slab row
#define ADDK_CASE(PT_, CT_) \
  if (pt == PT_ && ct == CT_) { \
    hipLaunchKernelGGL((conv_kernel<PT_, CT_, MODE, PREC_F32>), grid, dim3(256), 0, st, k); \
    return addk_check_launch("conv"); }
  ADDK_CASE(1, 2) ADDK_CASE(1, 3) ADDK_CASE(1, 4) ADDK_CASE(1, 5) ADDK_CASE(1, 8)
  ADDK_CASE(2, 2) ADDK_CASE(2, 3) ADDK_CASE(2, 4) ADDK_CASE(2, 5) ADDK_CASE(2, 8)
#undef ADDK_CASE
  addk_set_error("conv: no tile config");
  return ADDK_ERR_UNSUPPORTED;
}

__global__ void mfma_selftest_kernel(float* out) {
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  // A[i][k] = i*4+k  (lane supplies A[li][kq]);  B[k][j] = (k+1)*(j+2) (lane supplies B[kq][li])
  float a = (float)(li * 4 + kq), b = (float)((kq + 1) * (li + 2));
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(kq * 4 + r) * 16 + li] = acc[r];   // D[row=kq*4+r][col=li]
}

}  // namespace

extern "C" int addk_set_conv_precision(int mode) {
  if (mode < 0 || mode > 3) { addk_set_error("conv precision must be 0 (fp32), 1 (f16x3), 2 (bf16x6) or 3 (tail_x3)"); return ADDK_ERR_INVALID; }
  g_prec = mode;
  return ADDK_OK;
}
extern "C" int addk_get_conv_precision(void) { return conv_precision(); }
extern "C" int addk_set_fast_paths(int mask) { g_fast = mask & (ADDK_FAST_PW | ADDK_FAST_CONV3 | ADDK_FAST_WGRAD3 | ADDK_FAST_DWTILE | ADDK_FAST_WGRAD_RS); return ADDK_OK; }
extern "C" int addk_get_fast_paths(void) { return fast_paths(); }

extern "C" int addk_selftest_mfma(float* out256, void* stream) {
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
  return addk_check_launch("selftest");
}

extern "C" int addk_conv_rows(int64_t P, int32_t Cout) {
  (void)Cout;
  int nt = cdiv(P, 64);           // one slab row per workgroup; every conv kernel launches exactly this many in x
  // [r3] small maps (the 63x127 and 32x64 levels: 251 / 64 rows) left most of the 256 CUs with one workgroup or none: one row per 32 pixels
  // up to 512 rows
  if (nt < 512) { nt = cdiv(P, 32); if (nt > 512) nt = 512; }
  return nt < 1024 ? nt : 1024;
}

extern "C" int addk_conv_fwd(const addk_conv_args* a, void* stream) {
  ADDK_REQUIRE(a && a->nsrc >= 1 && a->nsrc <= ADDK_MAX_SRC, "conv_fwd: nsrc out of range");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->Cout > 0, "conv_fwd: empty shape");
  ADDK_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0 && a->dil > 0, "conv_fwd: bad kernel geometry");
  ADDK_REQUIRE(a->w && a->y && a->ldy >= a->Cout, "conv_fwd: null/short output");
  ADDK_REQUIRE(!a->stats || a->stats_ld == 0 || a->stats_ld >= a->Cout, "conv_fwd: stats_ld < Cout");
  // every output pixel must map inside the padded input (guards against OOB reads by construction:
  // taps outside [0,H) are masked, so only the geometry of the output grid needs checking)
  ConvK k;
  long ctot = 0;
  k.vecA = 1;
  for (int i = 0; i < a->nsrc; ++i) {
    k.src[i] = a->src[i];
    ADDK_REQUIRE(a->src[i].x && a->src[i].C > 0 && a->src[i].ld >= a->src[i].C, "conv_fwd: bad source %d", i);
    ADDK_REQUIRE((a->src[i].a == nullptr) == (a->src[i].b == nullptr), "conv_fwd: a/b must come together");
    if (!src_vec_ok(a->src[i])) k.vecA = 0;
    ctot += a->src[i].C;
  }
  ADDK_REQUIRE(a->w_choff + ctot <= a->cin_total, "conv_fwd: sources exceed cin_total");
  ADDK_REQUIRE(a->ldw >= a->KH * a->KW * a->cin_total, "conv_fwd: ldw too small");
  if (a->wpack && a->KH == 1 && a->KW == 1) {             // a 1x1 conv whose caller set up the split kernel's weight pack: that path first
    int r = addk_c3_try_fwd(a, addk_conv_rows((long)a->N * a->OH * a->OW, a->Cout), stream);
    if (r <= 0) return r;
  }
  if (pw_enabled()) {      // small pointwise shapes: register-stationary kernel (pw.hip)
    int r = addk_pw_try_fwd(a, addk_conv_rows((long)a->N * a->OH * a->OW, a->Cout), stream);
    if (r <= 0) return r;
  }
  if (a->wpack) {                                          // wide 3x3 stride-1: halo-patch kernel (conv3.hip)
    int r = addk_c3_try_fwd(a, addk_conv_rows((long)a->N * a->OH * a->OW, a->Cout), stream);
    if (r <= 0) return r;
  }
  for (int i = 0; i < a->nsrc; ++i)
    ADDK_REQUIRE(a->src[i].rs_hw == 0, "conv_fwd: source %d is a resampled map (rs_hw) but the launch is not one addk_conv_fwd_resample_ok() accepts", i);
  k.nsrc = a->nsrc;
  k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.KH = a->KH; k.KW = a->KW; k.stride = a->stride; k.pad = a->pad; k.dil = a->dil;
  k.Cn = a->Cout; k.ldw = a->ldw; k.cin_total = a->cin_total; k.w_choff = a->w_choff; k.ldy = a->ldy;
  k.w = a->w; k.y = a->y; k.bias = a->bias; k.bias_n = a->bias_n; k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.accumulate = 0; k.dst = addk_src{nullptr, nullptr, nullptr, 0, 0, 0, 0};
  k.sub = 1; k.ph = k.pw = 0; k.MH = k.MW = 0; k.ntaps_l = 0; k.kill = 0; k.ncls = 0;
  bool chan4 = true;
  for (int i = 0; i < a->nsrc; ++i) chan4 = chan4 && (a->src[i].C % 4 == 0);
  k.vecB = aligned16(a->w) && a->ldw % 4 == 0 && a->cin_total % 4 == 0 && a->w_choff % 4 == 0 && chan4;
  k.vecY = aligned16(a->y) && a->ldy % 4 == 0;
  k.P = (long)a->N * a->OH * a->OW;
  ADDK_REQUIRE(k.P < (1L << 30) && (long)a->N * a->H * a->W < (1L << 30), "conv_fwd: tensor too large for 32-bit pixel indexing");
  return launch<MODE_FWD>(k, (hipStream_t)stream);
}

extern "C" int addk_conv_dgrad(const addk_conv_dgrad_args* a, void* stream) {
  ADDK_REQUIRE(a && a->dy && a->w && a->g && a->dst.x, "conv_dgrad: null pointer");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->Cout > 0 && a->dst.C > 0, "conv_dgrad: empty shape");
  ADDK_REQUIRE(a->lddy >= a->Cout && a->ldg >= a->dst.C && a->dst.ld >= a->dst.C, "conv_dgrad: short stride");
  ADDK_REQUIRE(a->w_choff + a->dst.C <= a->cin_total && a->ldw >= a->KH * a->KW * a->cin_total, "conv_dgrad: weight layout");
  ADDK_REQUIRE((a->dst.a == nullptr) == (a->dst.b == nullptr), "conv_dgrad: a/b must come together");
  if (pw_enabled()) {
    int r = addk_pw_try_dgrad(a, addk_conv_rows((long)a->N * a->H * a->W, a->dst.C), stream);
    if (r <= 0) return r;
  }
  if (a->wpack) {
    int r = addk_c3_try_dgrad(a, addk_conv_rows((long)a->N * a->H * a->W, a->dst.C), stream);
    if (r <= 0) return r;
  }
  ConvK k;
  k.nsrc = 1;
  k.src[0] = addk_src{a->dy, nullptr, nullptr, a->lddy, a->Cout, 0, 0};
  k.vecA = src_vec_ok(k.src[0]);
  k.N = a->N; k.H = a->OH; k.W = a->OW;      // A side = dy
  k.OH = a->H; k.OW = a->W;                  // M side = input pixels
  k.KH = a->KH; k.KW = a->KW; k.stride = a->stride; k.pad = a->pad; k.dil = a->dil;
  k.Cn = a->dst.C; k.ldw = a->ldw; k.cin_total = a->cin_total; k.w_choff = a->w_choff; k.ldy = a->ldg;
  k.w = a->w; k.y = a->g; k.bias = nullptr; k.bias_n = nullptr; k.slab = (double*)a->dab; k.slab_ld = a->dst.C;
  k.dst = a->dst; k.accumulate = a->accumulate;
  k.vecB = aligned16(a->w) && a->ldw % 4 == 0 && a->cin_total % 4 == 0 && a->w_choff % 4 == 0 && a->dst.C % 4 == 0;
  k.vecY = aligned16(a->g) && a->ldg % 4 == 0 && src_vec_ok(a->dst);
  k.P = (long)a->N * a->H * a->W;
  ADDK_REQUIRE(k.P < (1L << 30) && (long)a->N * a->OH * a->OW < (1L << 30), "conv_dgrad: tensor too large for 32-bit pixel indexing");
  k.sub = 1; k.ph = k.pw = 0; k.MH = k.MW = 0; k.ntaps_l = 0; k.kill = 0; k.ncls = 0;
  const int rows = addk_conv_rows(k.P, a->dst.C);
  if (a->stride == 2 && a->KH * a->KW <= 9 && rows % 4 == 0 && a->H >= 2 && a->W >= 2) {
    // Stride 2: an input pixel only receives the taps whose offset matches its parity, (1,2,2,4) of the 9 taps of a 3x3
    // for the four (row, column) parity classes.  One launch per class over that class's quarter of the pixels does
    // 9/4 tap-chunks per pixel instead of 9 masked ones; each class writes its own quarter of the (dA,dB) slab rows.
    const long Pfull = k.P;
    // classes that no tap reaches (3 of 4 for a 1x1 stride-2 conv: FactorizedReduce) have a zero gradient: they are only
    // launched when the gradient buffer is being first-touched (to write the zeros), and the (dA,dB) slab rows are shared
    // out among the classes that do have taps.
    int ntl[4], tl[4][9], nvalid = 0;
    for (int cl = 0; cl < 4; ++cl) {
      const int ph = cl >> 1, pw = cl & 1;
      ntl[cl] = 0;
      for (int kh = 0; kh < a->KH; ++kh)
        for (int kw = 0; kw < a->KW; ++kw)
          if ((ph + a->pad - kh * a->dil) % 2 == 0 && (pw + a->pad - kw * a->dil) % 2 == 0) tl[cl][ntl[cl]++] = kh * a->KW + kw;
      if (ntl[cl]) ++nvalid;
    }
    if (nvalid == 0 || rows % nvalid != 0) nvalid = 0;       // 0: fall back to a quarter of the rows per class, every class launched
    if (rows >= 16) {
      // ONE launch for all classes (each alone is rows/4 workgroups = one per CU at the stems' sizes: 1 wave per SIMD).
      // The slab rows (= workgroups) are shared out in proportion to the classes' tap counts so that they finish together;
      // classes without taps get workgroups beyond the slab (they only store zeros, and only on a first touch).
      ConvK c = k;
      c.sub = 2; c.ncls = 0;
      int tot = 0, nv = 0;
      for (int cl = 0; cl < 4; ++cl) { tot += ntl[cl]; if (ntl[cl]) ++nv; }
      int gx0 = 0, left = rows, seen = 0;
      long pmax = 0;
      for (int pass = 0; pass < 2; ++pass)          // classes with taps first: their workgroups are the slab rows 0..rows-1
        for (int cl = 0; cl < 4; ++cl) {
          if ((ntl[cl] > 0) != (pass == 0)) continue;
          if (pass == 1 && a->accumulate) continue;            // nothing to add
          ConvK::Cls& q = c.cl[c.ncls];
          q.ph = cl >> 1; q.pw = cl & 1;
          q.MH = (a->H - q.ph + 1) / 2; q.MW = (a->W - q.pw + 1) / 2;
          q.P = (long)a->N * q.MH * q.MW;
          q.ntaps_l = ntl[cl] ? ntl[cl] : 1; q.kill = ntl[cl] ? 0 : 1; q.pad_ = 0;
          for (int i = 0; i < 9; ++i) q.taplist[i] = i < ntl[cl] ? tl[cl][i] : 0;
          int gx;
          if (pass == 0) {
            ++seen;
            gx = seen == nv ? left : (int)((long)rows * ntl[cl] / tot);
            if (gx < 1) gx = 1;
            if (gx > left - (nv - seen)) gx = left - (nv - seen);
            q.slab = k.slab ? k.slab + (long)gx0 * k.slab_ld * 2 : nullptr;
            left -= gx;
          } else {
            gx = rows / 4; q.slab = nullptr;
          }
          q.gx0 = gx0; gx0 += gx;
          if (q.P > pmax) pmax = q.P;
          ++c.ncls;
        }
      if (nv == 0) return ADDK_OK;               // cannot happen for a convolution (some tap always matches some class)
      c.P = pmax;
      return launch<MODE_DGRAD>(c, (hipStream_t)stream, gx0);
    }
    int vi = 0;
    for (int cl = 0; cl < 4; ++cl) {
        const int ph = cl >> 1, pw = cl & 1;
        ConvK c = k;
        c.sub = 2; c.ph = ph; c.pw = pw;
        c.MH = (a->H - ph + 1) / 2; c.MW = (a->W - pw + 1) / 2;
        c.P = (long)a->N * c.MH * c.MW;
        c.ntaps_l = ntl[cl];
        for (int i = 0; i < ntl[cl]; ++i) c.taplist[i] = tl[cl][i];
        int gx = rows / 4;
        if (c.ntaps_l == 0) {
          c.taplist[0] = 0; c.ntaps_l = 1; c.kill = 1;
          if (nvalid) {
            if (a->accumulate) continue;            // nothing to add
            c.slab = nullptr;
          }
        } else if (nvalid) {
          gx = rows / nvalid;
          if (c.slab) c.slab = k.slab + (long)vi * gx * k.slab_ld * 2;
          ++vi;
        }
        if (!nvalid && c.slab) c.slab = k.slab + (long)cl * (rows / 4) * k.slab_ld * 2;
        int rc = launch<MODE_DGRAD>(c, (hipStream_t)stream, gx);
        if (rc) return rc;
    }
    (void)Pfull;
    return ADDK_OK;
  }
  return launch<MODE_DGRAD>(k, (hipStream_t)stream);
}
