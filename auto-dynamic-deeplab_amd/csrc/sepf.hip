// Fused SepConv half (reference modeling/operations.py:51-53 and :55-57): lazy BatchNorm / ReLU prologue -> depthwise
// KS x KS (stride 1, dilation 1, 'same') -> pointwise 1x1 on the fp32 matrix cores -> epilogue, ONE launch, 2-D LDS tiles.
//
// Round 2's fused form staged KS input rows per 64-pixel output row segment: a 5x re-read of the input through the
// texture path (307 MB for a 10 MB tensor) that made it slower than the two launches it replaced.  Here a workgroup owns a
// (4R) x 16 pixel tile of ALL channels:
//   * the haloed input patch [(4R+KS-1)][16+KS-1][C] is staged ONCE (read amplification 1.6-2.5x, from L2), prologue and
//     zero padding applied on the way in.  Pixel stride KP floats with KP = 8 (mod 16): the 16-byte tap reads of a wave
//     (lane (li, kq) reads pixel li + dx, channel quad 4g + kq) touch every bank exactly once per 16-lane group for every
//     tap shift — brute-forced over the ds_read_b128 lane groups of MI355X_MICROARCH.md §LDS (C = 40 -> KP 40, no padding);
//   * wave w owns rows [wR, wR+R) x 16 pixels: lane (li = pixel, kq) computes the depthwise output of channels
//     16g + 4kq .. +3 — exactly the B fragment v_mfma_f32_16x16x4_f32 wants from it (k = the lane's 4 channels, one k-step
//     per element), so the depthwise result goes from the VALU registers straight into the matrix pipe: no transpose, no LDS
//     round trip.  With R = 2 an input row is read once for the two output rows it feeds (sliding accumulators), tap
//     weights of the current group in registers;
//   * pointwise weights sit in LDS as ready A fragments ([g][tile][lane] float4, conflict-free);
//   * training epilogue: raw store, depthwise output stored for the backward pass, BatchNorm (sum, sumsq) partials ->
//     slab row -> the LAST workgroup finalizes the statistics (bnfin.h): no bn_finalize launch;
//     inference epilogue: own frozen BatchNorm + the other branches of the cell block (ADD.py:108).
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "bnfin.h"

namespace {

struct SepfK {
  addk_src src; int N, H, W, C;
  const float* dww; const float* pww; int ldw;
  float* y; int ldy; float* t; int ldt;
  double* slab; int slab_ld; int rows;
  const float* ea; const float* eb; int nterm; addk_src term[ADDK_MAX_TERMS];
  int tiles_x, tiles_y, gx;
  int wt;                    // 1 (default): write-through (sc1) output stores, ADDK_SEPF_WT=0 restores write-back ones
  BnFin fin;
};

__device__ __forceinline__ float4 fma4(float4 w, float4 v, float4 a) {
  return make_float4(fmaf(w.x, v.x, a.x), fmaf(w.y, v.y, a.y), fmaf(w.z, v.z, a.z), fmaf(w.w, v.w, a.w));
}

template <int KS, int KG, int KP, int R>
struct SepfGeo {
  static constexpr int CT = KG, PH = 4 * R + KS - 1, PW = 16 + KS - 1, NPIX = PH * PW, KQ = KP / 4;
  static constexpr int PATCH = NPIX * KP + 8;                     // floats (+8: the clamped tail read of the last pixel stays inside)
  static constexpr int DWL = KS * KS * KG * 16, PWL = KG * CT * 64 * 4;
  static constexpr int RED = (4 * CT * 16 * 2 > 514 ? 4 * CT * 16 * 2 : 514) * 2;   // floats: [4][CT*16][2] doubles, or the finalize scratch
  static constexpr int EAB = 2 * KG * 16;                                            // inference epilogue coefficients (ea, eb)
  static constexpr int NPL = 256 / KQ;                                               // pixel lanes of the staging map: thread = (pixel lane, channel quad)
  static constexpr int NLD = (NPIX + NPL - 1) / NPL;                                 // patch loads per thread
  static constexpr int UNMAX = (KS == 5 && R == 2) ? 6 : 10;                         // (the 5x5 two-row variant is short of registers: two rounds)
  static constexpr int UN = NLD <= UNMAX ? NLD : (NLD + 1) / 2;                      // ... per round: one round where 10 loads in flight suffice
  static constexpr int TPX = 4 * R * 16, TLD = TPX * KP;                             // first sum term of the inference epilogue, staged like the patch
  static constexpr int NTL = (TPX + NPL - 1) / NPL;
  static constexpr int PAB = 2 * KG * 16;                                            // prologue coefficients (a, b) of the input
  static constexpr size_t LDS = (size_t)(PATCH + DWL + PWL + RED + EAB + TLD + PAB) * 4;
};

#ifdef ADDK_SEPF_DIAG
// diagnostic build (scripts/sepf_phases.sh): every workgroup adds the length of its phases in 100 MHz reference ticks (s_memrealtime) and
// keeps the earliest start / latest end of the launch, so that the in-kernel time can be set against the launch's wall time
__device__ unsigned long long g_sepf_diag[64][8];
__device__ unsigned long long g_sepf_span[2] = {~0ull, 0ull};
// [r5] per LAUNCH of a dependent chain: first / last workgroup start, first / last workgroup end; the launch index is a device
// counter the last-finishing workgroup advances (launches of the chain never overlap)
__device__ unsigned long long g_sepf_chain[64][4];
__device__ unsigned int g_sepf_launch = 0, g_sepf_ticket = 0;
#define SEPF_STAMP(i) const unsigned long long diag_t##i = __builtin_amdgcn_s_memrealtime()
#else
#define SEPF_STAMP(i)
#endif

template <int KS, int KG, int KP, int R>
__device__ __forceinline__ void sepf_body(const SepfK& p, float* sm) {
  SEPF_STAMP(0);
  typedef SepfGeo<KS, KG, KP, R> G;
  constexpr int CT = G::CT, PH = G::PH, PW = G::PW, NPIX = G::NPIX, KQ = G::KQ, HK = KS / 2;
  float* patch = sm;
  float* dwl = patch + G::PATCH;                 // [KS*KS][KG*16]
  float* pwl = dwl + G::DWL;                     // [KG][CT][64] float4
  double* red = reinterpret_cast<double*>(pwl + G::PWL);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int C = p.C, nq = C >> 2;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y; const int n = b / p.tiles_y;
  const int oh0 = ty * (4 * R), ow0 = tx * 16;

  // ---- stage (round 4).  Every global load of the tile is requested before anything waits — weights, the first sum term of the inference
  // epilogue, and the whole input patch — and the patch then moves to LDS ONE 16-CHANNEL GROUP AT A TIME, each group's depthwise + matrix
  // work running while the later groups' loads are still arriving (vmcnt is in order: using group g's registers waits for nothing younger).
  // The three dependent round trips of round 3 (weights -> patch -> epilogue operands: 2.5 + 1.7 + 2.7 us of a 9.8 us workgroup,
  // profiles/r04_sepf_phases.txt) are one, and the memory system stays busy under the compute phase. ----
  constexpr bool PIPE = KG == 5;                                 // the group pipeline pays at 80 channels (one workgroup per CU: 16.4 -> 15.2, 18.8 -> 17.6 us per
                                                                 // launch); at 40 channels two workgroups per CU already overlap each other and its registers cost
                                                                 // them occupancy (24.2 -> 27.0 us measured): those variants issue everything up front and store once
  constexpr int NU = PIPE ? (NPIX + 63) / 64 : 1, KGP = PIPE ? KG : 1;   // patch quads per thread and channel group: thread = (pixel lane pl of 64, quad qq of the group)
  constexpr int NWD = (KG * 16 * KS * KS + 255) / 256, NWP = (KG * CT * 64 + 255) / 256, NTU = G::TPX / 64;
  float* eab = reinterpret_cast<float*>(red) + G::RED;
  float* tl = eab + G::EAB;                                      // [4R x 16 pixels][KP]: first sum term
  float* pab = tl + G::TLD;                                      // [2][KG*16]: prologue coefficients
  const int qq = t & 3;
  const int ih0 = oh0 - HK, iw0 = ow0 - HK;
  const bool t0 = p.nterm > 0;
  int pp[R]; bool pin[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int oh = oh0 + wave * R + r, ow = ow0 + li;
    pin[r] = oh < p.H && ow < p.W;
    pp[r] = (n * p.H + oh) * p.W + ow;
  }
  float4 pv[KGP][NU]; unsigned pokm[KGP];
  if constexpr (PIPE) {
    const int pl = t >> 2;
    // (1) weights and coefficient vectors -> registers
    float wd[NWD]; float4 wp[NWP];
  #pragma unroll
    for (int j = 0; j < NWD; ++j) {
      const int i = t + 256 * j;
      wd[j] = ((const gfloat*)p.dww)[i < C * KS * KS ? i : 0];
    }
  #pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int s = t + 256 * j;
      const int g = s / (CT * 64), rem = s - g * (CT * 64), ct = rem >> 6, ln = rem & 63;
      const int nn = ct * 16 + (ln & 15), k = 16 * g + 4 * (ln >> 4);
      const bool ok = s < KG * CT * 64 && nn < C && k < C;
      float4 v = ld4(ok ? p.pww + (long)nn * p.ldw + k : p.pww);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      wp[j] = v;
    }
    float4 cab = make_float4(1.f, 1.f, 1.f, 1.f), ceab = zero4();
    if (t < 2 * nq) {
      const int which = t / nq, cq = t - which * nq;
      cab = which ? zero4() : cab;
      if (p.src.a) cab = ld4((which ? p.src.b : p.src.a) + 4 * cq);
      if (p.ea) ceab = ld4((which ? p.eb : p.ea) + 4 * cq);
    }
    // (2) the first sum term (a cell block has two branches: one term): parked in LDS, read by the epilogue
    float4 tq[KG][NTU];
    if (t0) {
  #pragma unroll
      for (int g = 0; g < KG; ++g)
  #pragma unroll
        for (int u = 0; u < NTU; ++u) {
          const int pix = pl + 64 * u, q = 4 * g + qq;
          const int oh = oh0 + (pix >> 4), ow = ow0 + (pix & 15);
          const bool ok = 4 * q < C && oh < p.H && ow < p.W;
          tq[g][u] = ld4(p.term[0].x + (ok ? ((long)(n * p.H + oh) * p.W + ow) * p.term[0].ld + 4 * q : 0));
        }
    }
    // (3) the input patch, group by group
  #pragma unroll
    for (int g = 0; g < KG; ++g) {
      pokm[g] = 0;
  #pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int pix = pl + 64 * u, q = 4 * g + qq;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int ih = ih0 + pr, iw = iw0 + pc;
        const bool ok = pix < NPIX && 4 * q < C && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        pv[g][u] = ld4(p.src.x + (ok ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + 4 * q : 0));
        pokm[g] |= (ok ? 1u : 0u) << u;
      }
    }
    __builtin_amdgcn_sched_barrier(0);             // the loads above stay above: hipcc otherwise sinks each one to its first use
    // (4) weights, coefficients and the sum term -> LDS; zero fill of what nobody writes (channels beyond C of the tap table, padding quads of the patch)
    for (int i = t; i < KS * KS * (KG * 16 - C); i += 256) {
      const int tp = i / (KG * 16 - C), c = C + i - tp * (KG * 16 - C);
      dwl[tp * (KG * 16) + c] = 0.f;
    }
  #pragma unroll
    for (int j = 0; j < NWD; ++j) {                                // [C][KS*KS] transposed into [tap][channel]
      const int i = t + 256 * j;
      if (i < C * KS * KS) { const int c = i / (KS * KS), tp = i - c * (KS * KS); dwl[tp * (KG * 16) + c] = wd[j]; }
    }
  #pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int s = t + 256 * j;
      if (s < KG * CT * 64) lds_st4(pwl + s * 4, wp[j]);
    }
    if (t < 2 * nq) {
      const int which = t / nq, cq = t - which * nq;
      lds_st4(pab + which * (KG * 16) + 4 * cq, cab);
      if (p.ea) lds_st4(eab + which * (KG * 16) + 4 * cq, ceab);
    }
    if (KQ > 4 * KG) {                                             // padding quads beyond the last channel group (KP = 56, 88)
      for (int i = t; i < NPIX * (KQ - 4 * KG); i += 256) {
        const int px = i / (KQ - 4 * KG), q = 4 * KG + i - px * (KQ - 4 * KG);
        lds_st4(patch + px * KP + 4 * q, zero4());
      }
    }
    if (t < 8) patch[NPIX * KP + t] = 0.f;
    if (t0) {
  #pragma unroll
      for (int g = 0; g < KG; ++g)
  #pragma unroll
        for (int u = 0; u < NTU; ++u) {
          const int pix = pl + 64 * u, q = 4 * g + qq;
          if (q < KQ) lds_st4(tl + pix * KP + 4 * q, tq[g][u]);
        }
    }
    __syncthreads();
  } else {
    constexpr int NPL = G::NPL, UN = G::UN;
    const int q = t % KQ, pl = t / KQ;
    const bool qact = pl < NPL && q < nq;
    const long xq = qact ? 4 * q : 0;
      float4 pv[UN]; bool pok[UN];
    auto patch_issue = [&](int base) {
  #pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int pix = base + u * NPL;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int ih = ih0 + pr, iw = iw0 + pc;
        pok[u] = qact && pix < NPIX && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        pv[u] = ld4(p.src.x + xq + (pok[u] ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld : 0));
      }
    };
    patch_issue(pl);
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
    if (p.src.a && qact) { av = ld4(p.src.a + 4 * q); bv = ld4(p.src.b + 4 * q); }
    // the first sum term (a cell block has two branches: one term) travels with the patch: loaded now, parked in LDS, read by the epilogue
    constexpr int NTL = G::NTL;
    float4 tq[NTL];
    if (t0) {
  #pragma unroll
      for (int u = 0; u < NTL; ++u) {
        const int pix = pl + u * NPL;
        const int oh = oh0 + (pix >> 4), ow = ow0 + (pix & 15);
        const bool ok = qact && pix < G::TPX && oh < p.H && ow < p.W;
        tq[u] = ld4(p.term[0].x + (ok ? ((long)(n * p.H + oh) * p.W + ow) * p.term[0].ld + 4 * q : 0));
      }
    }
    __builtin_amdgcn_sched_barrier(0);             // the loads above stay above: hipcc otherwise sinks each one to its first use
    // weights (tiny, L2-resident).  Channels beyond C of the tap table are zeroed by the threads that do not write a weight there (no
    // barrier between a zero fill and the weights any more)
    for (int i = t; i < KS * KS * (KG * 16 - C); i += 256) {
      const int tp = i / (KG * 16 - C), c = C + i - tp * (KG * 16 - C);
      dwl[tp * (KG * 16) + c] = 0.f;
    }
    for (int i = t; i < C * KS * KS; i += 256) {                 // coalesced read of [C][KS*KS], transposed into [tap][channel]
      const int c = i / (KS * KS), tp = i - c * (KS * KS);
      dwl[tp * (KG * 16) + c] = ((const gfloat*)p.dww)[i];
    }
    for (int s = t; s < KG * CT * 64; s += 256) {
      const int g = s / (CT * 64), rem = s - g * (CT * 64), ct = rem >> 6, ln = rem & 63;
      const int nn = ct * 16 + (ln & 15), k = 16 * g + 4 * (ln >> 4);
      const bool ok = nn < C && k < C;
      float4 v = ld4(ok ? p.pww + (long)nn * p.ldw + k : p.pww);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      lds_st4(pwl + s * 4, v);
    }
    if (t0 && qact) {
  #pragma unroll
      for (int u = 0; u < NTL; ++u) {
        const int pix = pl + u * NPL;
        if (pix < G::TPX) lds_st4(tl + pix * KP + 4 * q, tq[u]);
      }
    }
    if (p.ea && t < 2 * nq) {                                    // frozen-BatchNorm coefficients of the inference epilogue -> LDS
      const int which = t / nq, qq = t - which * nq;
      lds_st4(eab + which * (KG * 16) + 4 * qq, ld4((which ? p.eb : p.ea) + 4 * qq));
    }
    {
      const bool relu = p.src.relu != 0;
      if (pl < NPL) {
        for (int base = pl; base < NPIX; base += UN * NPL) {
          if (base != pl) patch_issue(base);
  #pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int pix = base + u * NPL;
            float4 z = fma4(av, pv[u], bv);
            if (relu) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
            z.x = pok[u] ? z.x : 0.f; z.y = pok[u] ? z.y : 0.f; z.z = pok[u] ? z.z : 0.f; z.w = pok[u] ? z.w : 0.f;
            if (pix < NPIX) lds_st4(patch + pix * KP + 4 * q, z);
          }
        }
      }
      if (t < 8) patch[NPIX * KP + t] = 0.f;
    }

    __syncthreads();
  }
  SEPF_STAMP(1);
  const bool relu = p.src.relu != 0;
  auto patch_store = [&](int g) {                                // prologue (lazy BatchNorm / ReLU), zero padding, group g of the patch -> LDS
    const int q = 4 * g + qq, pl = t >> 2;
    if (q >= KQ) return;
    const float4 av = lds_ld4(pab + 4 * q), bv = lds_ld4(pab + KG * 16 + 4 * q);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int pix = pl + 64 * u;
      const bool ok = (pokm[g] >> u) & 1u;
      float4 z = fma4(av, pv[g][u], bv);
      if (relu) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
      z.x = ok ? z.x : 0.f; z.y = ok ? z.y : 0.f; z.z = ok ? z.z : 0.f; z.w = ok ? z.w : 0.f;
      if (pix < NPIX) lds_st4(patch + pix * KP + 4 * q, z);
    }
  };

  // ---- compute: wave = rows [wave*R, wave*R + R) x 16 pixels; group g's patch slice is stored, the workgroup meets, group g is consumed ----
#ifdef ADDK_SEPF_DIAG
  unsigned long long diag_t23 = 0;
#endif
  f32x4 macc[R][CT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < CT; ++i) macc[r][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (PIPE) { patch_store(g); __syncthreads(); }
#ifdef ADDK_SEPF_DIAG
    const unsigned long long diag_tg = __builtin_amdgcn_s_memrealtime();
    if (g == 0) { diag_t23 = diag_tg; }
#endif
    const int q = 4 * g + kq;
    const int qr = q < KQ ? q : q - 2;             // quads past the padded pixel (only when KP < 16 KG): re-read a valid quad, its tap weights are 0
    float4 wr[KS * KS];
#pragma unroll
    for (int tp = 0; tp < KS * KS; ++tp) wr[tp] = lds_ld4(dwl + tp * (KG * 16) + 4 * q);
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero4();
    const float* pb = patch + ((wave * R) * PW + li) * KP + 4 * qr;
#pragma unroll
    for (int i = 0; i < R + KS - 1; ++i)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const float4 v = lds_ld4(pb + (i * PW + dx) * KP);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int kh = i - r;
          if (kh >= 0 && kh < KS) acc[r] = fma4(wr[kh * KS + dx], v, acc[r]);
        }
      }
    if (p.t && 4 * q < C) {
#pragma unroll
      for (int r = 0; r < R; ++r) if (pin[r]) st4(p.t + (long)pp[r] * p.ldt + 4 * q, acc[r]);
    }
    float4 wf[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wf[i] = lds_ld4(pwl + ((g * CT + i) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < CT; ++i)
          macc[r][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(wf[i], e), get4(acc[r], e), macc[r][i], 0, 0, 0);
  }

  SEPF_STAMP(4);
  // ---- epilogue: lane holds channels i*16 + kq*4 + {0..3} of pixel pp[r] ----
  float s1[CT][4], s2[CT][4];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[i][e] = 0.f; s2[i][e] = 0.f; }
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int c = i * 16 + kq * 4;
    if (c >= C) continue;                                   // C % 4 == 0: a quad is valid as a whole
    float4 ea = make_float4(1.f, 1.f, 1.f, 1.f), eb = zero4();
    if (p.ea) { ea = lds_ld4(eab + c); eb = lds_ld4(eab + KG * 16 + c); }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!pin[r]) continue;
      float4 v = make_float4(macc[r][i][0], macc[r][i][1], macc[r][i][2], macc[r][i][3]);
      if (p.ea) v = fma4(ea, v, eb);
      if (t0) {                                              // same order and expressions as before: term 0 (from LDS), then the rest
        const addk_src& T = p.term[0];
        const float4 u = prologue4(lds_ld4(tl + ((wave * R + r) * 16 + li) * KP + c), T.a, T.b, c, 4, T.relu != 0, true);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      for (int ti = 1; ti < p.nterm; ++ti) {
        const addk_src& T = p.term[ti];
        const float4 u = prologue4(ld4(T.x + (long)pp[r] * T.ld + c), T.a, T.b, c, 4, T.relu != 0, true);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (p.wt) st4_wt(p.y + (long)pp[r] * p.ldy + c, v); else st4(p.y + (long)pp[r] * p.ldy + c, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float f = get4(v, e); s1[i][e] += f; s2[i][e] = fmaf(f, f, s2[i][e]); }
    }
  }
#ifdef ADDK_SEPF_DIAG
  if (t == 0) {
    const unsigned long long diag_t5 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = g_sepf_diag[blockIdx.x & 63];
    atomicAdd(&d[0], diag_t1 - diag_t0); atomicAdd(&d[1], diag_t23 - diag_t1); atomicAdd(&d[2], 0ull);
    atomicAdd(&d[3], diag_t4 - diag_t23); atomicAdd(&d[4], diag_t5 - diag_t4); atomicAdd(&d[5], 1ull);
    atomicMin(&g_sepf_span[0], diag_t0); atomicMax(&g_sepf_span[1], diag_t5);
    const unsigned id = __hip_atomic_load(&g_sepf_launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 63u;
    atomicMin(&g_sepf_chain[id][0], diag_t0); atomicMax(&g_sepf_chain[id][1], diag_t0);
    atomicMin(&g_sepf_chain[id][2], diag_t5); atomicMax(&g_sepf_chain[id][3], diag_t5);
    __threadfence();
    if (atomicAdd(&g_sepf_ticket, 1u) == (unsigned)p.gx - 1u) { g_sepf_ticket = 0; __threadfence(); atomicAdd(&g_sepf_launch, 1u); }
  }
#endif
  if (p.slab) {            // butterfly over the 16 pixel lanes (fp64), the four waves through LDS, one slab row per workgroup
    double (*rd)[CT * 16][2] = reinterpret_cast<double (*)[CT * 16][2]>(red);
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double a = (double)s1[i][e], bsum = (double)s2[i][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); bsum += __shfl_xor(bsum, m); }
        if (li == 0) { rd[wave][i * 16 + kq * 4 + e][0] = a; rd[wave][i * 16 + kq * 4 + e][1] = bsum; }
      }
    __syncthreads();
    const bool fused = p.fin.a != nullptr;
    if (t < CT * 16 && t < C) {
      double* o = p.slab + ((long)blockIdx.x * p.slab_ld + t) * 2;
      const double o0 = rd[0][t][0] + rd[1][t][0] + rd[2][t][0] + rd[3][t][0];
      const double o1 = rd[0][t][1] + rd[1][t][1] + rd[2][t][1] + rd[3][t][1];
      if (fused) { slab_store_wt(o, o0); slab_store_wt(o + 1, o1); }
      else {
        ((gdouble*)o)[0] = o0; ((gdouble*)o)[1] = o1;
        for (int r = blockIdx.x + p.gx; r < p.rows; r += p.gx) {       // rows no workgroup owns
          gdouble* z = (gdouble*)p.slab + ((long)r * p.slab_ld + t) * 2;
          z[0] = 0.0; z[1] = 0.0;
        }
      }
    }
    if (fused) bn_finalize_by_last_block(p.fin, p.slab, p.slab_ld, C, blockIdx.x, (unsigned)p.gx, red);
  }
}

template <int KS, int KG, int KP, int R>
__global__ void __launch_bounds__(256, 2) sepf_kernel(const SepfK p) {
  extern __shared__ __attribute__((aligned(16))) float sepf_sm[];
  sepf_body<KS, KG, KP, R>(p, sepf_sm);
}
template <int KS, int KG, int KP, int R>
__global__ void __launch_bounds__(256, 2) sepf_batch_kernel(const SepfK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float sepf_sm[];
  const SepfK& p = tab[blockIdx.z];          // by reference: a local copy with its runtime-indexed term[] would live in scratch
  if ((int)blockIdx.x >= p.gx) return;
  sepf_body<KS, KG, KP, R>(p, sepf_sm);
}

struct SepfCfg { int ks, kg, kp, r; };
inline int sepf_key(const SepfCfg& c) { return (c.ks << 16) | (c.kg << 12) | (c.kp << 4) | c.r; }

bool sepf_fill(const addk_sep_args* a, SepfK& k, SepfCfg& c) {
  if (!a || !(a->K == 3 || a->K == 5) || a->N <= 0 || a->H <= 0 || a->W <= 0) return false;
  const addk_src& s = a->src;
  const int kg = cdiv(s.C, 16);
  if (!(kg == 3 || kg == 5) || a->Cout != s.C || !s.x || !src_vec_ok(s) || !a->dw_w || !a->pw_w || !a->y) return false;
  if (!aligned16(a->y) || a->ldy % 4 || a->ldy < a->Cout || !aligned16(a->pw_w) || a->ldw % 4 || a->ldw < s.C) return false;
  if (a->t && (!aligned16(a->t) || a->ldt % 4 || a->ldt < s.C)) return false;
  if (a->nterm < 0 || a->nterm > ADDK_MAX_TERMS || (a->ea == nullptr) != (a->eb == nullptr)) return false;
  if ((a->nterm > 0 || a->ea) && a->stats) return false;               // the sum epilogue is an inference form
  if (a->ea && (!aligned16(a->ea) || !aligned16(a->eb))) return false;
  if (a->fin.a && (!a->stats || !a->fin_counter || !a->fin.b || a->fin.count <= 0)) return false;
  for (int i = 0; i < a->nterm; ++i) if (!a->term[i].x || a->term[i].C != a->Cout || !src_vec_ok(a->term[i])) return false;
  int kp = s.C; while (kp % 16 != 8) kp += 4;
  if (!((kg == 3 && (kp == 40 || kp == 56)) || (kg == 5 && (kp == 72 || kp == 88)))) return false;
  k = SepfK{};
  k.src = s; k.N = a->N; k.H = a->H; k.W = a->W; k.C = s.C;
  k.dww = a->dw_w; k.pww = a->pw_w; k.ldw = a->ldw; k.y = a->y; k.ldy = a->ldy; k.t = a->t; k.ldt = a->ldt;
  k.slab = (double*)a->stats; k.slab_ld = a->stats_ld > 0 ? a->stats_ld : a->Cout;
  k.ea = a->ea; k.eb = a->eb; k.nterm = a->nterm;
  for (int i = 0; i < a->nterm; ++i) k.term[i] = a->term[i];
  // two rows per wave where that still gives the chip >= 1.5 workgroups per CU (the LDS patch of a KG = 5 tile is 56 KB at R = 1)
  const long blocks2 = (long)a->N * cdiv(a->H, 8) * cdiv(a->W, 16);
  // 80-channel tiles need 100-127 KB of LDS: one workgroup per CU.  That is fine while the launch has at most two rounds of them
  // (config 2: 256 workgroups at 64x128) and LOSES to the separate depthwise / pointwise launches beyond (F = 40, 80 channels at
  // 128x256 = 1024 workgroups: step 72.2 ms fused vs 66.5 ms unfused) — those shapes stay on the unfused kernels
  if (kg == 5 && (long)a->N * cdiv(a->H, 4) * cdiv(a->W, 16) > 512) return false;
  c.ks = a->K; c.kg = kg; c.kp = kp; c.r = (kg == 3 && blocks2 >= 384) ? 2 : 1;
  k.tiles_x = cdiv(a->W, 16); k.tiles_y = cdiv(a->H, 4 * c.r); k.gx = a->N * k.tiles_y * k.tiles_x;
  k.wt = addk_env("ADDK_SEPF_WT", 1);
  k.rows = a->stats_rows;
  if (k.slab && k.gx > k.rows) return false;             // the caller sizes the slab with addk_sep_rows
  if (a->fin.a) {
    k.fin.a = a->fin.a; k.fin.b = a->fin.b; k.fin.mean = a->fin.mean; k.fin.invstd = a->fin.invstd; k.fin.gamma = a->fin.gamma; k.fin.beta = a->fin.beta;
    k.fin.running_mean = a->fin.running_mean; k.fin.running_var = a->fin.running_var; k.fin.count = a->fin.count;
    k.fin.momentum = a->fin.momentum; k.fin.eps = a->fin.eps;
    bnfin_bind_ws(k.fin, a->fin_counter, k.gx);
  }
  return true;
}

template <int KS, int KG, int KP, int R>
int sepf_go(bool batch, dim3 grid, hipStream_t st, const SepfK* one, const SepfK* tab) {
  typedef SepfGeo<KS, KG, KP, R> G;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepf_kernel<KS, KG, KP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepf_batch_kernel<KS, KG, KP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    attr = true;
  }
  if (batch) hipLaunchKernelGGL((sepf_batch_kernel<KS, KG, KP, R>), grid, dim3(256), G::LDS, st, tab);
  else hipLaunchKernelGGL((sepf_kernel<KS, KG, KP, R>), grid, dim3(256), G::LDS, st, *one);
  return addk_check_launch("sep_fwd");
}

int sepf_dispatch(const SepfCfg& c, bool batch, dim3 grid, hipStream_t st, const SepfK* one, const SepfK* tab) {
#define ADDK_SEPF(KS_, KG_, KP_, R_) if (c.ks == KS_ && c.kg == KG_ && c.kp == KP_ && c.r == R_) return sepf_go<KS_, KG_, KP_, R_>(batch, grid, st, one, tab);
  ADDK_SEPF(3, 3, 40, 1) ADDK_SEPF(3, 3, 40, 2) ADDK_SEPF(5, 3, 40, 1) ADDK_SEPF(5, 3, 40, 2)
  ADDK_SEPF(3, 3, 56, 1) ADDK_SEPF(3, 3, 56, 2) ADDK_SEPF(5, 3, 56, 1) ADDK_SEPF(5, 3, 56, 2)
  ADDK_SEPF(3, 5, 72, 1) ADDK_SEPF(5, 5, 72, 1) ADDK_SEPF(3, 5, 88, 1) ADDK_SEPF(5, 5, 88, 1)
#undef ADDK_SEPF
  addk_set_error("sep_fwd: no instantiation");
  return ADDK_ERR_UNSUPPORTED;
}

}  // namespace

#ifdef ADDK_SEPF_DIAG
// out[0..4]: phase ticks (weights, patch loads + LDS stores, barrier, compute, epilogue issue) summed over workgroups, out[5]: workgroups,
// out[6]: latest end - earliest start of all workgroups since the last call (10 ns ticks); resets
// the chain record: out[64][4] = (first start, last start, first end, last end) of the launches since the last call, 10 ns ticks; *n = launches; resets
extern "C" int addk_sepf_chain(unsigned long long* out256, unsigned int* n) {
  unsigned long long h[64][4]; unsigned int z = 0;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sepf_chain), sizeof h) != hipSuccess) return ADDK_ERR_INVALID;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_sepf_launch), sizeof z) != hipSuccess) return ADDK_ERR_INVALID;
  memcpy(out256, h, sizeof h);
  for (int i = 0; i < 64; ++i) { h[i][0] = ~0ull; h[i][1] = 0; h[i][2] = ~0ull; h[i][3] = 0; }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sepf_launch), &z, sizeof z);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sepf_ticket), &z, sizeof z);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sepf_chain), h, sizeof h) == hipSuccess ? ADDK_OK : ADDK_ERR_INVALID;
}
extern "C" int addk_sepf_diag(unsigned long long* out8) {
  unsigned long long h[64][8], sp[2];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sepf_diag), sizeof h) != hipSuccess) return ADDK_ERR_INVALID;
  if (hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_sepf_span), sizeof sp) != hipSuccess) return ADDK_ERR_INVALID;
  for (int k = 0; k < 8; ++k) { out8[k] = 0; for (int i = 0; i < 64; ++i) out8[k] += h[i][k]; }
  out8[6] = sp[1] > sp[0] ? sp[1] - sp[0] : 0;
  memset(h, 0, sizeof h); sp[0] = ~0ull; sp[1] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sepf_span), sp, sizeof sp);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sepf_diag), h, sizeof h) == hipSuccess ? ADDK_OK : ADDK_ERR_INVALID;
}
#endif
// slab rows a fused launch writes (= its workgroups): the caller sizes `stats` with max(this, addk_conv_rows)
extern "C" int addk_sep_rows(const addk_sep_args* a) {
  SepfK k; SepfCfg c;
  addk_sep_args b = *a; b.stats = nullptr; b.fin.a = nullptr;
  return sepf_fill(&b, k, c) ? k.gx : 0;
}
// bytes of the zero-initialised workspace `fin_counter` points to, for a launch of `nblocks` workgroups and slab rows of ld channels
extern "C" int64_t addk_bn_fin_ws_bytes(int32_t nblocks, int32_t ld) { return nblocks > 0 && ld > 0 ? bnfin_ws_bytes(nblocks, ld) : 0; }
extern "C" int addk_sep_fwd_supported(const addk_sep_args* a) {
  SepfK k; SepfCfg c;
  return (addk_get_fast_paths() & ADDK_FAST_PW) && sepf_fill(a, k, c) ? 1 : 0;
}
extern "C" int addk_sep_fwd(const addk_sep_args* a, void* stream) {
  SepfK k; SepfCfg c;
  ADDK_REQUIRE(sepf_fill(a, k, c), "sep_fwd: shape not covered by the fused kernel (K in {3,5}, C == Cout in (32,48] or (64,80], aligned, stats_rows >= addk_sep_rows)");
  return sepf_dispatch(c, false, dim3(k.gx), (hipStream_t)stream, &k, nullptr);
}
// batched form (one dependency level): key >= 0 groups launches that share a kernel variant
extern "C" int addk_sep_fwd_batch_key(const addk_sep_args* a) {
  SepfK k; SepfCfg c;
  if (!(addk_get_fast_paths() & ADDK_FAST_PW) || !sepf_fill(a, k, c)) return -1;
  return sepf_key(c);
}
extern "C" int64_t addk_sep_fwd_batch_prepare(const addk_sep_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  if (!a || n <= 0 || !meta) { addk_set_error("sep_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  const int64_t total = (int64_t)n * sizeof(SepfK);
  if (host_blob && blob_bytes < total) { addk_set_error("sep_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  int key0 = -1, gx = 0;
  for (int i = 0; i < n; ++i) {
    SepfK k; SepfCfg c;
    if (!sepf_fill(&a[i], k, c)) { addk_set_error("sep_batch_prepare: launch %d is not covered", i); return ADDK_ERR_INVALID; }
    const int key = sepf_key(c);
    if (i == 0) key0 = key;
    if (key != key0) { addk_set_error("sep_batch_prepare: mixed kernel variants"); return ADDK_ERR_INVALID; }
    if (k.gx > gx) gx = k.gx;
    if (host_blob) reinterpret_cast<SepfK*>(host_blob)[i] = k;
  }
  meta[0] = key0; meta[1] = n; meta[2] = gx; meta[3] = 1;
  return total;
}
extern "C" int addk_sep_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[1] > 0 && meta[2] > 0, "sep_batch_run: bad args");
  const int key = (int)meta[0];
  SepfCfg c{key >> 16, (key >> 12) & 15, (key >> 4) & 255, key & 15};
  return sepf_dispatch(c, true, dim3((unsigned)meta[2], 1, (unsigned)meta[1]), (hipStream_t)stream, nullptr, reinterpret_cast<const SepfK*>(dev_blob));
}
