// A WHOLE SepConv in one launch, inference form (reference modeling/operations.py:46-62 with the BatchNorms frozen):
//     relu(a0 x + b0) -> depthwise K x K -> pointwise 1x1 -> BatchNorm (a1, b1) -> ReLU -> depthwise K x K -> pointwise 1x1
//     -> own frozen BatchNorm (ea, eb) + the other branches of the cell block (ADD.py:108), or the raw result.
// Round 3 ran the two halves as two launches of csrc/sepf.hip with the 10 MB intermediate tensor written and re-read in between, and
// priced their fusion as a wash.  Round 4's measurements say otherwise: a dependent launch of this size costs ~7 us beyond its in-kernel
// time and ~3.1 us + bytes / 5.6 TB/s at best (profiles/r04_sepf_phases.txt, r04_persistent_vs_launch_chain.txt) — the second launch's floor
// (boundary, 10 MB write-back, 10 MB re-read with halo) is what the fusion removes, against recomputing the first half on the second
// half's halo (x1.7 of its pixels at K = 3, x2.5 at K = 5).
//
// A workgroup owns a 4 x 16-pixel output tile of ALL channels:
//   * the input patch [(4 + 2(K-1))][(16 + 2(K-1))][C] is staged ONCE in LDS (prologue and zero padding on the way in; pixel stride KP = 8
//     mod 16 floats as in sepf.hip), with every other global load of the tile (weights, coefficient vectors, the first sum term) in the
//     same round trip;
//   * half 1 runs on the (4 + K-1) x (16 + K-1) MID region the second depthwise conv reads: the mid pixels are flattened into 16-pixel MFMA
//     column tiles dealt to the four waves; lane (li, kq) builds the depthwise output of its mid pixel, channels 16g + 4kq .. +3, from K*K
//     LDS reads per group — the B fragment of v_mfma_f32_16x16x4_f32, the pointwise weights waiting in LDS as A fragments (or streamed from
//     L2 where 160 KB do not hold them: 80 channels at K = 5); BatchNorm + ReLU in registers, mid pixels OUTSIDE the image set to zero
//     (they are the second depthwise conv's zero padding);
//   * the mid tile is written over the input patch (dead by then: a barrier on each side) and half 2 + the epilogue are sepf.hip's.
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

struct SepwK {
  addk_src src; int N, H, W, C;
  const float* dw1; const float* pw1; const float* ma; const float* mb;
  const float* dw2; const float* pw2; int ldw;
  float* y; int ldy;
  const float* ea; const float* eb; int nterm; addk_src term[ADDK_MAX_TERMS];
  int tiles_x, tiles_y, gx;
};

__device__ __forceinline__ float4 fma4w(float4 w, float4 v, float4 a) {
  return make_float4(fmaf(w.x, v.x, a.x), fmaf(w.y, v.y, a.y), fmaf(w.z, v.z, a.z), fmaf(w.w, v.w, a.w));
}

template <int KS, int KG, int KP, bool PWL>
struct SepwGeo {
  static constexpr int CT = KG, HK = KS / 2, KQ = KP / 4;
  static constexpr int MH = 4 + 2 * HK, MW = 16 + 2 * HK, NM = MH * MW, NT1 = (NM + 15) / 16;      // mid region, its 16-pixel column tiles
  static constexpr int IH = 4 + 4 * HK, IW = 16 + 4 * HK, NI = IH * IW;                            // input patch
  static constexpr int T1W = (NT1 + 3) / 4;                                                         // mid tiles per wave
  static constexpr int PATCH = NI * KP + 8;
  static constexpr int DWL = KS * KS * KG * 16, PWLF = PWL ? KG * CT * 64 * 4 : 0;
  static constexpr int COEF = 6 * KG * 16;                                                          // (a0, b0), (a1, b1), (ea, eb)
  static constexpr int TLD = 64 * KP;                                                               // first sum term of the output tile
  static constexpr size_t LDS = (size_t)(PATCH + 2 * DWL + 2 * PWLF + COEF + TLD) * 4;
  static constexpr int NPL = 256 / KQ, NLD = (NI + NPL - 1) / NPL;
};

template <int KS, int KG, int KP, bool PWL>
__global__ void __launch_bounds__(256, 2) sepw_kernel(const SepwK p) {
  typedef SepwGeo<KS, KG, KP, PWL> G;
  constexpr int CT = G::CT, HK = G::HK, KQ = G::KQ, MW = G::MW, NM = G::NM, NT1 = G::NT1, IW = G::IW, NI = G::NI, T1W = G::T1W, NPL = G::NPL;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;                                  // [NI][KP] input patch, later [NM][KP] mid tile
  float* dwl1 = patch + G::PATCH;                     // [KS*KS][KG*16]
  float* dwl2 = dwl1 + G::DWL;
  float* pwl1 = dwl2 + G::DWL;                        // [KG][CT][64] float4 (PWL only)
  float* pwl2 = pwl1 + G::PWLF;
  float* coef = pwl2 + G::PWLF;                       // a0 b0 a1 b1 ea eb, KG*16 floats each
  float* tl = coef + G::COEF;                         // [64][KP]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int C = p.C, nq = C >> 2;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y; const int n = b / p.tiles_y;
  const int oh0 = ty * 4, ow0 = tx * 16;
  const int ih0 = oh0 - 2 * HK, iw0 = ow0 - 2 * HK;

  // ---- stage: every global load of the tile is requested before anything waits ----
  const int q = t % KQ, pl = t / KQ;
  const bool qact = pl < NPL && q < nq;
  const long xq = qact ? 4 * q : 0;
  constexpr int UN = 8;                               // patch loads in flight per thread and round
  float4 pv[UN]; bool pok[UN];
  auto patch_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int pix = base + u * NPL;
      const int pr = pix / IW, pc = pix - pr * IW;
      const int ih = ih0 + pr, iw = iw0 + pc;
      pok[u] = qact && pix < NI && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      pv[u] = ld4(p.src.x + xq + (pok[u] ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld : 0));
    }
  };
  patch_issue(pl);
  // coefficient vectors: thread c4 < 6 * nq loads quad (c4 % nq) of vector (c4 / nq)
  float4 cv = zero4();
  const int cvec = t / nq, cquad = t - cvec * nq;
  if (t < 6 * nq) {
    const float* v = cvec == 0 ? p.src.a : cvec == 1 ? p.src.b : cvec == 2 ? p.ma : cvec == 3 ? p.mb : cvec == 4 ? p.ea : p.eb;
    const float dflt = (cvec == 0 || cvec == 2 || cvec == 4) ? 1.f : 0.f;
    cv = make_float4(dflt, dflt, dflt, dflt);
    if (v) cv = ld4(v + 4 * cquad);
  }
  // first sum term of the output tile
  constexpr int NTL = (64 + NPL - 1) / NPL;
  float4 tq[NTL];
  const bool t0 = p.nterm > 0;
  if (t0) {
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
      const int pix = pl + u * NPL;
      const int oh = oh0 + (pix >> 4), ow = ow0 + (pix & 15);
      const bool ok = qact && pix < 64 && oh < p.H && ow < p.W;
      tq[u] = ld4(p.term[0].x + (ok ? ((long)(n * p.H + oh) * p.W + ow) * p.term[0].ld + 4 * q : 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // depthwise weights (both halves) -> [tap][channel], channels beyond C zero; pointwise weights -> A fragments
  for (int i = t; i < 2 * KS * KS * (KG * 16 - C); i += 256) {
    const int h = i / (KS * KS * (KG * 16 - C)), r = i - h * (KS * KS * (KG * 16 - C));
    const int tp = r / (KG * 16 - C), c = C + r - tp * (KG * 16 - C);
    (h ? dwl2 : dwl1)[tp * (KG * 16) + c] = 0.f;
  }
  for (int i = t; i < 2 * C * KS * KS; i += 256) {
    const int h = i / (C * KS * KS), r = i - h * (C * KS * KS);
    const int c = r / (KS * KS), tp = r - c * (KS * KS);
    (h ? dwl2 : dwl1)[tp * (KG * 16) + c] = ((const gfloat*)(h ? p.dw2 : p.dw1))[r];
  }
  if (PWL) {
    for (int s = t; s < 2 * KG * CT * 64; s += 256) {
      const int h = s / (KG * CT * 64), r = s - h * (KG * CT * 64);
      const int g = r / (CT * 64), rem = r - g * (CT * 64), ct = rem >> 6, ln = rem & 63;
      const int nn = ct * 16 + (ln & 15), k = 16 * g + 4 * (ln >> 4);
      const bool ok = nn < C && k < C;
      const float* w = h ? p.pw2 : p.pw1;
      float4 v = ld4(ok ? w + (long)nn * p.ldw + k : w);
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      lds_st4((h ? pwl2 : pwl1) + r * 4, v);
    }
  }
  if (t < 6 * nq) lds_st4(coef + cvec * (KG * 16) + 4 * cquad, cv);
  if (t0 && qact) {
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
      const int pix = pl + u * NPL;
      if (pix < 64) lds_st4(tl + pix * KP + 4 * q, tq[u]);
    }
  }
  __syncthreads();                                    // the prologue coefficients are in LDS
  {
    const bool relu = p.src.relu != 0;
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
    if (qact) { av = lds_ld4(coef + 4 * q); bv = lds_ld4(coef + KG * 16 + 4 * q); }
    if (pl < NPL) {
      for (int base = pl; base < NI; base += UN * NPL) {
        if (base != pl) patch_issue(base);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int pix = base + u * NPL;
          float4 z = fma4w(av, pv[u], bv);
          if (relu) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
          z.x = pok[u] ? z.x : 0.f; z.y = pok[u] ? z.y : 0.f; z.z = pok[u] ? z.z : 0.f; z.w = pok[u] ? z.w : 0.f;
          if (pix < NI) lds_st4(patch + pix * KP + 4 * q, z);       // (padding quads nq <= q < KQ get zeros: pok is false there)
        }
      }
    }
    if (t < 8) patch[NI * KP + t] = 0.f;
  }
  __syncthreads();

  // ---- half 1 on the mid region: wave w owns mid column tiles w, w + 4, ... ----
  float4 mid[T1W][CT];
#pragma unroll
  for (int u = 0; u < T1W; ++u) {
    const int tile = wave + 4 * u;
    const int j = 16 * tile + li;
    const bool jok = tile < NT1 && j < NM;
    const int jj = jok ? j : 0;
    const int mr = jj / MW, mc = jj - mr * MW;
    f32x4 macc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) macc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tile < NT1) {                                 // wave-uniform
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const int qg = 4 * g + kq;
        const int qr = qg < KQ ? qg : qg - 2;          // quads past the padded pixel: re-read a valid quad, its tap weights are 0
        float4 wf[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          if (PWL) wf[i] = lds_ld4(pwl1 + ((g * CT + i) * 64 + lane) * 4);
          else {
            const int nn = i * 16 + li, k = 16 * g + 4 * kq;
            const bool ok = nn < C && k < C;
            float4 v = ld4(ok ? p.pw1 + (long)nn * p.ldw + k : p.pw1);
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            wf[i] = v;
          }
        }
        float4 acc = zero4();
        const float* pb = patch + (mr * IW + mc) * KP + 4 * qr;
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
          for (int dx = 0; dx < KS; ++dx)
            acc = fma4w(lds_ld4(dwl1 + (dy * KS + dx) * (KG * 16) + 4 * qg), lds_ld4(pb + (dy * IW + dx) * KP), acc);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < CT; ++i)
            macc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(wf[i], e), get4(acc, e), macc[i], 0, 0, 0);
      }
    }
    // BatchNorm of half 1, ReLU; a mid pixel outside the image is the second depthwise conv's zero padding
    const int gh = oh0 - HK + mr, gw = ow0 - HK + mc;
    const bool inimg = jok && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int c = i * 16 + kq * 4;
      float4 v = make_float4(macc[i][0], macc[i][1], macc[i][2], macc[i][3]);
      if (c < C) {
        const float4 a1 = lds_ld4(coef + 2 * (KG * 16) + c), b1 = lds_ld4(coef + 3 * (KG * 16) + c);
        v = fma4w(a1, v, b1);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      mid[u][i] = inimg ? v : zero4();
    }
  }
  __syncthreads();                                    // every wave is done reading the input patch
#pragma unroll
  for (int u = 0; u < T1W; ++u) {
    const int tile = wave + 4 * u;
    const int j = 16 * tile + li;
    if (tile < NT1 && j < NM) {
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const int qm = 4 * i + kq;
        if (4 * qm < C) lds_st4(patch + j * KP + 4 * qm, mid[u][i]);
      }
    }
  }
  __syncthreads();

  // ---- half 2: wave = output row `wave` x 16 pixels of the tile, reading the mid tile [MH][MW] ----
  const int oh = oh0 + wave, ow = ow0 + li;
  const bool pin = oh < p.H && ow < p.W;
  const int pp = (n * p.H + oh) * p.W + ow;
  f32x4 oacc[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    const int qg = 4 * g + kq;
    const int qr = qg < KQ ? qg : qg - 2;
    float4 wf[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      if (PWL) wf[i] = lds_ld4(pwl2 + ((g * CT + i) * 64 + lane) * 4);
      else {
        const int nn = i * 16 + li, k = 16 * g + 4 * kq;
        const bool ok = nn < C && k < C;
        float4 v = ld4(ok ? p.pw2 + (long)nn * p.ldw + k : p.pw2);
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        wf[i] = v;
      }
    }
    float4 acc = zero4();
    const float* pb = patch + (wave * MW + li) * KP + 4 * qr;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx)
        acc = fma4w(lds_ld4(dwl2 + (dy * KS + dx) * (KG * 16) + 4 * qg), lds_ld4(pb + (dy * MW + dx) * KP), acc);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < CT; ++i)
        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(wf[i], e), get4(acc, e), oacc[i], 0, 0, 0);
  }
  // ---- epilogue: own frozen BatchNorm + the other branches of the cell block, as sepf.hip's inference epilogue ----
  if (pin) {
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int c = i * 16 + kq * 4;
      if (c >= C) continue;
      float4 v = make_float4(oacc[i][0], oacc[i][1], oacc[i][2], oacc[i][3]);
      if (p.ea) v = fma4w(lds_ld4(coef + 4 * (KG * 16) + c), v, lds_ld4(coef + 5 * (KG * 16) + c));
      if (t0) {
        const addk_src& T = p.term[0];
        const float4 u = prologue4(lds_ld4(tl + (wave * 16 + li) * KP + c), T.a, T.b, c, 4, T.relu != 0, true);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      for (int ti = 1; ti < p.nterm; ++ti) {
        const addk_src& T = p.term[ti];
        const float4 u = prologue4(ld4(T.x + (long)pp * T.ld + c), T.a, T.b, c, 4, T.relu != 0, true);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      st4_wt(p.y + (long)pp * p.ldy + c, v);
    }
  }
}

struct SepwCfg { int ks, kg, kp, pwl; };

bool sepw_fill(const addk_sepconv_args* a, SepwK& k, SepwCfg& c) {
  if (!a || !(a->K == 3 || a->K == 5) || a->N <= 0 || a->H <= 0 || a->W <= 0) return false;
  const addk_src& s = a->src;
  const int kg = cdiv(s.C, 16);
  if (!(kg == 3 || kg == 5) || !s.x || s.rs_hw || !src_vec_ok(s) || (s.a == nullptr) != (s.b == nullptr)) return false;
  if (!a->dw1_w || !a->pw1_w || !a->dw2_w || !a->pw2_w || !a->y || !aligned16(a->y) || a->ldy % 4 || a->ldy < s.C) return false;
  if (!aligned16(a->pw1_w) || !aligned16(a->pw2_w) || a->ldw % 4 || a->ldw < s.C) return false;
  if ((a->mid_a == nullptr) != (a->mid_b == nullptr) || (a->mid_a && (!aligned16(a->mid_a) || !aligned16(a->mid_b)))) return false;
  if ((a->ea == nullptr) != (a->eb == nullptr) || (a->ea && (!aligned16(a->ea) || !aligned16(a->eb)))) return false;
  if (a->nterm < 0 || a->nterm > ADDK_MAX_TERMS) return false;
  for (int i = 0; i < a->nterm; ++i) if (!a->term[i].x || a->term[i].C != s.C || a->term[i].rs_hw || !src_vec_ok(a->term[i])) return false;
  int kp = s.C; while (kp % 16 != 8) kp += 4;
  if (!((kg == 3 && (kp == 40 || kp == 56)) || (kg == 5 && (kp == 72 || kp == 88)))) return false;
  k = SepwK{};
  k.src = s; k.N = a->N; k.H = a->H; k.W = a->W; k.C = s.C;
  k.dw1 = a->dw1_w; k.pw1 = a->pw1_w; k.ma = a->mid_a; k.mb = a->mid_b; k.dw2 = a->dw2_w; k.pw2 = a->pw2_w; k.ldw = a->ldw;
  k.y = a->y; k.ldy = a->ldy; k.ea = a->ea; k.eb = a->eb; k.nterm = a->nterm;
  for (int i = 0; i < a->nterm; ++i) k.term[i] = a->term[i];
  k.tiles_x = cdiv(a->W, 16); k.tiles_y = cdiv(a->H, 4); k.gx = a->N * k.tiles_y * k.tiles_x;
  c.ks = a->K; c.kg = kg; c.kp = kp;
  c.pwl = a->K != 5;             // K = 5: the input patch (101 KB at 80 channels; 46 KB at 40, where two workgroups must share a CU) leaves no room for the pointwise weights
  return true;
}

template <int KS, int KG, int KP, bool PWL>
int sepw_go(dim3 grid, hipStream_t st, const SepwK& k) {
  typedef SepwGeo<KS, KG, KP, PWL> G;
  static_assert(G::LDS <= 160 * 1024, "sepw: LDS budget");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepw_kernel<KS, KG, KP, PWL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS); attr = true; }
  hipLaunchKernelGGL((sepw_kernel<KS, KG, KP, PWL>), grid, dim3(256), G::LDS, st, k);
  return addk_check_launch("sepconv_fwd");
}

}  // namespace

extern "C" int addk_sepconv_fwd_supported(const addk_sepconv_args* a) {
  SepwK k; SepwCfg c;
  return (addk_get_fast_paths() & ADDK_FAST_PW) && sepw_fill(a, k, c) ? 1 : 0;
}
extern "C" int addk_sepconv_fwd(const addk_sepconv_args* a, void* stream) {
  SepwK k; SepwCfg c;
  ADDK_REQUIRE(sepw_fill(a, k, c), "sepconv_fwd: shape not covered (K in {3,5}, channels in (32,48] or (64,80], 16-byte aligned tensors)");
  const dim3 grid(k.gx);
  hipStream_t st = (hipStream_t)stream;
#define ADDK_SEPW(KS_, KG_, KP_, PWL_) if (c.ks == KS_ && c.kg == KG_ && c.kp == KP_ && c.pwl == (PWL_ ? 1 : 0)) return sepw_go<KS_, KG_, KP_, PWL_>(grid, st, k);
  ADDK_SEPW(3, 3, 40, true) ADDK_SEPW(5, 3, 40, false) ADDK_SEPW(3, 3, 56, true) ADDK_SEPW(5, 3, 56, false)
  ADDK_SEPW(3, 5, 72, true) ADDK_SEPW(3, 5, 88, true) ADDK_SEPW(5, 5, 72, false) ADDK_SEPW(5, 5, 88, false)
#undef ADDK_SEPW
  addk_set_error("sepconv_fwd: no instantiation");
  return ADDK_ERR_UNSUPPORTED;
}
