// Fused BACKWARD of a SepConv half (reference modeling/operations.py:51-53 / :55-57, autograd of  y = pw(dw(relu(a x + b)))):
//   dt = W_pw^T dy                      (data gradient of the pointwise 1x1: fp32 matrix cores)
//   dz = dw_flipped * dt                (data gradient of the depthwise KS x KS)
//   dW_dw[c][tap] = sum_p dt[o(p, tap)] z[p],   (dA, dB) = sum_p (m dz x, m dz),   dx (+)= a m dz      (m = ReLU mask)
// in ONE launch over the same 2-D tiles as the forward kernel (sepf.hip).  Round 2 ran two launches (pw data gradient 23-60 us,
// depthwise backward 25-38 us per level batch) with the 10 MB gradient dt written and re-read in between: 6.5 ms of the 37 ms
// step.  Here dt never leaves the chip:
//   * stage 1: the workgroup computes dt on its haloed patch [(4R+KS-1)][16+KS-1] — every wave walks 16-pixel groups of the
//     patch, loads dy straight into the MFMA B fragment (lane (li, kq): pixel li, channels 16g + 4kq .. +3), the transposed
//     pointwise weights wait in LDS as ready A fragments — and drops the result into the LDS patch [pixel][KP] whose stride
//     KP = 8 (mod 16) makes the tap reads of stage 2 conflict-free (sepf.hip);
//   * stage 2: lane (li, kq) owns R input pixels x 4 channels of group g: per kernel row it reads KS shifted dt quads, feeds dz
//     and the KS weight-gradient products, and reduces the latter over the 16 pixel lanes with four DPP row adds (fixed
//     order: bit-reproducible); per-wave partials meet in LDS, one row [C][KS*KS] of the weight-gradient workspace and one row
//     [C][2] of the (dA, dB) slab per workgroup.
// The pointwise WEIGHT gradient stays with the batched register-streaming kernel (wgrad.hip), which reads dy and the stored
// depthwise output.
#include <stdlib.h>
#include "common.h"

namespace {

struct SepbK {
  const float* dy; int lddy;
  addk_src src; int N, H, W, C;
  const float* dww; const float* pww; int ldw;
  float* g; int ldg; int accumulate;
  double* dab; float* ws;
  int tiles_x, tiles_y, gx;
};

__device__ __forceinline__ float4 fma4b(float4 w, float4 v, float4 a) {
  return make_float4(fmaf(w.x, v.x, a.x), fmaf(w.y, v.y, a.y), fmaf(w.z, v.z, a.z), fmaf(w.w, v.w, a.w));
}
// sum over the 16 lanes of a DPP row (= the 16 pixel lanes li of one channel quad), same value in every lane, fixed order
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

template <int KS, int KG, int KP, int R>
struct SepbGeo {
  static constexpr int CT = KG, PH = 4 * R + KS - 1, PW = 16 + KS - 1, NPIX = PH * PW, KQ = KP / 4, NT16 = (NPIX + 15) / 16;
  static constexpr int PATCH = NT16 * 16 * KP + 8;
  static constexpr int DWL = KS * KS * KG * 16, PWL = KG * CT * 64 * 4;
  static constexpr int DWS = 4 * KS * KS * KG * 16;                // per-wave weight-gradient partials [4][KS*KS][KG*16]
  static constexpr int RED = 4 * KG * 16 * 2 * 2;                  // floats: [4][KG*16][2] doubles
  static constexpr size_t LDS = (size_t)(PATCH + DWL + PWL + DWS + RED) * 4;
};

template <int KS, int KG, int KP, int R>
__device__ __forceinline__ void sepb_body(const SepbK& p, float* sm) {
  typedef SepbGeo<KS, KG, KP, R> G;
  constexpr int CT = G::CT, PW = G::PW, NPIX = G::NPIX, KQ = G::KQ, HK = KS / 2, NT = KS * KS;
  float* patch = sm;
  float* dwl = patch + G::PATCH;                 // [NT][KG*16] tap weights, FLIPPED: dwl[f] = w[NT-1-f]
  float* pwl = dwl + G::DWL;                     // [KG][CT][64] float4: A fragments of W^T
  float* dws = pwl + G::PWL;                     // [4][NT][KG*16]
  double* red = reinterpret_cast<double*>(dws + G::DWS);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int C = p.C;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y; const int n = b / p.tiles_y;
  const int ih0 = ty * (4 * R), iw0 = tx * 16;

  // ---- [r4] the first stage-1 tile's dy quads are requested before anything else: they travel while the weights are staged ----
  auto dy_geom = [&](int j, const float*& dp, bool& ok) {
    const int pix = 16 * j + li;
    const int pr = pix / PW, pc = pix - pr * PW;
    const int oh = ih0 - HK + pr, ow = iw0 - HK + pc;
    ok = j < G::NT16 && pix < NPIX && (unsigned)oh < (unsigned)p.H && (unsigned)ow < (unsigned)p.W;
    dp = p.dy + (ok ? ((long)(n * p.H + oh) * p.W + ow) * p.lddy : 0);
  };
  auto dy_load = [&](int j, float4 (&d)[KG]) {
    const float* dp; bool ok;
    dy_geom(j, dp, ok);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int k = 16 * g + 4 * kq;
      d[g] = ld4(dp + ((ok && k < C) ? k : 0));
    }
  };
  float4 dcur[KG], dnxt[KG];
  dy_load(wave, dcur);
  __builtin_amdgcn_sched_barrier(0);
  // ---- weights -> LDS (the tap table's channels beyond C are zeroed by threads that write no weight there: no barrier in between) ----
  for (int i = t; i < NT * (KG * 16 - C); i += 256) {
    const int tp = i / (KG * 16 - C), c = C + i - tp * (KG * 16 - C);
    dwl[tp * (KG * 16) + c] = 0.f;
  }
  for (int i = t; i < C * NT; i += 256) {
    const int c = i / NT, tp = i - c * NT;
    dwl[(NT - 1 - tp) * (KG * 16) + c] = ((const gfloat*)p.dww)[i];
  }
  for (int s = t; s < KG * CT * 64; s += 256) {          // A[row = ci = 16 i + (ln & 15)][k = co = 16 g + 4 (ln >> 4) + e] = W[co][ci]
    const int g = s / (CT * 64), rem = s - g * (CT * 64), i = rem >> 6, ln = rem & 63;
    const int ci = i * 16 + (ln & 15), co = 16 * g + 4 * (ln >> 4);
    const bool ok = ci < C && co < C;
    const gfloat* wp = (const gfloat*)(p.pww + (ok ? (long)co * p.ldw + ci : 0));
    const long st = ok ? p.ldw : 0;
    float4 v = make_float4(wp[0], wp[st], wp[2 * st], wp[3 * st]);
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    lds_st4(pwl + s * 4, v);
  }
  if (KQ > C / 4) {                                      // zero the padding quads of every patch pixel (never written by stage 1)
    for (int i = t; i < G::NT16 * 16 * (KQ - C / 4); i += 256) {
      const int px = i / (KQ - C / 4), q = C / 4 + i - px * (KQ - C / 4);
      lds_st4(patch + px * KP + 4 * q, zero4());
    }
  }
  if (t < 8) patch[G::NT16 * 16 * KP + t] = 0.f;
  __syncthreads();

  // ---- stage 1: dt = W^T dy on the haloed patch, 16 pixels at a time ----
  for (int j = wave; j < G::NT16; j += 4) {
    const int pix = 16 * j + li;
    const float* dp; bool ok;
    dy_geom(j, dp, ok);
    dy_load(j + 4, dnxt);                                  // the next tile of this wave: in flight under this tile's matrix work (masked beyond the patch)
    float4 d[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int k = 16 * g + 4 * kq;
      const bool okk = ok && k < C;
      float4 v = dcur[g];
      v.x = okk ? v.x : 0.f; v.y = okk ? v.y : 0.f; v.z = okk ? v.z : 0.f; v.w = okk ? v.w : 0.f;
      d[g] = v;
      dcur[g] = dnxt[g];
    }
    f32x4 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      float4 wf[CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) wf[i] = lds_ld4(pwl + ((g * CT + i) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < CT; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(wf[i], e), get4(d[g], e), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int q = 4 * i + kq;                          // lane holds channels 16 i + 4 kq + {0..3} of pixel `pix`
      if (q < KQ) lds_st4(patch + pix * KP + 4 * q, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));
    }
  }
  __syncthreads();

  // ---- stage 2: depthwise backward; wave = input rows [wave*R, wave*R + R) x 16 pixels ----
  int pp[R]; bool pin[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ih = ih0 + wave * R + r, iw = iw0 + li;
    pin[r] = ih < p.H && iw < p.W;
    pp[r] = (n * p.H + ih) * p.W + iw;
  }
  const bool relu = p.src.relu != 0;
  float* mydws = dws + wave * (NT * KG * 16);
  double (*rd)[KG * 16][2] = reinterpret_cast<double (*)[KG * 16][2]>(red);
  // real loops over the channel group and the kernel row: fully unrolled, hipcc hoists every LDS read of the 15 (group, row)
  // bodies to the top (234-256 VGPRs and scratch spills for KS = 5); one body at a time needs ~100
  // [r4] the operands of the NEXT channel group (input values, the gradient to accumulate into) are requested at the top of a group's
  // body and arrive under its LDS / VALU work: the loop used to open with a dependent round trip per group and close with another
  float4 xn[R], on[R];
  auto pre = [&](int g) {
    const int q = 4 * g + kq;
    const bool cok = g < KG && 4 * q < C;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool okx = pin[r] && cok;
      xn[r] = ld4(p.src.x + (okx ? (long)pp[r] * p.src.ld + 4 * q : 0));
      on[r] = zero4();
      if (p.g && p.accumulate) on[r] = ld4(p.g + (okx ? (long)pp[r] * p.ldg + 4 * q : 0));
    }
  };
  pre(0);
#pragma unroll 1
  for (int g = 0; g < KG; ++g) {
    const int q = 4 * g + kq;
    const int qr = q < KQ ? q : q - 2;
    const bool cok = 4 * q < C;
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
    if (p.src.a && cok) { av = ld4(p.src.a + 4 * q); bv = ld4(p.src.b + 4 * q); }
    float4 x[R], z[R], dz[R], oacc[R]; bool m[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = xn[r]; oacc[r] = on[r]; }
    pre(g + 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool okx = pin[r] && cok;
      const float4 zp = fma4b(av, x[r], bv);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[r][e] = okx && (!relu || get4(zp, e) > 0.f);
      z[r] = make_float4(m[r][0] ? zp.x : 0.f, m[r][1] ? zp.y : 0.f, m[r][2] ? zp.z : 0.f, m[r][3] ? zp.w : 0.f);
      if (!relu) z[r] = make_float4(okx ? zp.x : 0.f, okx ? zp.y : 0.f, okx ? zp.z : 0.f, okx ? zp.w : 0.f);
      dz[r] = zero4();
    }
    const float* pb = patch + ((wave * R) * PW + li) * KP + 4 * qr;
#pragma unroll 1
    for (int fr = 0; fr < KS; ++fr) {                    // flipped kernel row: patch row i = fr + r feeds input row r
      float4 wr[KS], dwa[KS];
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) { wr[dx] = lds_ld4(dwl + (fr * KS + dx) * (KG * 16) + 4 * q); dwa[dx] = zero4(); }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          const float4 v = lds_ld4(pb + ((fr + r) * PW + dx) * KP);
          dz[r] = fma4b(wr[dx], v, dz[r]);
          dwa[dx] = fma4b(v, z[r], dwa[dx]);
        }
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        float4 s;
        s.x = row_sum16(dwa[dx].x); s.y = row_sum16(dwa[dx].y); s.z = row_sum16(dwa[dx].z); s.w = row_sum16(dwa[dx].w);
        if (li == 0) lds_st4(mydws + (fr * KS + dx) * (KG * 16) + 4 * q, s);     // flipped tap index f = fr*KS + dx
      }
    }
    double sA[4], sB[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sA[e] = 0.0; sB[e] = 0.0; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 gm;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = m[r][e] ? get4(dz[r], e) : 0.f;
        set4(gm, e, d);
        sA[e] += (double)d * (double)get4(x[r], e); sB[e] += (double)d;
      }
      if (p.g && pin[r] && cok) {
        float4 gv = make_float4(gm.x * av.x, gm.y * av.y, gm.z * av.z, gm.w * av.w);
        float* gp = p.g + (long)pp[r] * p.ldg + 4 * q;
        if (p.accumulate) { const float4 o = oacc[r]; gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
        st4(gp, gv);
      }
    }
    if (p.dab) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double a2 = sA[e], b2 = sB[e];
#pragma unroll
        for (int mk = 1; mk < 16; mk <<= 1) { a2 += __shfl_xor(a2, mk); b2 += __shfl_xor(b2, mk); }
        if (li == 0) { rd[wave][4 * q + e][0] = a2; rd[wave][4 * q + e][1] = b2; }
      }
    }
  }
  __syncthreads();
  // ---- one workspace row per workgroup: the four waves in fixed order; the flipped tap index goes back to [c][tap] ----
  for (int i = t; i < C * NT; i += 256) {
    const int c = i / NT, tp = i - c * NT;
    const int o = (NT - 1 - tp) * (KG * 16) + c;
    const float s = (dws[o] + dws[NT * KG * 16 + o]) + (dws[2 * NT * KG * 16 + o] + dws[3 * NT * KG * 16 + o]);
    ((gfloat*)p.ws)[((long)blockIdx.x * C + c) * NT + tp] = s;
  }
  if (p.dab && t < C) {
    gdouble* o = (gdouble*)p.dab + ((long)blockIdx.x * C + t) * 2;
    o[0] = (rd[0][t][0] + rd[1][t][0]) + (rd[2][t][0] + rd[3][t][0]);
    o[1] = (rd[0][t][1] + rd[1][t][1]) + (rd[2][t][1] + rd[3][t][1]);
  }
}

template <int KS, int KG, int KP, int R>
__global__ void __launch_bounds__(256, 2) sepb_kernel(const SepbK p) {
  extern __shared__ __attribute__((aligned(16))) float sepb_sm[];
  sepb_body<KS, KG, KP, R>(p, sepb_sm);
}
template <int KS, int KG, int KP, int R>
__global__ void __launch_bounds__(256, 2) sepb_batch_kernel(const SepbK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float sepb_sm[];
  const SepbK p = tab[blockIdx.z];
  if ((int)blockIdx.x >= p.gx) return;
  sepb_body<KS, KG, KP, R>(p, sepb_sm);
}

struct SepbCfg { int ks, kg, kp, r; };
inline int sepb_key(const SepbCfg& c) { return (c.ks << 16) | (c.kg << 12) | (c.kp << 4) | c.r; }

bool sepb_fill(const addk_sep_bwd_args* a, SepbK& k, SepbCfg& c) {
  if (!a || !(a->K == 3 || a->K == 5) || a->N <= 0 || a->H <= 0 || a->W <= 0) return false;
  const addk_src& s = a->src;
  const int kg = cdiv(s.C, 16);
  if (!(kg == 3 || kg == 5) || a->Cout != s.C || !s.x || !src_vec_ok(s) || !a->dw_w || !a->pw_w || !a->dy || !a->ws) return false;
  if (!aligned16(a->dy) || a->lddy % 4 || a->lddy < a->Cout || a->ldw < s.C) return false;
  if (a->g && (!aligned16(a->g) || a->ldg % 4 || a->ldg < s.C)) return false;
  if ((long)a->N * a->H * a->W >= (1L << 30)) return false;
  int kp = s.C; while (kp % 16 != 8) kp += 4;
  if (!((kg == 3 && (kp == 40 || kp == 56)) || (kg == 5 && (kp == 72 || kp == 88)))) return false;
  k = SepbK{};
  k.dy = a->dy; k.lddy = a->lddy; k.src = s; k.N = a->N; k.H = a->H; k.W = a->W; k.C = s.C;
  k.dww = a->dw_w; k.pww = a->pw_w; k.ldw = a->ldw; k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate;
  k.dab = (double*)a->dab; k.ws = a->ws;
  const long blocks2 = (long)a->N * cdiv(a->H, 8) * cdiv(a->W, 16);
  // 80-channel tiles need 100-127 KB of LDS: one workgroup per CU.  That is fine while the launch has at most two rounds of them
  // (config 2: 256 workgroups at 64x128) and LOSES to the separate depthwise / pointwise launches beyond (F = 40, 80 channels at
  // 128x256 = 1024 workgroups: step 72.2 ms fused vs 66.5 ms unfused) — those shapes stay on the unfused kernels
  if (kg == 5 && (long)a->N * cdiv(a->H, 4) * cdiv(a->W, 16) > 512) return false;
  c.ks = a->K; c.kg = kg; c.kp = kp; c.r = (kg == 3 && blocks2 >= 384) ? 2 : 1;
  k.tiles_x = cdiv(a->W, 16); k.tiles_y = cdiv(a->H, 4 * c.r); k.gx = a->N * k.tiles_y * k.tiles_x;
  return true;
}

template <int KS, int KG, int KP, int R>
int sepb_go(bool batch, dim3 grid, hipStream_t st, const SepbK* one, const SepbK* tab) {
  typedef SepbGeo<KS, KG, KP, R> G;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepb_kernel<KS, KG, KP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepb_batch_kernel<KS, KG, KP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    attr = true;
  }
  if (batch) hipLaunchKernelGGL((sepb_batch_kernel<KS, KG, KP, R>), grid, dim3(256), G::LDS, st, tab);
  else hipLaunchKernelGGL((sepb_kernel<KS, KG, KP, R>), grid, dim3(256), G::LDS, st, *one);
  return addk_check_launch("sep_bwd");
}

int sepb_dispatch(const SepbCfg& c, bool batch, dim3 grid, hipStream_t st, const SepbK* one, const SepbK* tab) {
#define ADDK_SEPB(KS_, KG_, KP_, R_) if (c.ks == KS_ && c.kg == KG_ && c.kp == KP_ && c.r == R_) \
    return sepb_go<KS_, KG_, KP_, R_>(batch, grid, st, one, tab);
  ADDK_SEPB(3, 3, 40, 1) ADDK_SEPB(3, 3, 40, 2) ADDK_SEPB(5, 3, 40, 1) ADDK_SEPB(5, 3, 40, 2)
  ADDK_SEPB(3, 3, 56, 1) ADDK_SEPB(3, 3, 56, 2) ADDK_SEPB(5, 3, 56, 1) ADDK_SEPB(5, 3, 56, 2)
  ADDK_SEPB(3, 5, 72, 1) ADDK_SEPB(5, 5, 72, 1) ADDK_SEPB(3, 5, 88, 1) ADDK_SEPB(5, 5, 88, 1)
#undef ADDK_SEPB
  addk_set_error("sep_bwd: no instantiation");
  return ADDK_ERR_UNSUPPORTED;
}

}  // namespace

// rows of `ws` ([rows][C][K*K] floats) and `dab` ([rows][C][2] fp64) the launch writes: one per workgroup; 0: shape not covered
extern "C" int addk_sep_bwd_rows(const addk_sep_bwd_args* a) {
  SepbK k; SepbCfg c;
  if (!a) return 0;
  addk_sep_bwd_args b = *a;
  if (!b.ws) b.ws = reinterpret_cast<float*>(16);           // geometry query before the workspace exists
  return (addk_get_fast_paths() & ADDK_FAST_PW) && sepb_fill(&b, k, c) ? k.gx : 0;
}
extern "C" int addk_sep_bwd(const addk_sep_bwd_args* a, void* stream) {
  SepbK k; SepbCfg c;
  ADDK_REQUIRE(sepb_fill(a, k, c), "sep_bwd: shape not covered (K in {3,5}, C == Cout in (32,48] or (64,80], aligned)");
  return sepb_dispatch(c, false, dim3(k.gx), (hipStream_t)stream, &k, nullptr);
}
extern "C" int addk_sep_bwd_batch_key(const addk_sep_bwd_args* a) {
  SepbK k; SepbCfg c;
  if (!(addk_get_fast_paths() & ADDK_FAST_PW) || !sepb_fill(a, k, c)) return -1;
  return sepb_key(c);
}
extern "C" int64_t addk_sep_bwd_batch_prepare(const addk_sep_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  if (!a || n <= 0 || !meta) { addk_set_error("sep_bwd_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  const int64_t total = (int64_t)n * sizeof(SepbK);
  if (host_blob && blob_bytes < total) { addk_set_error("sep_bwd_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  int key0 = -1, gx = 0;
  for (int i = 0; i < n; ++i) {
    SepbK k; SepbCfg c;
    if (!sepb_fill(&a[i], k, c)) { addk_set_error("sep_bwd_batch_prepare: launch %d is not covered", i); return ADDK_ERR_INVALID; }
    const int key = sepb_key(c);
    if (i == 0) key0 = key;
    if (key != key0) { addk_set_error("sep_bwd_batch_prepare: mixed kernel variants"); return ADDK_ERR_INVALID; }
    if (k.gx > gx) gx = k.gx;
    if (host_blob) reinterpret_cast<SepbK*>(host_blob)[i] = k;
  }
  meta[0] = key0; meta[1] = n; meta[2] = gx; meta[3] = 1;
  return total;
}
extern "C" int addk_sep_bwd_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[1] > 0 && meta[2] > 0, "sep_bwd_batch_run: bad args");
  const int key = (int)meta[0];
  SepbCfg c{(key >> 16) & 15, (key >> 12) & 15, (key >> 4) & 255, key & 15};
  return sepb_dispatch(c, true, dim3((unsigned)meta[2], 1, (unsigned)meta[1]), (hipStream_t)stream, nullptr, reinterpret_cast<const SepbK*>(dev_blob));
}
