// conv3n.hip — the split-bf16 halo-patch convolution for AT MOST 48 OUTPUT CHANNELS (the cells' dil_conv_3x3 / dil_conv_5x5 at 40 channels,
// operations.py:32-43 at F = 20, level 1; forward and data gradient) on 16-wide channel tiles: v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel.  conv3b_kernel's tiles are 32 output channels x 32 pixels x 16 input channels per matrix instruction: 40 output channels pad to 64 and
// 40 input channels to 48 — 52 % of its matrix work is useful, and the matrix phase is 63-73 % of those launches' life (profiles/r05_c3b_ablation.txt).  Here:
//   * output channels in three 16-row tiles: 40 -> 48;
//   * the K = 32 of one instruction spans TWO TAPS x 16 channels of a chunk (lane groups 0-1 read tap A's two 8-channel halves, groups 2-3 tap B's), and a
//     chunk with at most 8 valid channels (40 = 16 + 16 + 8) spans FOUR TAPS x 8 channels: 13 + 13 + 7 = 33 K-steps for a 5x5 where conv3b runs 75 taps
//     of K = 16 — 0.66 x the matrix cycles per pixel;
//   * a wave owns 64 pixels x ALL 48 channels (12 accumulator tiles): a weight fragment set (9 x 16 bytes per lane) serves 72 matrix instructions, the same
//     L1 bytes per matrix cycle as conv3b; the four waves of a workgroup are the two rows x two 64-pixel halves of a 2 x 128-pixel tile (rows d apart, as in
//     conv3b's two-row tiles);
//   * LDS image per plane [row][pixel][16 ch] bf16, LINEAR: the 16x16x32 fragment read (16 consecutive pixels x one 16-byte half per 16-lane group, two
//     groups per tap) touches 16 distinct 16-byte slots of the 256-byte bank row for any pixel shift — no swizzle needed.
// Staging (lazy BatchNorm / ReLU prologue, zero padding, checkerboard sign, split into bf16 planes), statistics (reduce-scatter in registers, rs16 below),
// two-pass data-gradient epilogue and blocked accumulation of the 3x3 are conv3b's (conv3b.h).  Weight packs: conv3.hip c3n_pack_body.
#include "conv3b.h"

namespace {

constexpr int CN_CT = 3;                    // 16-channel output tiles
constexpr int CN_BPX = 128;                 // pixels per tile row
constexpr int cn_pwp(int ks) { return CN_BPX + (ks - 1) * 2; }       // LDS row pitch in pixels (dilation <= 2)

// reduce-scatter of 16 per-lane values over the 16 lanes of a DPP row (lane bits 3..0): lane l ends with the row's sum of value l & 15 in v[0]
template <int N, int M>
__device__ __forceinline__ void rs16_step(float (&v)[16], const int l16) {
  const bool b = (l16 & M) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float keep = b ? v[i + N] : v[i], send = b ? v[i] : v[i + N];
    v[i] = keep + __shfl_xor(send, M);
  }
}
__device__ __forceinline__ void rs16(float (&v)[16], const int l16) {
  rs16_step<8, 8>(v, l16); rs16_step<4, 4>(v, l16); rs16_step<2, 2>(v, l16); rs16_step<1, 1>(v, l16);
}

template <int KS, int MODE, int NP>
__global__ void __launch_bounds__(256, 1) conv3n_kernel(const C3K p) {
  constexpr int TAPS = KS * KS, HK = KS / 2, NPAIR = (TAPS + 1) / 2, NQUAD = (TAPS + 3) / 4;
  constexpr int PR_ = KS + 1;                                      // patch rows of a two-row tile
  constexpr int PWP = cn_pwp(KS);
  constexpr int NS = (PR_ * PWP * 4 + 255) / 256;                  // 16-byte (4-channel) patch slots per thread
  static_assert(NS <= 32, "slot mask is 32 bits");
  constexpr int PLANE = PR_ * PWP * 2;                             // uint4 units per plane
  constexpr bool BLK = KS == 3;                                    // blocked accumulation, as conv3b's narrow 3x3 (the 5x5 keeps one set there too)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red = reinterpret_cast<double*>(smem);                  // [4 waves][48][2]
  unsigned* wmax = reinterpret_cast<unsigned*>(smem);             // NP = 2 (split-fp16, conv3b.h): the waves' largest staged magnitudes (aliases red[], written once at the very end)
  uint4* Pl = reinterpret_cast<uint4*>(smem + 4 * 48 * 16);
  uint2* Pl2 = reinterpret_cast<uint2*>(Pl);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, g = lane >> 4;
  const int jr = wave >> 1, wh = wave & 1;                        // this wave's tile row and 64-pixel half
  const int bx = blockIdx.x, gx = gridDim.x;
  const int d = p.dil;
  double tot_a = 0.0, tot_b = 0.0;
  const int q = t & 3;
  auto slot_geo = [&](int k, int& r, int& sp, bool& live) {
    const int pix = (t + 256 * k) >> 2;
    r = pix / PWP; sp = pix - r * PWP;
    live = r < PR_ && sp < CN_BPX + (KS - 1) * p.dil;
  };
  unsigned pmask = 0;                         // bit k: parity of (r*d + sp) of this thread's patch slot k (checkerboard sign)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int r, sp; bool live;
    slot_geo(k, r, sp, live);
    pmask |= (unsigned)((r * d + sp) & 1) << k;
  }
  // fragment read bases (uint4 units) per kernel column: pixel l16 + 64 wh + kw d of patch row jr; pair steps add this lane's 8-channel half (g & 1)
  int xbp[KS], xbq[KS];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xbq[kw] = ((jr * PWP) + 64 * wh + l16 + kw * d) * 2;
    xbp[kw] = xbq[kw] + (g & 1);
  }
  const uint4* wpl = reinterpret_cast<const uint4*>(p.wp) + lane;
  const float winv = NP == 2 ? p.wsc[0] : 1.f;

  const int tpx = p.ntiles >> 3;
  const bool swz = (p.ntiles & 7) == 0 && p.ntiles >= 64;
  for (int tlin = bx; tlin < p.ntiles; tlin += gx) {
    const int tile = swz ? (tlin & 7) * tpx + (tlin >> 3) : tlin;
    const int rowid = tile / p.spr, sx = tile - rowid * p.spr;
    const int n = rowid / p.HT, prq = rowid - n * p.HT;
    const int oh = (prq / d) * (2 * d) + prq % d;
    const int ow0 = sx * CN_BPX;
    unsigned vmask = 0;
    const int pbase = (n * p.IH + oh - HK * d) * p.IW + ow0 - HK * d;
    const unsigned par0 = (unsigned)(oh + ow0);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int r, sp; bool live;
      slot_geo(k, r, sp, live);
      const int ih = oh + (r - HK) * d, iw = ow0 - HK * d + sp;
      const bool ok = live && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      vmask |= (ok ? 1u : 0u) << k;
    }
    f32x4 acc[CN_CT][4], acc2[BLK ? CN_CT : 1][4];
#pragma unroll
    for (int i = 0; i < CN_CT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (BLK) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    float4 ra[NS];
    float4 pa = make_float4(1.f, 1.f, 1.f, 1.f), pb = zero4();
    bool prelu = false, pch = false, ptail = false;
    int kf = 0;                                   // NP = 2: exponent field of the tile's running operand scale (conv3b.h)
    auto load_patch = [&](int s_, int c0_) {      // branch-free: masked slots read the source base and are zeroed at store time
      const addk_src S = p.src[s_];
      const int c = c0_ + 4 * q;
      pch = c < S.C;
      ptail = S.C - c0_ <= 8;                      // a chunk of <= 8 channels is read by quad steps, which touch the first 8-channel half only
      prelu = S.relu != 0;
      pa = make_float4(1.f, 1.f, 1.f, 1.f); pb = zero4();
      if (S.a && pch) { pa = ld4(S.a + c); pb = ld4(S.b + c); }
      const float* sb = S.x + (pch ? c : 0);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        int r, sp; bool live;
        slot_geo(k, r, sp, live);
        const int po = ((vmask >> k) & 1u) ? pbase + r * d * p.IW + sp : 0;
        ra[k] = ld4(sb + (long)po * S.ld);
      }
    };
    auto prologue = [&](int k) {
      float4 v = ra[k];
      v.x = fmaf(pa.x, v.x, pb.x); v.y = fmaf(pa.y, v.y, pb.y); v.z = fmaf(pa.z, v.z, pb.z); v.w = fmaf(pa.w, v.w, pb.w);
      if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const bool ok = pch && ((vmask >> k) & 1u);
      const float sg = ok ? ((((pmask >> k) ^ par0) & 1u) ? -1.f : 1.f) : 0.f;
      v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
      return v;
    };
    // NP = 2 (split-fp16): prologue before the barrier in front of the staging, the wave's largest magnitude to wmax[]; behind the barrier every thread folds the
    // four maxima into the tile's running scale (conv3b.h)
    auto prep_patch = [&]() {
      unsigned mx = 0;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const float4 v = prologue(k);
        ra[k] = v;
        const unsigned b = absbits4(v);
        mx = b > mx ? b : mx;
      }
      mx = wave_umax(mx);
      if (lane == 0) wmax[wave] = mx;
    };
    auto update_scale = [&]() {
      unsigned m = wmax[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const unsigned b = wmax[w]; m = b > m ? b : m; }
      const int want = f16_scale_field(m);
      if (kf != 0 && want < kf) {
        const int rf = 127 + want - kf;
        const float r = rf > 0 ? __uint_as_float((unsigned)rf << 23) : 0.f;
#pragma unroll
        for (int i = 0; i < CN_CT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[i][j] *= r; if (BLK) acc2[i][j] *= r; }
      }
      if (kf == 0 || want < kf) kf = want;
    };
    auto store_patch = [&]() {                    // prologue, zero padding, checkerboard sign, split into planes (linear image)
      const float sc = NP == 2 ? __uint_as_float((unsigned)kf << 23) : 1.f;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 v;
        if (NP == 2) { v = ra[k]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        else v = prologue(k);
        int r, sp; bool live;
        slot_geo(k, r, sp, live);
        if (r < PR_ && !(ptail && q >= 2)) {
          uint2 pl[NP];
          split4<NP>(v, pl);
          const int slot = (r * PWP + sp) * 2 + (q >> 1);
#pragma unroll
          for (int m = 0; m < NP; ++m) Pl2[(m * PLANE + slot) * 2 + (q & 1)] = pl[m];
        }
      }
    };
    // weight fragments of K-step T: CN_CT tiles x NP planes x 16 bytes per lane
    auto load_w = [&](int T, uint4 (&dst)[CN_CT][NP]) {
      const int Tc = T < p.nT ? T : p.nT - 1;
      const uint4* src = wpl + (long)Tc * (CN_CT * NP * 64);
#pragma unroll
      for (int i = 0; i < CN_CT; ++i)
#pragma unroll
        for (int m = 0; m < NP; ++m) dst[i][m] = src[(i * NP + m) * 64];
    };
    auto mma = [&](f32x4& c, const uint4* w, const uint4* x) {
      if constexpr (NP == 2) {          // split-fp16: l*wh + h*wl + h*wh
        auto Wh = [&](int m) { return __builtin_bit_cast(f16x8, w[m]); };
        auto Xh = [&](int m) { return __builtin_bit_cast(f16x8, x[m]); };
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh(1), Xh(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh(0), Xh(1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh(0), Xh(0), c, 0, 0, 0);
        return;
      }
      auto W = [&](int m) { return __builtin_bit_cast(bf16x8, w[m]); };
      auto X = [&](int m) { return __builtin_bit_cast(bf16x8, x[m]); };
      if (NP == 3) {          // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(2), X(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(0), X(2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(1), X(1), c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(1), X(0), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(0), X(1), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W(0), X(0), c, 0, 0, 0);
    };
    // one K-step: the lane's fragment offset `off` (uint4 units, without the pixel tile), the step's weights `w`, next step's prefetched into `wn`
    uint4 wr[2][CN_CT][NP];
    auto kstep = [&](int off, int T, int cur) {
      load_w(T + 1, wr[cur ^ 1]);
      uint4 xr[2][NP];
#pragma unroll
      for (int m = 0; m < NP; ++m) xr[0][m] = Pl[m * PLANE + off];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j + 1 < 4) {
#pragma unroll
          for (int m = 0; m < NP; ++m) xr[(j + 1) & 1][m] = Pl[m * PLANE + off + (j + 1) * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CN_CT; ++i) mma(acc[i][j], wr[cur][i], xr[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    int s = 0, c0 = 0, T = 0;
    load_w(0, wr[0]);
    load_patch(0, 0);
    if (NP == 2) prep_patch();
    __syncthreads();                 // every wave is done with the previous tile's patch
    if (NP == 2) update_scale();
    store_patch();
    __syncthreads();
    while (true) {
      int s2 = s, c2 = c0 + C3_BK;
      if (c2 >= p.src[s].C) { c2 = 0; ++s2; }
      const bool more = s2 < p.nsrc;
      const bool quad = p.src[s].C - c0 <= 8;            // at most 8 valid channels in this chunk: four taps per K-step
      if (more) load_patch(s2, c2);
      if (!quad) {
#pragma unroll
        for (int st = 0; st < NPAIR; ++st) {
          const int tA = 2 * st, tB = 2 * st + 1 < TAPS ? 2 * st + 1 : TAPS - 1;     // (an odd tap count: the second half of the last step carries zero weights)
          const int oA = xbp[tA % KS] + (tA / KS) * PWP * 2, oB = xbp[tB % KS] + (tB / KS) * PWP * 2;
          kstep(g < 2 ? oA : oB, T + st, st & 1);
        }
        T += NPAIR;
        if (NPAIR & 1) {
#pragma unroll
          for (int i = 0; i < CN_CT; ++i)
#pragma unroll
            for (int m = 0; m < NP; ++m) wr[0][i][m] = wr[1][i][m];      // the set fetched during the last step is the next chunk's first
        }
      } else {
#pragma unroll
        for (int st = 0; st < NQUAD; ++st) {
          int o[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int tp = 4 * st + u < TAPS ? 4 * st + u : TAPS - 1; o[u] = xbq[tp % KS] + (tp / KS) * PWP * 2; }
          kstep(g == 0 ? o[0] : g == 1 ? o[1] : g == 2 ? o[2] : o[3], T + st, st & 1);
        }
        T += NQUAD;
        if (NQUAD & 1) {
#pragma unroll
          for (int i = 0; i < CN_CT; ++i)
#pragma unroll
            for (int m = 0; m < NP; ++m) wr[0][i][m] = wr[1][i][m];
        }
      }
      if (BLK) {
#pragma unroll
        for (int i = 0; i < CN_CT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc2[i][j] += acc[i][j]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
      if (NP == 2 && more) prep_patch();
      __syncthreads();
      if (!more) break;
      s = s2; c0 = c2;
      if (NP == 2) update_scale();
      store_patch();
      __syncthreads();
    }
    if (BLK) {
#pragma unroll
      for (int i = 0; i < CN_CT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc2[i][j];
    }

    const float inv_run = NP == 2 ? __uint_as_float((unsigned)(254 - kf) << 23) : 1.f;
    // ---- epilogue: lane holds pixel (64 wh + 16 j + l16) of tile row jr, channels 16 i + 4 g + {0..3} ----
    const bool want_red = p.slab != nullptr;
    float s1[CN_CT][4], s2v[CN_CT][4];
    float4 eav[CN_CT], ebv[CN_CT];
#pragma unroll
    for (int i = 0; i < CN_CT; ++i) {
      const int c = 16 * i + 4 * g;
      const int nrem = p.Cn - c;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[i][e] = 0.f; s2v[i][e] = 0.f; }
      eav[i] = MODE == MODE_FWD ? zero4() : make_float4(1.f, 1.f, 1.f, 1.f); ebv[i] = zero4();
      if (nrem <= 0) continue;
      if (MODE == MODE_FWD) {
        if (p.bias) eav[i] = ld4g(p.bias + c, nrem, false);
        if (p.bias_n) { const float4 b = ld4g(p.bias_n + (long)n * p.Cn + c, nrem, false); eav[i].x += b.x; eav[i].y += b.y; eav[i].z += b.z; eav[i].w += b.w; }
      } else if (p.dst.a) { eav[i] = ld4g(p.dst.a + c, nrem, p.vecY); ebv[i] = ld4g(p.dst.b + c, nrem, p.vecY); }
    }
    const int orow = oh + jr * d;
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += 2) {
      float4 xq[2][CN_CT], oq[2][CN_CT];
      if (MODE == MODE_DGRAD) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int lp = 64 * wh + 16 * (j0 + jj) + l16;
          const long pp = ((long)n * p.H + orow) * p.W + ow0 + lp;
          const bool pin = ow0 + lp < p.W && orow < p.H;
#pragma unroll
          for (int i = 0; i < CN_CT; ++i) {
            const int c = 16 * i + 4 * g;
            const int nrem = p.Cn - c;
            xq[jj][i] = zero4(); oq[jj][i] = zero4();
            if (!pin || nrem <= 0) continue;
            xq[jj][i] = ld4g(p.dst.x + pp * p.dst.ld + c, nrem, p.vecY);
            if (p.accumulate) oq[jj][i] = ld4g(p.y + pp * p.ldy + c, nrem, p.vecY);
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = j0 + jj;
        const int lp = 64 * wh + 16 * j + l16;
        const long pp = ((long)n * p.H + orow) * p.W + ow0 + lp;
        const bool pin = ow0 + lp < p.W && orow < p.H;
#pragma unroll
        for (int i = 0; i < CN_CT; ++i) {
          const int c = 16 * i + 4 * g;
          const int nrem = p.Cn - c;
          if (!pin || nrem <= 0) continue;
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          const bool odd = ((par0 + (unsigned)(jr * d + lp)) & 1u) != 0;                                             // undo the checkerboard sign of this pixel
          if (NP == 2) { const float f = odd ? -inv_run : inv_run; v.x = v.x * f * winv; v.y = v.y * f * winv; v.z = v.z * f * winv; v.w = v.w * f * winv; }      // ... and the two operand scales
          else if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
          if (MODE == MODE_FWD) {
            v.x += eav[i].x; v.y += eav[i].y; v.z += eav[i].z; v.w += eav[i].w;
            st4g(p.y + pp * p.ldy + c, v, nrem, p.vecY);
            if (want_red) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float f = (e < nrem) ? get4(v, e) : 0.f; s1[i][e] += f; s2v[i][e] += f * f; }
            }
          } else {
            const float4 x = xq[jj][i];
            float4 gq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xe = get4(x, e), ae = get4(eav[i], e), be = get4(ebv[i], e), dz = get4(v, e);
              const bool mk = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
              set4(gq, e, mk ? dz * ae : 0.f);
              if (want_red && mk) { s1[i][e] += dz * xe; s2v[i][e] += dz; }
            }
            if (p.accumulate) { const float4 o = oq[jj][i]; gq.x += o.x; gq.y += o.y; gq.z += o.z; gq.w += o.w; }
            st4g(p.y + pp * p.ldy + c, gq, nrem, p.vecY);
          }
        }
      }
    }
    if (want_red) {          // the 16 pixel lanes of a DPP row hold the same 12 channels (16 i + 4 g + e): reduce-scatter, lane l16 < 12 keeps value l16
      float va[16], vb[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) { va[v] = v < 12 ? s1[v >> 2][v & 3] : 0.f; vb[v] = v < 12 ? s2v[v >> 2][v & 3] : 0.f; }
      rs16(va, l16); rs16(vb, l16);
      tot_a += (double)va[0]; tot_b += (double)vb[0];
    }
  }
  if (p.slab) {
    __syncthreads();
    if (l16 < 12) {
      double* r = red + (wave * 48 + 16 * (l16 >> 2) + 4 * g + (l16 & 3)) * 2;
      r[0] = tot_a; r[1] = tot_b;
    }
    __syncthreads();
    if (t < 48 && t < p.Cn) {
      double* o = p.slab + ((long)bx * p.slab_ld + t) * 2;
      o[0] = red[2 * t] + red[2 * (48 + t)] + red[2 * (96 + t)] + red[2 * (144 + t)];
      o[1] = red[2 * t + 1] + red[2 * (48 + t) + 1] + red[2 * (96 + t) + 1] + red[2 * (144 + t) + 1];
      for (int r = bx + gx; r < p.slab_rows; r += gx) { double* z = p.slab + ((long)r * p.slab_ld + t) * 2; z[0] = 0.0; z[1] = 0.0; }      // rows no workgroup owns
    }
  }
}

}  // namespace

// 1 = launched, 0 = no instantiation; grid.x workgroups walk p.ntiles tiles of 2 rows x 128 pixels
int c3n_run(const void* kp, int ks, int mode, int np, dim3 grid, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
#define C3N_GO(K_, M_, P_) { \
    static bool attr = false; \
    auto fn = &conv3n_kernel<K_, M_, P_>; \
    const size_t lds = 4 * 48 * 16 + (size_t)P_ * (K_ + 1) * cn_pwp(K_) * 32; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, k); return 1; }
#define C3N_K(K_) if (ks == K_) { \
    if (mode == MODE_FWD) { if (np == 3) C3N_GO(K_, MODE_FWD, 3) else C3N_GO(K_, MODE_FWD, 2) } \
    else { if (np == 3) C3N_GO(K_, MODE_DGRAD, 3) else C3N_GO(K_, MODE_DGRAD, 2) } }
  C3N_K(3) C3N_K(5)
#undef C3N_K
#undef C3N_GO
  return 0;
}
