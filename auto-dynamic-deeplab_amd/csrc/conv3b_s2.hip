// conv3b_s2.hip — the stride-2 instantiations of conv3b_kernel (conv3b.h; launch logic: conv3.hip): stem2's 3x3 stride-2 forward on de-interleaved patch rows and
// its data gradient: four input-pixel parity classes computed from ONE staged image of the two dy rows they share (conv3s_kernel).
#include "conv3b.h"

#define C3B_GO(KERNEL, THREADS) { \
    static bool attr = false; \
    auto fn = &KERNEL; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL(fn, grid, dim3(THREADS), lds, st, k); return 1; }
#ifdef ADDK_C3B_DIAG
#define C3B_DIAG_READER(NAME) void NAME(unsigned long long* acc12) { \
    unsigned long long h[64][12]; \
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3b_diag), sizeof h) != hipSuccess) { (void)hipGetLastError(); return; } \
    for (int q = 0; q < 12; ++q) for (int i = 0; i < 64; ++i) acc12[q] += h[i][q]; \
    memset(h, 0, sizeof h); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c3b_diag), h, sizeof h); }
#else
#define C3B_DIAG_READER(NAME)
#endif

int c3b_run_s2f(const void* kp, int wc, int bpx, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
  if (wc != 4) return 0;
  if (bpx == 64) { if (np == 3) C3B_GO((conv3b_kernel<4, 3, MODE_FWD, 3, false, 1, 64, 2>), 256) else C3B_GO((conv3b_kernel<4, 3, MODE_FWD, 2, false, 1, 64, 2>), 256) }
  return 0;
}
// ---- [r5] the stride-2 data gradient with ONE staging for its four parity classes ------------------------------------------------------------------
// Input pixel (2a + pi, 2b + pj) of the 3x3 / stride 2 / pad 1 convolution collects dy(a + th, b + tw) W[kh][kw] over th <= pi, tw <= pj (kh = 1 for pi = 0,
// else {2, 0}[th]; the same for columns): 1, 2, 2, 4 taps for the classes (pi, pj) — all over the SAME two dy rows a, a + 1 and columns b .. b + 1.
// conv3b_s2d_kernel ran the classes as four stride-1 convolutions side by side, each staging its own copy of those rows (loads, prologue, split, LDS stores:
// 12.7 vector instructions per matrix instruction, 0.18 of the 6-term peak: VERDICT r04).  Here a workgroup stages rows a, a + 1 of 64 + 1 dy columns ONCE per
// 16-channel chunk and runs all nine (class, tap) products from that image into four accumulator sets: wave (ct, bh) owns gradient channels [32 ct, +32) of
// the 32 positions b0 + 32 bh .. of all four classes.  The packed weights are the four class streams of c3b_pack_s2d_body (conv3.hip), unchanged.
template <int NP>
__global__ void __launch_bounds__(256, 2) conv3s_kernel(const C3K p) {
  constexpr int BPX = 64, PWP = BPX + 16, PLANE = 2 * PWP * 2, NS = (2 * PWP * 4 + 255) / 256, BC = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red = reinterpret_cast<double*>(smem);                  // [2 position halves][64][2]
  unsigned* wmax = reinterpret_cast<unsigned*>(smem);             // NP = 2 (split-fp16, conv3b.h): the waves' largest staged magnitudes (aliases red[], written once at the very end)
  uint4* Pl = reinterpret_cast<uint4*>(smem + 2 * BC * 16);
  uint2* Pl2 = reinterpret_cast<uint2*>(Pl);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ct = wv & 1, bh = wv >> 1, lp32 = lane & 31, hh = lane >> 5;
  const int bx = blockIdx.x, gx = gridDim.x;
  const int q = t & 3;
  double tot_a = 0.0, tot_b = 0.0;
  // class c = 2 pi + pj has (1 + pi)(1 + pj) taps; the nine (class, tap) pairs in class order, tap = th * (1 + pj) + tw
  constexpr int PC[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
  constexpr int PTH[9] = {0, 0, 0, 0, 1, 0, 0, 1, 1};
  constexpr int PTW[9] = {0, 0, 1, 0, 0, 0, 1, 0, 1};
  constexpr int PTAP[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};
  constexpr int PRE[4] = {0, 1, 3, 5}, TC[4] = {1, 2, 2, 4};
  auto slot_geo = [&](int k, int& r, int& sp) { const int pix = (t + 256 * k) >> 2; r = pix / PWP; sp = pix - r * PWP; };
  unsigned pmask = 0;
#pragma unroll
  for (int k = 0; k < NS; ++k) { int r, sp; slot_geo(k, r, sp); pmask |= (unsigned)((r + sp) & 1) << k; }
  int xb[2];
#pragma unroll
  for (int tw = 0; tw < 2; ++tw) { const int pj = 32 * bh + lp32 + tw; xb[tw] = pj * 2 + (hh ^ ((pj >> 3) & 1)); }
  const uint4* wpl = reinterpret_cast<const uint4*>(p.wp) + (long)ct * NP * 64 + lane;
  const long cls_stride = p.wp_blk;                                // uint4 units per "tap unit" of the class streams (pack: unit * planes)
  const int nch = (p.src[0].C + C3_BK - 1) / C3_BK;
  const float winv = NP == 2 ? p.wsc[0] : 1.f;

  for (int tlin = bx; tlin < p.ntiles; tlin += gx) {
    const int rowid = tlin / p.spr, sx = tlin - rowid * p.spr;
    const int n = rowid / p.HT, a = rowid - n * p.HT;
    const int b0 = sx * BPX;
    unsigned vmask = 0;
    const int pbase = (n * p.IH + a) * p.IW + b0;
    const unsigned par0 = (unsigned)(a + b0);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int r, sp; slot_geo(k, r, sp);
      const bool ok = r < 2 && sp <= BPX && a + r < p.IH && b0 + sp < p.IW;
      vmask |= (ok ? 1u : 0u) << k;
    }
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float4 ra[NS];
    bool pch = false;
    int kf = 0;                                   // NP = 2: exponent field of the tile's running operand scale (conv3b.h)
    auto load_patch = [&](int c0_) {
      const addk_src S = p.src[0];                                  // dy: no lazy BatchNorm, no ReLU (the BatchNorm backward has been applied in place)
      const int c = c0_ + 4 * q;
      pch = c < S.C;
      const float* sb = S.x + (pch ? c : 0);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        int r, sp; slot_geo(k, r, sp);
        const int po = ((vmask >> k) & 1u) ? pbase + r * p.IW + sp : 0;
        ra[k] = ld4(sb + (long)po * S.ld);
      }
    };
    auto signed_slot = [&](int k) {
      float4 v = ra[k];
      const bool ok = pch && ((vmask >> k) & 1u);
      const float sg = ok ? ((((pmask >> k) ^ par0) & 1u) ? -1.f : 1.f) : 0.f;
      v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
      return v;
    };
    auto prep_patch = [&]() {                     // NP = 2: sign / zero padding before the barrier, the wave's largest magnitude to wmax[] (conv3b.h)
      unsigned mx = 0;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const float4 v = signed_slot(k);
        ra[k] = v;
        const unsigned b = absbits4(v);
        mx = b > mx ? b : mx;
      }
      mx = wave_umax(mx);
      if (lane == 0) wmax[wv] = mx;
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
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[c][e] *= r;
      }
      if (kf == 0 || want < kf) kf = want;
    };
    auto store_patch = [&]() {
      const float sc = NP == 2 ? __uint_as_float((unsigned)kf << 23) : 1.f;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 v;
        if (NP == 2) { v = ra[k]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        else v = signed_slot(k);
        int r, sp; slot_geo(k, r, sp);
        if (r < 2) {
          uint2 pl[NP];
          split4<NP>(v, pl);
          const int slot = (r * PWP + sp) * 2 + ((q >> 1) ^ ((sp >> 3) & 1));
#pragma unroll
          for (int m = 0; m < NP; ++m) Pl2[(m * PLANE + slot) * 2 + (q & 1)] = pl[m];
        }
      }
    };
    auto load_w = [&](int chunk, int k9, uint4* dst) {            // weight fragments of pair k9 of `chunk` (clamped: the prefetch runs past the last chunk)
      const int ch = chunk < nch ? chunk : nch - 1;
      const int c = PC[k9];
      const uint4* src = wpl + (long)PRE[c] * cls_stride + (long)((ch * TC[c] + PTAP[k9]) * 2) * NP * 64;
#pragma unroll
      for (int m = 0; m < NP; ++m) dst[m] = src[m * 64];
    };
    auto mma = [&](f32x16& c, const uint4* w, const uint4* x) {
      if constexpr (NP == 2) {          // split-fp16: l*wh + h*wl + h*wh
        auto Wh = [&](int m) { return __builtin_bit_cast(f16x8, w[m]); };
        auto Xh = [&](int m) { return __builtin_bit_cast(f16x8, x[m]); };
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(1), Xh(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(0), Xh(1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(0), Xh(0), c, 0, 0, 0);
        return;
      }
      auto W = [&](int m) { return __builtin_bit_cast(bf16x8, w[m]); };
      auto X = [&](int m) { return __builtin_bit_cast(bf16x8, x[m]); };
      if (NP == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(2), X(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(1), c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(0), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(1), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(0), c, 0, 0, 0);
    };
    uint4 wr[3][NP];
    load_w(0, 0, wr[0]);
    load_w(0, 1, wr[1]);
    load_patch(0);
    if (NP == 2) prep_patch();
    __syncthreads();
    if (NP == 2) update_scale();
    store_patch();
    __syncthreads();
    for (int chunk = 0; chunk < nch; ++chunk) {
      const bool more = chunk + 1 < nch;
      if (more) load_patch((chunk + 1) * C3_BK);
      // the four dy fragments of the chunk: (th, tw) = patch row th, column shift tw
      uint4 xf[2][2][NP];
#pragma unroll
      for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int tw = 0; tw < 2; ++tw)
#pragma unroll
          for (int m = 0; m < NP; ++m) xf[th][tw][m] = Pl[m * PLANE + xb[tw] + th * PWP * 2];
#pragma unroll
      for (int k9 = 0; k9 < 9; ++k9) {
        const int nk = k9 + 2;                                     // two pairs ahead, into the next chunk at the end
        load_w(nk < 9 ? chunk : chunk + 1, nk < 9 ? nk : nk - 9, wr[nk % 3]);
        __builtin_amdgcn_sched_barrier(0);
        mma(acc[PC[k9]], wr[k9 % 3], xf[PTH[k9]][PTW[k9]]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (9 pairs: the two sets fetched ahead sit in ring slots 0 and 1 again — no rotation)
      if (NP == 2 && more) prep_patch();
      __syncthreads();
      if (!more) break;
      if (NP == 2) update_scale();
      store_patch();
      __syncthreads();
    }

    // ---- epilogue: class (pi, pj) of position (a, b0 + 32 bh + lp32) is gradient pixel (2a + pi, 2b + pj); channels 32 ct + 8 g + 4 hh + {0..3} ----
    const bool want_red = p.slab != nullptr;
    float s1[4][4], s2v[4][4];
    float4 eav[4], ebv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = ct * 32 + 8 * g + 4 * hh;
      const int nrem = p.Cn - c;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[g][e] = 0.f; s2v[g][e] = 0.f; }
      eav[g] = make_float4(1.f, 1.f, 1.f, 1.f); ebv[g] = zero4();
      if (nrem > 0 && p.dst.a) { eav[g] = ld4g(p.dst.a + c, nrem, p.vecY); ebv[g] = ld4g(p.dst.b + c, nrem, p.vecY); }
    }
    const int b = b0 + 32 * bh + lp32;
    const bool flip = ((par0 + (unsigned)lp32) & 1u) != 0;          // the checkerboard sign of position (a, b) (32 bh is even)
    const float inv_run = NP == 2 ? __uint_as_float((unsigned)(254 - kf) << 23) : 1.f, fsc = flip ? -inv_run : inv_run;
#pragma unroll
    for (int c0 = 0; c0 < 4; c0 += 2) {                             // two classes at a time: request everything they read, then compute and store
      float4 xq[2][4], oq[2][4];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = c0 + cc, oh = 2 * a + (cl >> 1), ow = 2 * b + (cl & 1);
        const bool pin = oh < p.H && ow < p.W;
        const long pp = ((long)n * p.H + oh) * p.W + ow;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = ct * 32 + 8 * g + 4 * hh;
          const int nrem = p.Cn - c;
          xq[cc][g] = zero4(); oq[cc][g] = zero4();
          if (!pin || nrem <= 0) continue;
          xq[cc][g] = ld4g(p.dst.x + pp * p.dst.ld + c, nrem, p.vecY);
          if (p.accumulate) oq[cc][g] = ld4g(p.y + pp * p.ldy + c, nrem, p.vecY);
        }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = c0 + cc, oh = 2 * a + (cl >> 1), ow = 2 * b + (cl & 1);
        const bool pin = oh < p.H && ow < p.W;
        const long pp = ((long)n * p.H + oh) * p.W + ow;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = ct * 32 + 8 * g + 4 * hh;
          const int nrem = p.Cn - c;
          if (!pin || nrem <= 0) continue;
          float4 v = make_float4(acc[cl][4 * g], acc[cl][4 * g + 1], acc[cl][4 * g + 2], acc[cl][4 * g + 3]);
          if (NP == 2) { v.x = v.x * fsc * winv; v.y = v.y * fsc * winv; v.z = v.z * fsc * winv; v.w = v.w * fsc * winv; }      // sign and the two operand scales
          else if (flip) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
          const float4 x = xq[cc][g];
          float4 gq;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = get4(x, e), ae = get4(eav[g], e), be = get4(ebv[g], e), dz = get4(v, e);
            const bool mk = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
            set4(gq, e, mk ? dz * ae : 0.f);
            if (want_red && mk) { s1[g][e] += dz * xe; s2v[g][e] += dz; }
          }
          if (p.accumulate) { const float4 o = oq[cc][g]; gq.x += o.x; gq.y += o.y; gq.z += o.z; gq.w += o.w; }
          st4g(p.y + pp * p.ldy + c, gq, nrem, p.vecY);
        }
      }
    }
    if (want_red) {
      float va[16], vb[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[4 * g + e] = s1[g][e]; vb[4 * g + e] = s2v[g][e]; }
      rs32(va, lp32); rs32(vb, lp32);
      tot_a += (double)va[0]; tot_b += (double)vb[0];
    }
  }
  if (p.slab) {
    __syncthreads();
    if (!(lp32 & 1)) {
      const int v = lp32 >> 1;
      double* r = red + ((bh * BC) + ct * 32 + 8 * (v >> 2) + 4 * hh + (v & 3)) * 2;
      r[0] = tot_a; r[1] = tot_b;
    }
    __syncthreads();
    if (t < BC && t < p.Cn) {
      double* o = p.slab + ((long)bx * p.slab_ld + t) * 2;
      o[0] = red[2 * t] + red[2 * (BC + t)]; o[1] = red[2 * t + 1] + red[2 * (BC + t) + 1];
      for (int r = bx + gx; r < p.slab_rows; r += gx) { double* z = p.slab + ((long)r * p.slab_ld + t) * 2; z[0] = 0.0; z[1] = 0.0; }
    }
  }
}

int c3b_run_s2d(const void* kp, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
  (void)lds;
#define C3S_GO(P_) { \
    static bool attr = false; \
    auto fn = &conv3s_kernel<P_>; \
    const size_t sz = 2 * 64 * 16 + (size_t)P_ * 2 * (64 + 16) * 32; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL(fn, grid, dim3(256), sz, st, k); return 1; }
  if (np == 3) C3S_GO(3) else C3S_GO(2)
#undef C3S_GO
  return 0;
}
C3B_DIAG_READER(c3b_diag_s2)
