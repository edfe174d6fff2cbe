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
#include "conv3b.h"

namespace {

constexpr int C3_RS = 20;                  // floats per patch pixel in LDS (16 + 4: ds_read_b64 fragment reads hit 64 distinct banks per half-wave)
constexpr int C3_MAXCH = 64;               // chunks per launch (K <= 1024 channels)
// widest patch row: 3x3 up to dilation 18 (ASPP), 5x5 up to dilation 2 (the cells' dil_conv_5x5)
constexpr int c3_pwmax(int ks) { return ks == 3 ? C3_BP + 2 * 18 : C3_BP + 4 * 2; }
constexpr int c3_maxdil(int ks) { return ks == 3 ? 18 : 2; }


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
  int s2d;                  // conv3b: 1 = the four parity-class packs of the stride-2 data gradient (c3b_pack_s2d_body); 2 = the 16-wide-tile kernel's pack (c3n_pack_body)
  int sbase[C3_MAXCH];      // s2d == 2: first K-step of the chunk (conv3n.hip: two taps x 16 channels per step, four taps x 8 for a chunk of <= 8 channels)
  int wrows;                // rows of the whole weight tensor [wrows][ldw] behind `w` (the amax pass scans it contiguously)
  float* amax;              // planes == 2 (split-fp16, conv3b.h): the 128-float trailer of the pack buffer — [0..63] the amax pass's per-workgroup maxima of |w|,
                            // [64] = 2^-kw for the kernel's epilogue (C3K.wsc), [65] = 2^kw
};
constexpr int C3_AMAX_WG = 64;             // workgroups of the amax pass = partial maxima in the trailer
constexpr int C3_TRAILER = 128;            // floats

// out[colblk][T = chunk*taps + tap][tile i][lane][m]  =  W(row = colblk*BC + i*16 + li, tap, k = 8*(m>>1) + 2*kq + (m&1))
__device__ __forceinline__ void c3_pack_body(const PackK& p, long first, long stride);
__global__ void __launch_bounds__(256) c3_pack_kernel(const PackK p) { c3_pack_body(p, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256); }
// all packs of a plan in one launch at the start of the step (the weights only change in the optimizer): block (x, y) works on descriptor y
__global__ void __launch_bounds__(256) c3_pack_batch_kernel(const PackK* __restrict__ descs) {
  c3_pack_body(descs[blockIdx.y], (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}
__device__ __forceinline__ void c3b_pack_body(const PackK& p, long first, long stride);
// split-fp16 packs (planes == 2): the pass in front of the pack scans the convolution's WHOLE weight tensor (contiguous, coalesced: forward and data-gradient
// packs of one convolution, and the data gradients towards its several sources, get the same scale) and leaves the largest |w| each of its C3_AMAX_WG
// workgroups saw in the trailer (max is order-independent: bit-reproducible); every pack thread folds the 64 partials into the tensor's scale 2^kw itself
__device__ __forceinline__ void c3_amax_body(const PackK& p) {
  if (p.planes != 2) return;
  float m = 0.f;
  const long n = (long)p.wrows * p.ldw;
  const gfloat* w = (const gfloat*)p.w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  __shared__ unsigned wm[4];
  const unsigned b = wave_umax(__float_as_uint(m));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned r = wm[0];
    for (int i = 1; i < 4; ++i) r = wm[i] > r ? wm[i] : r;
    p.amax[blockIdx.x] = __uint_as_float(r);
  }
}
__global__ void __launch_bounds__(256) c3_pack_amax_kernel(const PackK p) { c3_amax_body(p); }
__global__ void __launch_bounds__(256) c3_pack_amax_batch_kernel(const PackK* __restrict__ descs) { c3_amax_body(descs[blockIdx.y]); }
// scale of a split-fp16 pack from the amax pass's partials; the first thread of the pack leaves its inverse for the convolution's epilogue
__device__ __forceinline__ float c3_pack_scale(const PackK& p, long first) {
  if (p.planes != 2) return 1.f;
  float m = 0.f;
  const float4* q = reinterpret_cast<const float4*>(p.amax);
#pragma unroll
  for (int i = 0; i < C3_AMAX_WG / 4; ++i) { const float4 v = q[i]; m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)); }
  const int kf = f16_scale_field(__float_as_uint(m));
  if (first == 0) { p.amax[C3_AMAX_WG] = __uint_as_float((unsigned)(254 - kf) << 23); p.amax[C3_AMAX_WG + 1] = __uint_as_float((unsigned)kf << 23); }
  return __uint_as_float((unsigned)kf << 23);
}
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


// packed weights: out (16-byte units) [colblk][T = chunk*taps + tap][tile (wave) i][plane][lane] = 8 bf16:
//   W(row = colblk*32*bct + i*32 + (lane & 31), tap, k = 8*(lane >> 5) + j), j = 0..7       (bct = 32-row tiles per block)
// Stride-2 data gradient (3x3, pad 1): input pixel (2a + pi, 2b + pj) collects dy(a + th, b + tw) W[kh][kw] over th <= pi, tw <= pj with
// kh = 1 for pi = 0, else {2, 0}[th] (same for columns) — 1, 2, 2, 4 of the nine taps per parity class.  Four packs back to back, class
// c = 2 pi + pj at (taps of the classes before) * unit, each [colblk][chunk * tc + tap][tile][plane][lane]; tap (th, tw) carries
// (-1)^(th + tw), its share of the checkerboard sign over the class grid (a, b).
__device__ __forceinline__ void c3b_pack_s2d_body(const PackK& p, long first, long stride) {
  const int BC = 32 * p.bct, NP = p.planes;
  const float wsc = c3_pack_scale(p, first);
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
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      const int kk = k0 + j;
      if (row < p.Cn && kk < p.cvalid[chunk]) v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(kh * 3 + kw) * p.cin_total + p.w_choff + row];
      if ((th + tw) & 1) v = -v;
      vv[j] = v * wsc;
#pragma unroll
      for (int k = 0; k < 3; ++k) { b[k][j] = bf16_hi(v); v = v - bf16_f(b[k][j]); }
    }
    uint4* o = out + pre * unit * NP + (((long)blk * nT + T) * p.bct + i) * NP * 64 + lane;
    if (NP == 2) { split8h(vv, o[0], o[64]); continue; }
    for (int k = 0; k < NP; ++k)
      o[k * 64] = make_uint4(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16), b[k][4] | (b[k][5] << 16), b[k][6] | (b[k][7] << 16));
  }
}

// conv3n_kernel (conv3n.hip): out (16-byte units) [K-step T][16-row tile i (3)][plane][lane] = 8 bf16: row co = 16 i + lane % 16; lane group g = lane / 16 holds, in a
// pair step, channels 8 (g & 1) .. + 7 of tap 2 st + (g >> 1), in a quad step (chunk of <= 8 channels) channels 0..7 of tap 4 st + g; taps beyond the kernel: zeros
__device__ __forceinline__ void c3n_pack_body(const PackK& p, long first, long stride) {
  const int NP = p.planes;
  const float wsc = c3_pack_scale(p, first);
  const int ks = p.taps == 9 ? 3 : 5;
  const int last = p.nchunks - 1;
  const int steps = p.sbase[last] + (p.cvalid[last] <= 8 ? (p.taps + 3) / 4 : (p.taps + 1) / 2);
  uint4* out = reinterpret_cast<uint4*>(p.out);
  for (long idx = first; idx < (long)steps * 3 * 64; idx += stride) {
    const int lane = (int)(idx & 63);
    const int r = (int)(idx >> 6);
    const int i = r % 3, T = r / 3;
    int chunk = 0;
    while (chunk < last && p.sbase[chunk + 1] <= T) ++chunk;
    const int st = T - p.sbase[chunk];
    const bool quad = p.cvalid[chunk] <= 8;
    const int g = lane >> 4, row = 16 * i + (lane & 15);
    const int tap = quad ? 4 * st + g : 2 * st + (g >> 1), kc0 = quad ? 0 : 8 * (g & 1);
    const bool flip = p.dil_odd == 1 && tap < p.taps && (((tap / ks) + (tap % ks)) & 1) != 0;
    unsigned b[3][8];
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      const int kk = kc0 + j;
      if (row < p.Cn && tap < p.taps && kk < p.cvalid[chunk]) {
        if (p.mode == MODE_FWD) v = p.w[(long)row * p.ldw + (long)tap * p.cin_total + p.cbase[chunk] + kk];
        else                    v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(p.taps - 1 - tap) * p.cin_total + p.w_choff + row];
      }
      if (flip) v = -v;
      vv[j] = v * wsc;
#pragma unroll
      for (int k = 0; k < 3; ++k) { b[k][j] = bf16_hi(v); v = v - bf16_f(b[k][j]); }
    }
    uint4* o = out + ((long)(T * 3 + i) * NP) * 64 + lane;
    if (NP == 2) { split8h(vv, o[0], o[64]); continue; }
    for (int k = 0; k < NP; ++k)
      o[k * 64] = make_uint4(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16), b[k][4] | (b[k][5] << 16), b[k][6] | (b[k][7] << 16));
  }
}

__device__ __forceinline__ void c3b_pack_body(const PackK& p, long first, long stride) {
  if (p.s2d == 2) { c3n_pack_body(p, first, stride); return; }
  if (p.s2d) { c3b_pack_s2d_body(p, first, stride); return; }
  const int BC = 32 * p.bct, NP = p.planes;
  const float wsc = c3_pack_scale(p, first);
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
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      const int kk = k0 + j;
      if (row < p.Cn && kk < p.cvalid[chunk]) {
        if (p.mode == MODE_FWD) v = p.w[(long)row * p.ldw + (long)tap * p.cin_total + p.cbase[chunk] + kk];
        else                    v = p.w[(long)(p.cbase[chunk] + kk) * p.ldw + (long)(p.taps - 1 - tap) * p.cin_total + p.w_choff + row];
      }
      if (flip) v = -v;
      vv[j] = v * wsc;
#pragma unroll
      for (int k = 0; k < 3; ++k) { b[k][j] = bf16_hi(v); v = v - bf16_f(b[k][j]); }
    }
    uint4* o = out + (((long)blk * nT + T) * p.bct + i) * NP * 64 + lane;
    if (NP == 2) { split8h(vv, o[0], o[64]); continue; }
    for (int k = 0; k < NP; ++k)
      o[k * 64] = make_uint4(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16), b[k][4] | (b[k][5] << 16), b[k][6] | (b[k][7] << 16));
  }
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
// [r5] the 16-wide-tile kernel (conv3n.hip) takes the stride-1 3x3 / 5x5 launches at dilation <= 2 with at most 48 output channels (the cells' dilated convs at 40
// channels, forward and data gradient).  ADDK_C3N=0: those launches stay on conv3b_kernel's 64-channel blocks.
inline bool c3n_shape(int Cn, int taps, int dil, int stride) {
  // (the 3x3 stays on conv3b_kernel: 24.8 against 25.4-26.8 us at 2x125x253 — one workgroup per CU leaves its 13 K-steps nothing to hide the staging behind;
  //  ADDK_C3N=2 routes it here too)
  const int on = addk_env("ADDK_C3N", 1);
  return on != 0 && Cn <= 48 && (taps == 25 || (taps == 9 && on == 2)) && dil <= 2 && stride == 1 && c3_planes(Cn, taps) != 0;
}
inline int c3n_steps(int taps, int valid) { return valid <= 8 ? (taps + 3) / 4 : (taps + 1) / 2; }
long c3_pack_floats(int Cn, int nchunks, long P, int taps, int dil = 0, int stride = 1) {
  if (dil > 0 && c3n_shape(Cn, taps, dil, stride))     // 16-wide tiles: 1 KB per (K-step, 16-row tile, plane); a chunk has at most (taps + 1) / 2 steps
    return (long)nchunks * ((taps + 1) / 2) * 3 * c3_planes(Cn, taps) * 256 + (c3_planes(Cn, taps) == 2 ? C3_TRAILER : 0);
  if (const int np = c3_planes(Cn, taps)) {            // bf16 / fp16 planes: 1 KB per (tap, 32-row tile, plane); split-fp16: + the scale trailer (PackK.amax)
    const int wc = c3b_wc(Cn, P);
    return (long)cdiv(Cn, 32 * wc) * nchunks * taps * wc * np * 256 + (np == 2 ? C3_TRAILER : 0);
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
  if (dil <= 2 && H < 2 * dil) return false;      // two-row tiles (the one-row forms of these shapes are not instantiated)
  // the split kernel has half-width (64-pixel) tiles and also takes the 32x64 maps of level 3 (dil_conv at 160 channels: 152 us on
  // the generic kernel); the fp32 halo kernel keeps its 128-pixel tiles and the larger maps
  if (c3_planes(Cn, KH * KW)) return W >= 48 && P >= 2048;
  return W >= 100 && P >= 8192;
}

// the pack of one launch on its own stream position (the plans hoist all packs into one addk_conv_pack_batch): split-fp16 packs run the amax pass first
void c3_pack_now(const PackK& pk, int pb, hipStream_t st) {
  if (pk.planes == 2) hipLaunchKernelGGL(c3_pack_amax_kernel, dim3(C3_AMAX_WG), dim3(256), 0, st, pk);
  hipLaunchKernelGGL(c3_pack_kernel, dim3(pb), dim3(256), 0, st, pk);
}

// stride-2 data gradient: one launch per parity class of the input pixel (each a stride-1 gather over dy with 1 / 2 / 2 / 4 taps and a
// strided scatter of its outputs); the statistics-slab rows (= workgroups) are shared out in proportion to the tap counts
int c3b_s2_dgrad(C3K& k, PackK& pk, int rows, hipStream_t st, bool packed, PackK* desc_out, int np) {
  const int wc = 2;
  pk.bct = wc; pk.mode = MODE_DGRAD; pk.Cn = k.Cn; pk.planes = np; pk.dil_odd = 0; pk.s2d = 1;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }
  const long unit = (long)cdiv(k.Cn, 32 * wc) * pk.nchunks * wc * 64;
  if (!packed) { int pb = cdiv(9 * unit, 256); if (pb > 4096) pb = 4096; c3_pack_now(pk, pb, st); }
  // one launch: a tile is dy row a x 64 positions b (gradient rows 2a, 2a + 1 x 128 pixels), every workgroup runs all four parity classes from one staged image
  const int HA = (k.OHo + 1) / 2, WA = (k.OWo + 1) / 2;
  k.H = k.OHo; k.W = k.OWo;                 // the gradient map (k.IH, k.IW: the dy map)
  k.HT = HA; k.spr = cdiv(WA, 64);
  k.ntiles = k.N * HA * k.spr;
  k.wp = pk.out;
  k.wp_blk = unit * np;                       // 16-byte units per tap unit of the four class streams (c3b_pack_s2d_body)
  k.nT = 9 * pk.nchunks;
  k.red32 = 1;
  k.slab_rows = rows;
  dim3 grid(rows < k.ntiles ? rows : k.ntiles, 1);
  if (!c3b_run_s2d(&k, np, grid, 0, st)) { addk_set_error("conv3b stride-2 data gradient: no instantiation"); return ADDK_ERR_UNSUPPORTED; }
  return addk_check_launch("conv3b stride-2 data gradient");
}

// the <= 48-channel launches on 16-wide tiles (conv3n.hip)
int c3n_launch(C3K& k, PackK& pk, int mode, int rows, hipStream_t st, bool packed, PackK* desc_out, int np) {
  pk.bct = 3; pk.mode = mode; pk.Cn = k.Cn; pk.planes = np; pk.dil_odd = k.dil & 1; pk.s2d = 2;
  int steps = 0;
  for (int c = 0; c < pk.nchunks; ++c) { pk.sbase[c] = steps; steps += c3n_steps(pk.taps, pk.cvalid[c]); }
  const int ks = pk.taps == 9 ? 3 : 5;
  k.nT = steps;
  k.wp = pk.out;
  k.wp_blk = 0;
  k.HT = cdiv(k.H, 2 * k.dil) * k.dil;
  k.spr = cdiv(k.W, 128);
  k.ntiles = k.N * k.HT * k.spr;
  k.red32 = 1;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }
  const long total = (long)steps * 3 * 64;
  int pb = cdiv(total, 256); if (pb > 4096) pb = 4096;
  if (!packed) c3_pack_now(pk, pb, st);
  // one workgroup per tile at most: a workgroup without a tile would still queue for a CU's LDS and registers (989 slab rows against 252 tiles at 2x125x253)
  k.slab_rows = rows;
  if (!c3n_run(&k, ks, mode, np, dim3(rows < k.ntiles ? rows : k.ntiles, 1), st)) { addk_set_error("conv3n: no instantiation for %d taps", pk.taps); return ADDK_ERR_UNSUPPORTED; }
  return addk_check_launch("conv3n");
}

int c3b_launch(C3K& k, PackK& pk, int mode, int rows, hipStream_t st, bool packed, PackK* desc_out, int np) {
  pk.s2d = 0;
  if (k.st == 2 && mode == MODE_DGRAD) return c3b_s2_dgrad(k, pk, rows, st, packed, desc_out, np);
  if (k.om == 1 && c3n_shape(k.Cn, pk.taps, k.dil, k.st)) return c3n_launch(k, pk, mode, rows, st, packed, desc_out, np);
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
  // (every 3x3 / 5x5 launch at dilation <= 2 of full width, or of half width on 3-wave blocks, takes them; c3_geometry_ok requires H >= 2 dil)
  const bool tworow = k.st == 1 && (ks == 3 || ks == 5) && !bigd && k.om == 1 && (bpx == C3_BP || (bpx == 64 && wc == 3)) && k.H >= 2 * k.dil;
  const int rpx = tworow ? bpx / 2 : bpx;                       // pixels per tile row
  k.HT = tworow ? cdiv(k.H, 2 * k.dil) * k.dil : k.H;
  k.spr = cdiv(k.W, rpx);
  k.ntiles = k.N * k.HT * k.spr;
  k.red32 = 1;
  if (desc_out) { *desc_out = pk; return ADDK_OK; }
  const long total = (long)cdiv(k.Cn, 32 * wc) * k.nT * wc * 64;   // pack threads: one per (tile, lane)
  int pb = cdiv(total, 256); if (pb > 4096) pb = 4096;
  if (!packed) c3_pack_now(pk, pb, st);
  const size_t lds = (size_t)((ph * 32 * wc * 16 + 15) & ~15) + (size_t)np * (tworow ? ks + 1 : ks) * cb_pwmax(ks, bigd, rpx, k.st) * 32;
  k.slab_rows = rows;
  k.ny = cdiv(k.Cn, 32 * wc);
  dim3 grid((rows < k.ntiles ? rows : k.ntiles) * k.ny, 1);      // no workgroups without a tile (they would queue for LDS and registers just to write zeros); the channel blocks of a tile side by side (conv3b_kernel)
  // the instantiations live in conv3b_tr3 / conv3b_tr5 / conv3b_row / conv3b_s2.hip (compiled in parallel)
  const bool done = tworow ? (ks == 3 ? c3b_run_tr3(&k, wc, rpx, mode, np, grid, lds, st) : c3b_run_tr5(&k, wc, rpx, mode, np, grid, lds, st)) != 0
                  : k.st == 2 ? (ks == 3 && mode == MODE_FWD && c3b_run_s2f(&k, wc, bpx, np, grid, lds, st) != 0)
                  : c3b_run_row(&k, wc, ks, bigd, bpx, mode, np, grid, lds, st) != 0;
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
  return c3_pack_floats(a->Cout, nch, (long)a->N * a->OH * a->OW, a->KH * a->KW, a->dil, a->stride);
}
extern "C" int64_t addk_conv_dgrad_pack_floats(const addk_conv_dgrad_args* a) {
  if (!a) return 0;
  if (!c3_geometry_ok(a->KH, a->KW, a->stride, a->pad, a->dil, a->H, a->W, a->OH, a->OW, (long)a->N * a->H * a->W, a->dst.C)) return 0;
  if (a->Cout % 4 || a->lddy % 4 || a->ldg % 4 || a->dst.ld % 4 || a->dst.C % 4) return 0;
  const int nch = cdiv(a->Cout, C3_BK);
  if (nch > C3_MAXCH) return 0;
  return c3_pack_floats(a->dst.C, nch, (long)a->N * a->H * a->W, a->KH * a->KW, a->dil, a->stride);
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
  pk.amax = c3_planes(a->Cout, pk.taps) == 2 ? a->wpack + (need - C3_TRAILER) : nullptr; k.wsc = pk.amax ? pk.amax + C3_AMAX_WG : nullptr;
  pk.wrows = a->Cout;
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
  pk.amax = c3_planes(a->dst.C, pk.taps) == 2 ? a->wpack + (need - C3_TRAILER) : nullptr; k.wsc = pk.amax ? pk.amax + C3_AMAX_WG : nullptr;
  pk.wrows = a->Cout;
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
extern "C" int addk_c3b_diag(unsigned long long* out12) {
  for (int k = 0; k < 12; ++k) out12[k] = 0;
  c3b_diag_tr3(out12); c3b_diag_tr5(out12); c3b_diag_row(out12); c3b_diag_s2(out12);      // every instantiating unit has its own counters
  return ADDK_OK;
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
  const int m = addk_get_conv_precision();
  if (m == 1 || m == 3)        // modes with split-fp16 launches: their packs take the tensor's scale from the amax pass (descriptors of other packs return at once)
    hipLaunchKernelGGL(c3_pack_amax_batch_kernel, dim3(C3_AMAX_WG, n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackK*>(dev_descs));
  hipLaunchKernelGGL(c3_pack_batch_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackK*>(dev_descs));
  return addk_check_launch("conv_pack_batch");
}
