// Weight gradient of the dense convolution for one source:
//     dW[co][tap][c] = sum_p dy[p, co] * Z[p @ tap, c],   Z = relu?(a*x+b) recomputed on the fly.
// GEMM with M = output channels (tile 16*CTY), N = input channels of one tap (tile 16*CTZ) and the
// reduction over pixels.  A block owns one (co tile, tap, c tile) and one slice of the pixel range;
// its four waves each take 16 of the 64 pixels staged per step (both tiles are staged in their memory
// order [pixel][channel]; MFMA fragments are read with ds_read_b32, rows padded so that the two pixel
// rows a 32-lane group touches fall on disjoint banks).  Wave partials are combined through LDS in a
// fixed order and the per-slice tiles go to a workspace that addk reduces deterministically into dW.
#include <stdlib.h>
#include "common.h"

namespace {

struct WgK {
  const float* dy; int lddy; int Cout;
  int N, H, W, OH, OW, KH, KW, stride, pad, dil;
  addk_src src;
  float* ws;
  int taps, nzt, nyt;      // tiles: taps, z (input-channel) tiles, y (output-channel) tiles
  int splits; int P; int chunkP;
  int vecY, vecZ;
  // reduction target (used by the batched reduce)
  float* dw; int ldw, cin_total, w_choff, accumulate;
};

// Descriptor of this block's convolution, BY VALUE (scalar registers, loaded once): either the kernel argument or the
// batch table's entry with its pointers declared global (common.h, gptr).
template <bool BATCH>
__device__ __forceinline__ WgK wg_desc(const WgK& pv, const WgK* __restrict__ ops, int op) {
  if (!BATCH) return pv;
  WgK k = ops[op];
  k.dy = gptr(k.dy); k.src.x = gptr(k.src.x); k.src.a = gptr(k.src.a); k.src.b = gptr(k.src.b); k.ws = gptr(k.ws); k.dw = gptr(k.dw);
  return k;
}

constexpr int KP = 64;
constexpr int ldpad(int bc) { return (bc % 32 == 16) ? bc : bc + 16; }

// Batched form: `work[b] = (op, bx, by, -)` lets ONE launch cover the weight gradients of many convolutions
// (they are mutually independent and individually too small to fill 256 CUs); `ops == nullptr` is the plain launch.
template <int CTY, int CTZ, bool BATCH>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {      // wave-uniform: keep the descriptor in scalar registers like a kernel argument
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int BCY = 16 * CTY, BCZ = 16 * CTZ;
  constexpr int LY = ldpad(BCY), LZ = ldpad(BCZ);
  constexpr int NYJ = (KP * BCY / 4 + 255) / 256;
  constexpr int NZJ = (KP * BCZ / 4 + 255) / 256;
  constexpr int STAGE = KP * LY + KP * LZ;
  constexpr int TILE = BCY * BCZ;
  constexpr int LDSF = STAGE > TILE ? STAGE : TILE;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* Ys = lds;
  float* Zs = lds + KP * LY;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  int bx = blk_x;
  const int zt = bx % p.nzt; bx /= p.nzt;
  const int tap = bx % p.taps; const int yt = bx / p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = yt * BCY, c0 = zt * BCZ;
  const int ohw = p.OH * p.OW;
  const int pbeg = blk_y * p.chunkP;
  int pend = pbeg + p.chunkP; if (pend > p.P) pend = p.P;
  // lazy-BN scale/shift of this thread's channel quads (fixed across pixel steps)
  float4 za[NZJ], zb[NZJ];
#pragma unroll
  for (int j = 0; j < NZJ; ++j) {
    int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
    int c = c0 + 4 * q;
    za[j] = make_float4(1.f, 1.f, 1.f, 1.f); zb[j] = zero4();
    if (p.src.a && row < KP && c < p.src.C) { za[j] = ld4g(p.src.a + c, p.src.C - c, p.vecZ); zb[j] = ld4g(p.src.b + c, p.src.C - c, p.vecZ); }
  }
  const bool zrelu = p.src.relu != 0;

  f32x4 acc[CTY][CTZ];
#pragma unroll
  for (int i = 0; i < CTY; ++i)
#pragma unroll
    for (int j = 0; j < CTZ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ry[NYJ], rz[NZJ];
  unsigned zmask = 0;
  auto load_step = [&](int p0) {
    zmask = 0;
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCY / 4), q = slot - row * (BCY / 4);
      int pp = p0 + row; int co = co0 + 4 * q;
      float4 v = zero4();
      if (row < KP && pp < pend && co < p.Cout) v = ld4g(p.dy + (long)pp * p.lddy + co, p.Cout - co, p.vecY);
      ry[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NZJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
      int pp = p0 + row; int c = c0 + 4 * q;
      float4 v = zero4();
      if (row < KP && pp < pend && c < p.src.C) {
        int n = pp / ohw; int rem = pp - n * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        int ih = oh * p.stride - p.pad + kh * p.dil, iw = ow * p.stride - p.pad + kw * p.dil;
        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
          const float* xp = p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + c;
          const int nrem = p.src.C - c;
          v = ld4g(xp, nrem, p.vecZ);
          zmask |= 1u << j;
        }
      }
      rz[j] = v;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCY / 4), q = slot - row * (BCY / 4);
      if (row < KP) lds_st4(&Ys[row * LY + 4 * q], ry[j]);
    }
#pragma unroll
    for (int j = 0; j < NZJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
      float4 v = rz[j];
      if (zmask & (1u << j)) {      // lazy prologue, applied after the MFMAs of the previous step
        const int nrem = p.src.C - (c0 + 4 * q);
        v.x = fmaf(za[j].x, v.x, zb[j].x); v.y = fmaf(za[j].y, v.y, zb[j].y);
        v.z = fmaf(za[j].z, v.z, zb[j].z); v.w = fmaf(za[j].w, v.w, zb[j].w);
        if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (nrem < 4) { if (nrem < 2) v.y = 0.f; if (nrem < 3) v.z = 0.f; v.w = 0.f; }
      }
      if (row < KP) lds_st4(&Zs[row * LZ + 4 * q], v);
    }
  };

  if (pbeg < pend) {
    load_step(pbeg);
    store_step();
    __syncthreads();
    for (int p0 = pbeg; p0 < pend; p0 += KP) {
      const bool more = p0 + KP < pend;
      if (more) load_step(p0 + KP);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int px = wave * 16 + ks * 4 + kq;
        float yf[CTY], zf[CTZ];
#pragma unroll
        for (int i = 0; i < CTY; ++i) yf[i] = Ys[px * LY + i * 16 + li];
#pragma unroll
        for (int j = 0; j < CTZ; ++j) zf[j] = Zs[px * LZ + j * 16 + li];
#pragma unroll
        for (int i = 0; i < CTY; ++i)
#pragma unroll
          for (int j = 0; j < CTZ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[i], zf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (more) { store_step(); __syncthreads(); }
    }
  }

  // combine the four waves in a fixed order (deterministic), tile layout [co][c]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < CTY; ++i)
#pragma unroll
        for (int j = 0; j < CTZ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int idx = (i * 16 + kq * 4 + r) * BCZ + j * 16 + li;
            lds[idx] = (w == 0) ? acc[i][j][r] : lds[idx] + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  // workspace layout: [split][co][tap][c] over the real (unpadded) extents
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * p.taps * C;
  for (int idx = t; idx < TILE; idx += 256) {
    int r = idx / BCZ, cc = idx - r * BCZ;
    int co = co0 + r, c = c0 + cc;
    if (co < p.Cout && c < C) wsb[((long)co * p.taps + tap) * C + c] = lds[idx];
  }
}

// Output-split variant for the 256-wide heads (ASPP / decoder): the four waves form a 2x2 grid over a
// (32*TY) x (32*TZ) output tile, every wave walks ALL staged pixels and owns a TYxTZ block of 16x16 accumulators
// (32-64 VGPRs instead of the 128 of the pixel-split form), so 2-3 blocks fit a CU and staging overlaps the MFMAs;
// no cross-wave reduction is needed.
template <int TY, int TZ, bool BATCH>
__global__ void __launch_bounds__(256) wgrad_os_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int BCY = 32 * TY, BCZ = 32 * TZ;
  constexpr int LY = ldpad(BCY), LZ = ldpad(BCZ);
  constexpr int NYJ = (KP * BCY / 4 + 255) / 256;
  constexpr int NZJ = (KP * BCZ / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[KP * LY + KP * LZ];
  float* Ys = lds;
  float* Zs = lds + KP * LY;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int wy = wave >> 1, wz = wave & 1;
  int bx = blk_x;
  const int zt = bx % p.nzt; bx /= p.nzt;
  const int tap = bx % p.taps; const int yt = bx / p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = yt * BCY, c0 = zt * BCZ;
  const int ohw = p.OH * p.OW;
  const int pbeg = blk_y * p.chunkP;
  int pend = pbeg + p.chunkP; if (pend > p.P) pend = p.P;
  float4 za[NZJ], zb[NZJ];
#pragma unroll
  for (int j = 0; j < NZJ; ++j) {
    int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
    int c = c0 + 4 * q;
    za[j] = make_float4(1.f, 1.f, 1.f, 1.f); zb[j] = zero4();
    if (p.src.a && row < KP && c < p.src.C) { za[j] = ld4g(p.src.a + c, p.src.C - c, p.vecZ); zb[j] = ld4g(p.src.b + c, p.src.C - c, p.vecZ); }
  }
  const bool zrelu = p.src.relu != 0;

  f32x4 acc[TY][TZ];
#pragma unroll
  for (int i = 0; i < TY; ++i)
#pragma unroll
    for (int j = 0; j < TZ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ry[NYJ], rz[NZJ];
  unsigned zmask = 0;
  auto load_step = [&](int p0) {
    zmask = 0;
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCY / 4), q = slot - row * (BCY / 4);
      int pp = p0 + row; int co = co0 + 4 * q;
      float4 v = zero4();
      if (row < KP && pp < pend && co < p.Cout) v = ld4g(p.dy + (long)pp * p.lddy + co, p.Cout - co, p.vecY);
      ry[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NZJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
      int pp = p0 + row; int c = c0 + 4 * q;
      float4 v = zero4();
      if (row < KP && pp < pend && c < p.src.C) {
        int n = pp / ohw; int rem = pp - n * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        int ih = oh * p.stride - p.pad + kh * p.dil, iw = ow * p.stride - p.pad + kw * p.dil;
        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
          v = ld4g(p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + c, p.src.C - c, p.vecZ);
          zmask |= 1u << j;
        }
      }
      rz[j] = v;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCY / 4), q = slot - row * (BCY / 4);
      if (row < KP) lds_st4(&Ys[row * LY + 4 * q], ry[j]);
    }
#pragma unroll
    for (int j = 0; j < NZJ; ++j) {
      int slot = t + 256 * j, row = slot / (BCZ / 4), q = slot - row * (BCZ / 4);
      float4 v = rz[j];
      if (zmask & (1u << j)) {
        const int nrem = p.src.C - (c0 + 4 * q);
        v.x = fmaf(za[j].x, v.x, zb[j].x); v.y = fmaf(za[j].y, v.y, zb[j].y);
        v.z = fmaf(za[j].z, v.z, zb[j].z); v.w = fmaf(za[j].w, v.w, zb[j].w);
        if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (nrem < 4) { if (nrem < 2) v.y = 0.f; if (nrem < 3) v.z = 0.f; v.w = 0.f; }
      }
      if (row < KP) lds_st4(&Zs[row * LZ + 4 * q], v);
    }
  };

  if (pbeg < pend) {
    load_step(pbeg);
    store_step();
    __syncthreads();
    for (int p0 = pbeg; p0 < pend; p0 += KP) {
      const bool more = p0 + KP < pend;
      if (more) load_step(p0 + KP);
#pragma unroll 4
      for (int ks = 0; ks < KP / 4; ++ks) {
        const int px = ks * 4 + kq;
        float yf[TY], zf[TZ];
#pragma unroll
        for (int i = 0; i < TY; ++i) yf[i] = Ys[px * LY + (wy * TY + i) * 16 + li];
#pragma unroll
        for (int j = 0; j < TZ; ++j) zf[j] = Zs[px * LZ + (wz * TZ + j) * 16 + li];
#pragma unroll
        for (int i = 0; i < TY; ++i)
#pragma unroll
          for (int j = 0; j < TZ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[i], zf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (more) { store_step(); __syncthreads(); }
    }
  }
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * p.taps * C;
#pragma unroll
  for (int i = 0; i < TY; ++i)
#pragma unroll
    for (int j = 0; j < TZ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int co = co0 + (wy * TY + i) * 16 + kq * 4 + r, c = c0 + (wz * TZ + j) * 16 + li;
        if (co < p.Cout && c < C) wsb[((long)co * p.taps + tap) * C + c] = acc[i][j][r];
      }
}

// Halo-patch variant for the wide 3x3 stride-1 convolutions (decoder, ASPP dilated branches, stem1): a block owns
// (64*NT output channels) x (16 input channels) x ALL NINE taps and walks its pixel range one 64-pixel row segment at a
// time.  Per segment it stages dy[64 px][64*NT] once and the three activation rows oh-d, oh, oh+d of the 16 channels
// ([3][64+2d px][16], BatchNorm/ReLU applied on the way in, zero padding after it) once, and every tap reads its
// shifted window of that patch from LDS: 64*NT + 48 floats fetched per pixel for 9*16*64*NT MACs, against 64*NT+64 per
// pixel PER TAP for the per-tap kernels above (2.5x the arithmetic intensity at NT=2, 1/9 of the global load
// instructions).  Wave w keeps co tiles [w*NT, w*NT+NT) x 9 taps = 9*NT 16x16 accumulators.
constexpr int H3_KP = 64;
constexpr int H3_ZW = H3_KP + 2 * 18;       // widest patch row (dilation 18)

template <int NT, bool BATCH>
__global__ void __launch_bounds__(256, 2) wgrad_h3_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int BCO = 64 * NT, LY = BCO + 16, YQ = BCO / 4, YRS = 256 / YQ;
  constexpr int NYJ = H3_KP / YRS;
  constexpr int NZJ = (3 * H3_ZW * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ys[H3_KP * LY];
  __shared__ __attribute__((aligned(16))) float Zs[3 * H3_ZW * 16];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int zt = blk_x % p.nzt, yt = blk_x / p.nzt;
  const int co0 = yt * BCO, c0 = zt * 16;
  const int d = p.dil, ZW = H3_KP + 2 * d;
  const int spr = (p.OW + H3_KP - 1) / H3_KP;          // segments per image row
  const int nseg = p.N * p.OH * spr;
  const int sbeg = blk_y * p.chunkP;
  int send = sbeg + p.chunkP; if (send > nseg) send = nseg;

  // fixed slot geometry
  const int yq = t & (YQ - 1), yrow0 = t / YQ;
  const int co = co0 + 4 * yq;
  const bool co_ok = co < p.Cout;
  const int zq = t & 3, zc = c0 + 4 * zq, nremz = p.src.C - zc;
  int zr[NZJ], zj[NZJ];
#pragma unroll
  for (int k = 0; k < NZJ; ++k) {
    const int pix = (t + 256 * k) >> 2;
    zr[k] = pix / ZW; zj[k] = pix - zr[k] * ZW;         // zr >= 3 marks a slot outside the patch
  }
  float4 za = make_float4(1.f, 1.f, 1.f, 1.f), zb = zero4();
  if (p.src.a && nremz > 0) { za = ld4g(p.src.a + zc, nremz, p.vecZ); zb = ld4g(p.src.b + zc, nremz, p.vecZ); }
  const bool zrelu = p.src.relu != 0;
  int zbase[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) zbase[tap] = (((tap / 3) * H3_ZW) + kq + (tap % 3) * d) * 16 + li;

  f32x4 acc[NT][9];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Branch-free staging: every slot always issues its 16-byte load (from a safe address when it is masked) so the
  // loads of a step go out back to back; masked slots are zeroed when they are written to LDS.
  float4 ry[NYJ], rz[NZJ];
  unsigned ymask = 0, zmask = 0;
  auto load_step = [&](int seg) {
    const int rowid = seg / spr, sx = seg - rowid * spr;
    const int n = rowid / p.OH, oh = rowid - n * p.OH;
    const int ow0 = sx * H3_KP;
    const long pp0 = (long)rowid * p.OW + ow0;
    const float* yb = p.dy + pp0 * p.lddy + co;
    ymask = 0; zmask = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      const int row = yrow0 + k * YRS;
      const bool ok = co_ok && ow0 + row < p.OW;
      ry[k] = ld4(ok ? yb + (long)row * p.lddy : p.dy);
      ymask |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      const int ih = oh + (zr[k] - 1) * d, iw = ow0 - d + zj[k];
      const bool ok = zr[k] < 3 && nremz > 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rz[k] = ld4(ok ? p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + zc : p.src.x);
      zmask |= (ok ? 1u : 0u) << k;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      float4 v = ry[k];
      const bool ok = (ymask >> k) & 1u;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      lds_st4(&Ys[(yrow0 + k * YRS) * LY + 4 * yq], v);
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      float4 v = rz[k];
      v.x = fmaf(za.x, v.x, zb.x); v.y = fmaf(za.y, v.y, zb.y); v.z = fmaf(za.z, v.z, zb.z); v.w = fmaf(za.w, v.w, zb.w);
      if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const bool ok = (zmask >> k) & 1u;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (zr[k] < 3) lds_st4(&Zs[(zr[k] * H3_ZW + zj[k]) * 16 + 4 * zq], v);
    }
  };

  if (sbeg < send) {
    load_step(sbeg);
    store_step();
    __syncthreads();
    const float* yw = &Ys[kq * LY + wave * NT * 16 + li];
    for (int seg = sbeg; seg < send; ++seg) {
      const bool more = seg + 1 < send;
      if (more) load_step(seg + 1);
      // software-pipelined fragment reads: the LDS reads of k-step s+1 are in flight while the 9*NT MFMAs of k-step s issue
      float yfA[NT], zfA[9], yfB[NT], zfB[9];
      auto rd = [&](int s4, float* yf, float* zf) {
#pragma unroll
        for (int i = 0; i < NT; ++i) yf[i] = yw[s4 * 4 * LY + i * 16];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) zf[tap] = Zs[zbase[tap] + s4 * 64];
      };
      auto mma = [&](const float* yf, const float* zf) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
            acc[i][tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[i], zf[tap], acc[i][tap], 0, 0, 0);
      };
      rd(0, yfA, zfA);
#pragma unroll
      for (int s4 = 0; s4 < H3_KP / 4; s4 += 2) {
        rd(s4 + 1, yfB, zfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(yfA, zfA);
        __builtin_amdgcn_sched_barrier(0);
        if (s4 + 2 < H3_KP / 4) rd(s4 + 2, yfA, zfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(yfB, zfB);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (more) { store_step(); __syncthreads(); }
    }
  }
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * 9 * C;
  const int c = c0 + li;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cow = co0 + (wave * NT + i) * 16 + kq * 4 + r;
      if (cow < p.Cout && c < C) {
        gfloat* o = wsb + (long)cow * 9 * C + c;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) o[tap * C] = acc[i][tap][r];
      }
    }
}

// Split-bf16 form of wgrad_h3_kernel (same work decomposition, same partial-tile layout, same epilogue): dy and the
// activations are written as h + m + l in bf16 (exact) and a product is the sum of its six largest bf16 x bf16 terms on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (conv3.hip: as accurate as the fp32 MFMA chain at 2.5x its rate).
// The MFMA's K axis is the PIXEL index here, while both operands arrive pixel-major ([pixel][channel] rows from NHWC
// memory): the staged LDS images stay pixel-major — per 16-channel tile [pixel][16 ch] bf16, 32-byte rows, 8-byte writes —
// and the fragments are fetched with the hardware transposing read ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-pixel x 16-channel block and every lane receives ITS channel's four pixels): two reads give a lane the 8 consecutive
// pixels of its channel that the 16x16x32 operand wants, for dy and for every tap-shifted window of the activation rows
// alike (any pixel shift is a row offset: always aligned).  A pixel's row sits at 32*p with the two 128-byte halves of
// every 8-pixel block swapped when bit 3 of p is set: the two groups of a half-wave (pixels p..p+3 and p+8..p+11) then hit
// disjoint banks for every shift (scripts/tr_read_probe.hip checks the lane map and the operand on the device).
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) wg_s16x4 wg_lds_s16x4;
__device__ __forceinline__ unsigned wg_bf16_hi(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float wg_bf16_f(unsigned b) { return __uint_as_float(b << 16); }
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one packed bf16 pair (round to nearest even): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned wg_cvt2(float a, float b) {
  const wg_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_bf16x2));
}
// [r5] NP = 2 is the split-fp16 form (common.h / conv3b.h): both operands of a weight gradient are activations, so BOTH carry a running power-of-two scale
// per workgroup — the largest magnitude of each staged segment goes through LDS in front of the barrier that ends the matrix phase (wg_publish_max), behind it
// every thread folds the waves' maxima into the two scales and, when a larger segment arrives, multiplies the accumulators by the exact ratio (wg_rescale).
// The partial tiles leave the kernel unscaled.
struct WgScale { int kfy, kfz; };
__device__ __forceinline__ void wg_publish_max(unsigned* wmx, int nw, int wave, int lane, unsigned my, unsigned mz) {
  my = wave_umax(my); mz = wave_umax(mz);
  if (lane == 0) { wmx[wave] = my; wmx[nw + wave] = mz; }
}
// returns the factor for the accumulators (1 = unchanged)
__device__ __forceinline__ float wg_rescale(const unsigned* wmx, int nw, WgScale& sc) {
  unsigned my = 0, mz = 0;
  for (int w = 0; w < nw; ++w) { const unsigned a = wmx[w], b = wmx[nw + w]; my = a > my ? a : my; mz = b > mz ? b : mz; }
  const int wy = f16_scale_field(my), wz = f16_scale_field(mz);
  int sh = 0;
  if (sc.kfy == 0) sc.kfy = wy; else if (wy < sc.kfy) { sh += wy - sc.kfy; sc.kfy = wy; }
  if (sc.kfz == 0) sc.kfz = wz; else if (wz < sc.kfz) { sh += wz - sc.kfz; sc.kfz = wz; }
  if (sh == 0) return 1.f;
  const int rf = 127 + sh;
  return rf > 0 ? __uint_as_float((unsigned)rf << 23) : 0.f;
}
__device__ __forceinline__ float wg_pow2(int field) { return __uint_as_float((unsigned)field << 23); }
__device__ __forceinline__ float4 wg_mul4(float4 v, float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; return v; }
template <int NP>
__device__ __forceinline__ void wg_split4(const float4 v, uint2 (&pl)[NP]) {
  if constexpr (NP == 2) { split4h(v, pl); return; }
  float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const unsigned p0 = wg_cvt2(a, b), p1 = wg_cvt2(c, d);
    pl[k] = make_uint2(p0, p1);
    if (k + 1 < NP) {
      a -= __uint_as_float(p0 << 16); b -= __uint_as_float(p0 & 0xffff0000u);
      c -= __uint_as_float(p1 << 16); d -= __uint_as_float(p1 & 0xffff0000u);
    }
  }
}
// byte offset of pixel row p inside a [pixel][16 ch] bf16 tile image
__device__ __forceinline__ int wg_prow(int p) { return (p << 5) ^ (((p >> 3) & 1) << 7); }

#ifdef ADDK_WG_DIAG
// diagnostic build (scripts/wgrad_phases.sh): every wave of wgrad_h3b_kernel adds its lifetime in shader-clock ticks (s_memtime) and in 100 MHz reference ticks
// (s_memrealtime) — their ratio is the clock the CUs ran at inside the kernel — and the shader ticks it spent in each phase of the segment loop:
// [0] life (shader) [1] life (reference) [2] waves [3] preparing the next segment's addresses [4] matrix phase (fragment reads + MFMA + the next segment's loads) [5] the split into registers
// (including the wait for the loads) [6] waiting at the barrier behind it [7] LDS stores and the barrier behind them
__device__ unsigned long long g_wg_diag[64][8];
#ifdef ADDK_WG_DIAG2
__device__ unsigned long long g_wg_diag2[64][2];
#endif
#define WG_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#endif
// NG = 2: a 512-thread workgroup of two wave groups, one per 16-channel input tile, which SHARE the staged dy tile (two 256-thread workgroups of
// neighbouring input tiles — which sit on one CU and run their phases together anyway: scripts/wgrad_phases.sh — split and store the same dy twice)
template <int NT, bool BATCH, int NP, int NG>
__global__ void __launch_bounds__(256 * NG, 2) wgrad_h3b_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int NTHR = 256 * NG, BCO = 64 * NT, YT = BCO / 16, YQ = BCO / 4, YRS = NTHR / YQ;
  constexpr int NYJ = H3_KP / YRS;
  constexpr int ZWP = 104;                                        // patch row pitch in pixels (>= 64 + 2*18, multiple of 8: the swizzle works on 8-pixel blocks)
  // bytes per dy tile image / per activation patch row (one plane).  The tile images are 32 bytes apart from a multiple of
  // the 256-byte bank period: the 8 tiles x 4 channel quads a half-wave stores for one pixel row then cover all 64 banks once
  constexpr int YIMG = H3_KP * 32 + 32, ZROW = ZWP * 32;
  // one input-channel tile of the patch (one plane); with two tiles, 64 bytes off the 256-byte bank period: the two 32-byte halves a pixel's eight
  // channel quads store then hit different banks (SQ_LDS_BANK_CONFLICT 9.5 % of the LDS cycles without)
  constexpr int ZTILE = 3 * ZROW + (NG > 1 ? 64 : 0);
  constexpr int YPL = YT * YIMG, ZPL = NG * ZTILE;                // bytes per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  unsigned char* Yb = wsm;                                        // [NP][YT][64 px][16 co]
  unsigned char* Zb = wsm + NP * YPL;                             // [NP][NG tiles][3 rows][ZWP px][16 ci]
  unsigned* wmx = reinterpret_cast<unsigned*>(wsm + NP * (YPL + ZPL));      // NP = 2: [2][4 NG] the waves' largest dy / activation magnitudes of the segment being staged
  WgScale fsc = {0, 0};

  const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6), wave = wave8 & 3, grp = wave8 >> 2;      // output-channel tiles of this wave; its input-channel tile
  const int zt = blk_x % p.nzt, yt = blk_x / p.nzt;
  const int co0 = yt * BCO, c0 = zt * 16 * NG;
  const int d = p.dil, ZW = H3_KP + 2 * d;
  const int spr = (p.OW + H3_KP - 1) / H3_KP;
  const int nseg = p.N * p.OH * spr;
  const int sbeg = blk_y * p.chunkP;
  int send = sbeg + p.chunkP; if (send > nseg) send = nseg;

  // Staging geometry.  Everything a thread needs per segment is a THREAD CONSTANT (a byte offset from a segment-uniform base pointer, an LDS
  // offset) plus segment scalars: no per-slot divisions, address arithmetic or validity bits in vector registers.
  //   dy: thread (yq, yrow0) owns channel quad yq of rows yrow0 + k YRS of the 64-pixel segment;
  //   patch: thread (zq, zj0) owns channel quad zq of patch columns zj0 and 64 + zj0 (the latter only below 2 d) of each of the 3 rows.
  // The NEXT segment's global loads are issued one per tap INSIDE the matrix phase (they are branch-free: a lane without a valid element
  // reads element 0 of a valid row and is masked when the patch is stored): measured with the in-kernel phase clock (scripts/wgrad_phases.sh),
  // issuing 14 KB of loads per wave in one burst held every wave for 11 % of its life at the CU's 64 B/clk address path.
  constexpr int NIT = NYJ + 6;                                      // load items per segment: NYJ dy rows, 3 patch rows x 2 halves
  static_assert(NIT <= 18, "one load item per (k-step, tap)");
  const int yq = t & (YQ - 1), yrow0 = t / YQ;
  const int co = co0 + 4 * yq;                                      // < Cout: BCO divides Cout (wg_fill)
  const unsigned yoff = ((unsigned)yrow0 * (unsigned)p.lddy + (unsigned)co) * 4u;
  const long ystep = (long)YRS * p.lddy;
  // LDS offset of row yrow0 + k YRS = (k even ? ysw0 : ysw1) + k YRS 32: the swizzle bit (bit 3 of the row) alternates with k when YRS = 8
  const int ytile = (yq >> 2) * YIMG + 8 * (yq & 3);
  const int ysw0 = ytile + wg_prow(yrow0), ysw1 = ytile + wg_prow(yrow0 + YRS) - (YRS << 5);
  constexpr int ZQ = 4 * NG;                                        // channel quads per patch pixel
  const int zq = t & (ZQ - 1), zj0 = t / ZQ, zc = c0 + 4 * zq, nremz = p.src.C - zc;
  const unsigned zldb = (unsigned)p.src.ld * 4u;
  const unsigned zoff = (unsigned)zj0 * zldb + (unsigned)zc * 4u;     // byte offset of (patch column zj0, channel zc) from the patch row's column 0
  const int zsw = (zq >> 2) * ZTILE + wg_prow(zj0) + 8 * (zq & 3);  // second half: + 64 * 32 (bit 3 of 64 + zj0 is bit 3 of zj0)
  const bool zhalf1 = zj0 < 2 * d;                                  // this thread has a column in the second half (64 + zj0 < ZW)
  float4 za = make_float4(1.f, 1.f, 1.f, 1.f), zb = zero4();
  if (p.src.a && nremz > 0) { za = ld4g(p.src.a + zc, nremz, p.vecZ); zb = ld4g(p.src.b + zc, nremz, p.vecZ); }
  const bool zrelu = p.src.relu != 0, zaff = p.src.a != nullptr;
  // transposed-read lane geometry: lane 16g + 4q + pp supplies (pixel row q of the block, channels 4pp..4pp+3)
  const int tq = li >> 2, tp = li & 3;
  const int lrow = 8 * kq + tq;                                   // this lane's pixel row inside a 32-pixel k-step (first read; second +4)

  f32x4 acc[NT][9];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ry[NYJ], rz[6];
  // l_*: the segment load_prep turns to next; c_*: the one it prepared last (scalars, advanced incrementally)
  int l_sx, l_oh, l_n, c_sx, c_oh, c_n;
  { const int rowid = sbeg / spr; l_sx = sbeg - rowid * spr; l_n = rowid / p.OH; l_oh = rowid - l_n * p.OH; c_sx = l_sx; c_oh = l_oh; c_n = l_n; }
  // what the load items of the prepared segment need, and the validity the store of that segment needs
  const float* yseg = p.dy; const float* zrowp[3] = {p.src.x, p.src.x, p.src.x};
  unsigned zo0 = 0, zo1 = 0;                                        // this lane's byte offsets in a patch row's image row (0: no valid element)
  int st_skip = 0;                                                  // leading pixels of the segment that belong to its left neighbour
  unsigned zrows = 0;                                               // bit r: patch row r lies inside the image
  unsigned long long zcm0 = 0, zcm1 = 0;                            // lane masks: this lane's first / second column lies inside the image (and its channels exist)
#ifdef ADDK_WG_DIAG2
  unsigned long long dsub[2] = {0, 0};        // shader ticks inside store_step: waiting for the loads, the dy part (split + LDS stores, drained)
#endif
  auto load_prep = [&](bool next) {                                 // next == false: prepare the last segment again (a harmless reload behind the block's final matrix phase)
    if (next) { c_sx = l_sx; c_oh = l_oh; c_n = l_n; if (++l_sx == spr) { l_sx = 0; if (++l_oh == p.OH) { l_oh = 0; ++l_n; } } }
    // the last segment of an image row is moved left to end at the row's end (OW >= 64: h3_ok); the st_skip pixels it then shares with its
    // neighbour get dy = 0 in store_step — every segment is a full one
    int ow0 = c_sx * H3_KP;
    st_skip = ow0 + H3_KP - p.OW; if (st_skip < 0) st_skip = 0;
    ow0 -= st_skip;
    const long rowid = (long)c_n * p.OH + c_oh;
    yseg = p.dy + (rowid * p.OW + ow0) * p.lddy;
    const int iw0 = ow0 - d;                                        // image column of patch column 0
    const bool c0ok = nremz > 0 && (unsigned)(iw0 + zj0) < (unsigned)p.W;
    const bool c1ok = nremz > 0 && zhalf1 && (unsigned)(iw0 + H3_KP + zj0) < (unsigned)p.W;
    zcm0 = __ballot(c0ok); zcm1 = __ballot(c1ok);
    const unsigned zbase = (unsigned)iw0 * zldb;                    // (wraps for iw0 < 0; a valid lane's sum does not)
    zo0 = c0ok ? zbase + zoff : 0u;
    zo1 = c1ok ? zbase + H3_KP * zldb + zoff : 0u;
    zrows = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ih = c_oh + (r - 1) * d;
      const bool rok = (unsigned)ih < (unsigned)p.H;
      zrows |= (rok ? 1u : 0u) << r;
      zrowp[r] = rok ? p.src.x + (((long)c_n * p.H + ih) * p.W) * p.src.ld : p.src.x;
    }
  };
  auto load_item = [&](int i) {
    if (i < NYJ) ry[i] = ld4so(yseg + i * ystep, yoff);
    else if (i < NIT) rz[i - NYJ] = ld4so(zrowp[(i - NYJ) >> 1], ((i - NYJ) & 1) ? zo1 : zo0);
  };
  // The split runs BEFORE the barrier that ends the matrix phase, into registers (the SIMD's arbiter serves the older of its two waves first: the wave that
  // leaves the matrix phase early splits under the other one's MFMAs instead of waiting at the barrier), the LDS stores behind it.
  constexpr bool PRE_Z = NP != 2 && !(NT == 2 && (NP == 3 || NG == 1));      // (the two-tile six-term forms have no registers for the patch's planes: only dy is split early,
  constexpr bool PRE_Y = NP != 2 && !(NT == 2 && NP == 3 && NG == 1);      //  and nothing at all in the 256-thread form, whose threads hold eight dy rows)
  // (NP = 2, split-fp16: the conversion needs the segment's scale, which exists behind that barrier only — prep2 in front of it, rescale2 + write_step behind)
  uint2 py[NYJ][NP], pz[6][NP];
  auto split_z = [&]() {
    const bool c0ok = __builtin_amdgcn_inverse_ballot_w64(zcm0), c1ok = __builtin_amdgcn_inverse_ballot_w64(zcm1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool rok = (zrows >> r) & 1u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !zhalf1) continue;
        float4 v = rz[2 * r + h];
        if (zaff) { v.x = fmaf(za.x, v.x, zb.x); v.y = fmaf(za.y, v.y, zb.y); v.z = fmaf(za.z, v.z, zb.z); v.w = fmaf(za.w, v.w, zb.w); }
        if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = rok && (h ? c1ok : c0ok);
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        wg_split4<NP>(v, pz[2 * r + h]);
        if (!PRE_Z) {
          unsigned char* o = Zb + zsw + r * ZROW + h * (H3_KP * 32);
#pragma unroll
          for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * ZPL) = pz[2 * r + h][m];
        }
      }
    }
  };
  auto prep2 = [&]() {                                              // NP = 2: masks and prologue in place, the wave's maxima to LDS
    unsigned my = 0, mz = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      if (st_skip && yrow0 + k * YRS < st_skip) ry[k] = zero4();
      const unsigned b = absbits4(ry[k]); my = b > my ? b : my;
    }
    const bool c0ok = __builtin_amdgcn_inverse_ballot_w64(zcm0), c1ok = __builtin_amdgcn_inverse_ballot_w64(zcm1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool rok = (zrows >> r) & 1u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 v = rz[2 * r + h];
        if (zaff) { v.x = fmaf(za.x, v.x, zb.x); v.y = fmaf(za.y, v.y, zb.y); v.z = fmaf(za.z, v.z, zb.z); v.w = fmaf(za.w, v.w, zb.w); }
        if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = rok && (h ? c1ok : c0ok) && (h == 0 || zhalf1);
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        rz[2 * r + h] = v;
        const unsigned b = absbits4(v); mz = b > mz ? b : mz;
      }
    }
    wg_publish_max(wmx, 4 * NG, wave8, lane, my, mz);
  };
  auto rescale2 = [&]() {
    const float r = wg_rescale(wmx, 4 * NG, fsc);
    if (r != 1.f) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] *= r;
    }
  };
  auto split_step = [&]() {
    if (NP == 2) { prep2(); return; }
#ifdef ADDK_WG_DIAG2
    WG_STAMP(sa);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(sb);
    dsub[0] += sb - sa;
#endif
    if (st_skip) {
#pragma unroll
      for (int k = 0; k < NYJ; ++k) if (yrow0 + k * YRS < st_skip) ry[k] = zero4();
    }
    if (PRE_Y) {
#pragma unroll
      for (int k = 0; k < NYJ; ++k) wg_split4<NP>(ry[k], py[k]);
    }
#ifdef ADDK_WG_DIAG2
    WG_STAMP(sc);
    dsub[1] += sc - sb;
#endif
    if (PRE_Z) split_z();
  };
  auto write_step = [&]() {
    if (NP == 2) {
      const float sy = wg_pow2(fsc.kfy), sz = wg_pow2(fsc.kfz);
#pragma unroll
      for (int k = 0; k < NYJ; ++k) {
        wg_split4<NP>(wg_mul4(ry[k], sy), py[0]);
        unsigned char* o = Yb + ((k & 1) ? ysw1 : ysw0) + k * (YRS << 5);
#pragma unroll
        for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * YPL) = py[0][m];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !zhalf1) continue;
          wg_split4<NP>(wg_mul4(rz[2 * r + h], sz), pz[0]);
          unsigned char* o = Zb + zsw + r * ZROW + h * (H3_KP * 32);
#pragma unroll
          for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * ZPL) = pz[0][m];
        }
      return;
    }
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      if (!PRE_Y) wg_split4<NP>(ry[k], py[k]);
      unsigned char* o = Yb + ((k & 1) ? ysw1 : ysw0) + k * (YRS << 5);
#pragma unroll
      for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * YPL) = py[k][m];
    }
    if (!PRE_Z) { split_z(); return; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !zhalf1) continue;
        unsigned char* o = Zb + zsw + r * ZROW + h * (H3_KP * 32);
#pragma unroll
        for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * ZPL) = pz[2 * r + h][m];
      }
  };
  // fragment of one 32-pixel k-step: 8 consecutive pixels of this lane's channel = two transposed reads, per plane
  auto rd = [&](const unsigned char* base, int plane_bytes, int pix0, wg_bf16x8* f) {
    const int o0 = wg_prow(pix0 + lrow) + 8 * tp, o1 = wg_prow(pix0 + lrow + 4) + 8 * tp;
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o0));
      const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o1));
      // whole-register reinterpretation: building the fragment element by element from the two results is miscompiled by
      // hipcc 7.2 (it drops the upper dword of each 64-bit result: scripts/tr_read_probe.hip)
      struct { wg_s16x4 a, b; } pr = {lo, hi};
      f[m] = __builtin_bit_cast(wg_bf16x8, pr);
    }
  };
  // one tap: the product terms smallest first, the NT accumulator chains interleaved term by term
  auto mma = [&](f32x4 (&c)[NT][9], int tap, const wg_bf16x8 (&y)[NT][NP], const wg_bf16x8* z) {
#define WG_TERM(YI, ZI) _Pragma("unroll") for (int i = 0; i < NT; ++i) c[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[i][YI], z[ZI], c[i][tap], 0, 0, 0);
#define WG_TERMH(YI, ZI) _Pragma("unroll") for (int i = 0; i < NT; ++i) c[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, y[i][YI]), __builtin_bit_cast(f16x8, z[ZI]), c[i][tap], 0, 0, 0);
    if constexpr (NP == 2) { WG_TERMH(1, 0) WG_TERMH(0, 1) WG_TERMH(0, 0) } else {
    if (NP == 3) { WG_TERM(2, 0) WG_TERM(0, 2) WG_TERM(1, 1) }
    WG_TERM(1, 0) WG_TERM(0, 1) WG_TERM(0, 0) }
#undef WG_TERM
#undef WG_TERMH
  };

#ifdef ADDK_WG_DIAG
  const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long dph[5] = {0, 0, 0, 0, 0};
#endif
  const unsigned char* Zg = Zb + grp * ZTILE;
  if (sbeg < send) {
    load_prep(true);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_item(i);
    split_step();
    if (NP == 2) { __syncthreads(); rescale2(); }
    write_step();
    __syncthreads();
    for (int seg = sbeg; seg < send; ++seg) {
      const bool more = seg + 1 < send;
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt0);
#endif
      load_prep(more);
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt1);
#endif
#pragma unroll
      for (int ks = 0; ks < H3_KP / 32; ++ks) {
        wg_bf16x8 yf[NT][NP];
#pragma unroll
        for (int i = 0; i < NT; ++i) rd(Yb + (wave * NT + i) * YIMG, YPL, ks * 32, yf[i]);
        wg_bf16x8 zf[2][NP];
        rd(Zg, ZPL, ks * 32, zf[0]);                              // tap 0: patch row 0, shift 0
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          load_item(ks * 9 + tap);
          if (tap + 1 < 9) rd(Zg + ((tap + 1) / 3) * ZROW, ZPL, ks * 32 + ((tap + 1) % 3) * d, zf[(tap + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc, tap, yf, zf[tap & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt2);
#endif
      if (more) split_step();
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt3);
#endif
      __syncthreads();
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt4);
#endif
      if (more) { if (NP == 2) rescale2(); write_step(); __syncthreads(); }
#ifdef ADDK_WG_DIAG
      WG_STAMP(dt5);
      dph[0] += dt1 - dt0; dph[1] += dt2 - dt1; dph[2] += dt3 - dt2; dph[3] += dt4 - dt3; dph[4] += dt5 - dt4;
#endif
    }
  }
#ifdef ADDK_WG_DIAG
  if (lane == 0) {
    unsigned long long* dslot = g_wg_diag[(blockIdx.x * 4 + wave8) & 63];
    atomicAdd(&dslot[0], __builtin_amdgcn_s_memtime() - diag_c0); atomicAdd(&dslot[1], __builtin_amdgcn_s_memrealtime() - diag_r0); atomicAdd(&dslot[2], 1ull);
    for (int i = 0; i < 5; ++i) atomicAdd(&dslot[3 + i], dph[i]);
#ifdef ADDK_WG_DIAG2
    atomicAdd(&dslot[1], 0ull); atomicAdd(&g_wg_diag2[(blockIdx.x * 4 + wave8) & 63][0], dsub[0]); atomicAdd(&g_wg_diag2[(blockIdx.x * 4 + wave8) & 63][1], dsub[1]);
#endif
  }
#endif
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * 9 * C;
  const int c = c0 + 16 * grp + li;
  if (NP == 2) {                       // the two operand scales leave the partial tile (exact; one after the other)
    const float iy = wg_pow2(254 - fsc.kfy), iz = wg_pow2(254 - fsc.kfz);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[i][j] = acc[i][j] * iy * iz;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cow = co0 + (wave * NT + i) * 16 + kq * 4 + r;
      if (cow < p.Cout && c < C) {
        gfloat* o = wsb + (long)cow * 9 * C + c;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) o[tap * C] = acc[i][tap][r];
      }
    }
}
// [r4] The same arithmetic and LDS images for the wide 1x1 heads (ASPP's 1x1 branch and its 1280 -> 256 concat conv, 256 <- 256..400 channels at
// 64x128: aspp_train.py:34-58): no halo, so the nine accumulator "taps" of wgrad_h3b_kernel become FOUR 16-channel input tiles per workgroup
// (128 output x 64 input channels, 96 MFMA per wave and 64-pixel segment).  These weight gradients ran on the fp32 MFMA kernel (wgrad_os_kernel<4,2>:
// 11.9 GF per exit in 200 us = 60 TFLOP/s) while their forward and data gradient already used the split-bf16 kernel.
constexpr int H1_TP = 4;
template <bool BATCH, int NP>
__global__ void __launch_bounds__(256, 2) wgrad_h1b_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int NT = 2, TP = H1_TP, BCO = 64 * NT, YT = BCO / 16, YQ = BCO / 4, YRS = 256 / YQ, NYJ = H3_KP / YRS;
  constexpr int YIMG = H3_KP * 32 + 32;                              // a [64 px][16 ch] bf16 tile image, 32 bytes off the bank period (see wgrad_h3b_kernel)
  constexpr int YPL = YT * YIMG, ZPL = TP * YIMG;
  constexpr int NIT = NYJ + TP;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  unsigned char* Yb = wsm;                                           // [NP][YT][64 px][16 co]
  unsigned char* Zb = wsm + NP * YPL;                                // [NP][TP][64 px][16 ci]
  unsigned* wmx = reinterpret_cast<unsigned*>(wsm + NP * (YPL + ZPL));      // NP = 2: [2][4] the waves' largest magnitudes of the segment being staged (wgrad_h3b_kernel)
  WgScale fsc = {0, 0};
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int zt = blk_x % p.nzt, yt = blk_x / p.nzt;
  const int co0 = yt * BCO, c0 = zt * 16 * TP;
  const int spr = (p.OW + H3_KP - 1) / H3_KP;
  const int nseg = p.N * p.OH * spr;
  const int sbeg = blk_y * p.chunkP;
  int send = sbeg + p.chunkP; if (send > nseg) send = nseg;
  // staging geometry: thread constants + segment scalars (wgrad_h3b_kernel); the activation tile needs no row / column validity, only the channel tail
  const int yq = t & (YQ - 1), yrow0 = t / YQ;
  const int co = co0 + 4 * yq;
  const unsigned yoff = ((unsigned)yrow0 * (unsigned)p.lddy + (unsigned)co) * 4u;
  const long ystep = (long)YRS * p.lddy;
  const int ytile = (yq >> 2) * YIMG + 8 * (yq & 3);
  const int ysw0 = ytile + wg_prow(yrow0), ysw1 = ytile + wg_prow(yrow0 + YRS) - (YRS << 5);
  const int zq = t & 3, zj0 = t >> 2;
  const int zsw = wg_prow(zj0) + 8 * zq;
  unsigned zoffk[TP]; unsigned zvalid = 0;
  float4 za[TP], zb[TP];
  const bool zrelu = p.src.relu != 0, zaff = p.src.a != nullptr;
#pragma unroll
  for (int k = 0; k < TP; ++k) {
    const int zc = c0 + 16 * k + 4 * zq;
    const bool ok = zc < p.src.C;                                   // whole quads: C % 4 == 0 (h1_ok)
    zvalid |= (ok ? 1u : 0u) << k;
    zoffk[k] = ok ? ((unsigned)zj0 * (unsigned)p.src.ld + (unsigned)zc) * 4u : 0u;
    za[k] = make_float4(1.f, 1.f, 1.f, 1.f); zb[k] = zero4();
    if (zaff && ok) { za[k] = ld4(p.src.a + zc); zb[k] = ld4(p.src.b + zc); }
  }
  const int tq = li >> 2, tp = li & 3;
  const int lrow = 8 * kq + tq;
  f32x4 acc[NT][TP];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ry[NYJ], rz[TP];
  int l_sx, l_oh, l_n, c_sx, c_oh, c_n;
  { const int rowid = sbeg / spr; l_sx = sbeg - rowid * spr; l_n = rowid / p.OH; l_oh = rowid - l_n * p.OH; c_sx = l_sx; c_oh = l_oh; c_n = l_n; }
  const float* yseg = p.dy; const float* zseg = p.src.x;
  int st_skip = 0;
  auto load_prep = [&](bool next) {
    if (next) { c_sx = l_sx; c_oh = l_oh; c_n = l_n; if (++l_sx == spr) { l_sx = 0; if (++l_oh == p.OH) { l_oh = 0; ++l_n; } } }
    int ow0 = c_sx * H3_KP;
    st_skip = ow0 + H3_KP - p.OW; if (st_skip < 0) st_skip = 0;      // the last segment of an image row is moved left; the pixels it shares get dy = 0
    ow0 -= st_skip;
    const long pix = ((long)c_n * p.OH + c_oh) * p.OW + ow0;
    yseg = p.dy + pix * p.lddy;
    zseg = p.src.x + pix * p.src.ld;
  };
  auto load_item = [&](int i) {
    if (i < NYJ) ry[i] = ld4so(yseg + i * ystep, yoff);
    else if (i < NIT) rz[i - NYJ] = ld4so(zseg, zoffk[i - NYJ]);
  };
  auto zpro = [&](int k) {
    float4 v = rz[k];
    if (zaff) { v.x = fmaf(za[k].x, v.x, zb[k].x); v.y = fmaf(za[k].y, v.y, zb[k].y); v.z = fmaf(za[k].z, v.z, zb[k].z); v.w = fmaf(za[k].w, v.w, zb[k].w); }
    if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    const bool ok = (zvalid >> k) & 1u;
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
  };
  auto prep2 = [&]() {                                               // NP = 2: masks and prologue in place, the wave's maxima to LDS (in front of the barrier)
    unsigned my = 0, mz = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      if (st_skip && yrow0 + k * YRS < st_skip) ry[k] = zero4();
      const unsigned b = absbits4(ry[k]); my = b > my ? b : my;
    }
#pragma unroll
    for (int k = 0; k < TP; ++k) { rz[k] = zpro(k); const unsigned b = absbits4(rz[k]); mz = b > mz ? b : mz; }
    wg_publish_max(wmx, 4, wave, lane, my, mz);
  };
  auto rescale2 = [&]() {
    const float r = wg_rescale(wmx, 4, fsc);
    if (r != 1.f) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] *= r;
    }
  };
  auto store_step = [&]() {
    const float sy = NP == 2 ? wg_pow2(fsc.kfy) : 1.f, sz = NP == 2 ? wg_pow2(fsc.kfz) : 1.f;
    if (NP != 2 && st_skip) {
#pragma unroll
      for (int k = 0; k < NYJ; ++k) if (yrow0 + k * YRS < st_skip) ry[k] = zero4();
    }
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      uint2 pl[NP];
      wg_split4<NP>(NP == 2 ? wg_mul4(ry[k], sy) : ry[k], pl);
      unsigned char* o = Yb + ((k & 1) ? ysw1 : ysw0) + k * (YRS << 5);
#pragma unroll
      for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * YPL) = pl[m];
    }
#pragma unroll
    for (int k = 0; k < TP; ++k) {
      const float4 v = NP == 2 ? wg_mul4(rz[k], sz) : zpro(k);
      uint2 pl[NP];
      wg_split4<NP>(v, pl);
      unsigned char* o = Zb + k * YIMG + zsw;
#pragma unroll
      for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(o + m * ZPL) = pl[m];
    }
  };
  auto rd = [&](const unsigned char* base, int plane_bytes, int pix0, wg_bf16x8* f) {
    const int o0 = wg_prow(pix0 + lrow) + 8 * tp, o1 = wg_prow(pix0 + lrow + 4) + 8 * tp;
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o0));
      const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o1));
      struct { wg_s16x4 a, b; } pr = {lo, hi};          // whole-register reinterpretation (see wgrad_h3b_kernel)
      f[m] = __builtin_bit_cast(wg_bf16x8, pr);
    }
  };
  auto mma = [&](f32x4 (&c)[NT][TP], int j, const wg_bf16x8 (&y)[NT][NP], const wg_bf16x8* z) {
#define WG_TERM(YI, ZI) _Pragma("unroll") for (int i = 0; i < NT; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[i][YI], z[ZI], c[i][j], 0, 0, 0);
#define WG_TERMH(YI, ZI) _Pragma("unroll") for (int i = 0; i < NT; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, y[i][YI]), __builtin_bit_cast(f16x8, z[ZI]), c[i][j], 0, 0, 0);
    if constexpr (NP == 2) { WG_TERMH(1, 0) WG_TERMH(0, 1) WG_TERMH(0, 0) } else {
    if (NP == 3) { WG_TERM(2, 0) WG_TERM(0, 2) WG_TERM(1, 1) }
    WG_TERM(1, 0) WG_TERM(0, 1) WG_TERM(0, 0) }
#undef WG_TERM
#undef WG_TERMH
  };
  if (sbeg < send) {
    load_prep(true);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_item(i);
    if (NP == 2) { prep2(); __syncthreads(); rescale2(); }
    store_step();
    __syncthreads();
    for (int seg = sbeg; seg < send; ++seg) {
      const bool more = seg + 1 < send;
      load_prep(more);
#pragma unroll
      for (int ks = 0; ks < H3_KP / 32; ++ks) {
        wg_bf16x8 yf[NT][NP];
#pragma unroll
        for (int i = 0; i < NT; ++i) rd(Yb + (wave * NT + i) * YIMG, YPL, ks * 32, yf[i]);
        wg_bf16x8 zf[2][NP];
        rd(Zb, ZPL, ks * 32, zf[0]);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
          load_item(2 * (ks * TP + j)); load_item(2 * (ks * TP + j) + 1);      // the next segment's loads, two per input tile (NIT <= 16)
          if (j + 1 < TP) rd(Zb + (j + 1) * YIMG, ZPL, ks * 32, zf[(j + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc, j, yf, zf[j & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (NP == 2 && more) prep2();
      __syncthreads();
      if (more) { if (NP == 2) rescale2(); store_step(); __syncthreads(); }
    }
  }
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * C;
  if (NP == 2) {
    const float iy = wg_pow2(254 - fsc.kfy), iz = wg_pow2(254 - fsc.kfz);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j) acc[i][j] = acc[i][j] * iy * iz;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cow = co0 + (wave * NT + i) * 16 + kq * 4 + r;
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const int c = c0 + 16 * j + li;
        if (cow < p.Cout && c < C) wsb[(long)cow * C + c] = acc[i][j][r];
      }
    }
}
constexpr size_t wg_h1b_lds(int np) { return (size_t)np * ((128 / 16) + H1_TP) * (H3_KP * 32 + 32) + 64; }      // + the waves' maxima (NP = 2)

inline bool wgrad_split_narrow() { return addk_env("ADDK_WGRAD_SPLIT_NARROW", 1) != 0; }
inline bool wgrad_split_enabled() { return addk_env("ADDK_WGRAD_SPLIT", 1) != 0; }
constexpr size_t wg_h3b_lds(int nt, int np, int ng) { return (size_t)np * ((64 * nt / 16) * (H3_KP * 32 + 32) + ng * (3 * 104 * 32 + (ng > 1 ? 64 : 0))) + 64; }      // + the waves' maxima (NP = 2)

// Halo-patch weight gradient of the cells' dense dilated convolutions (dil_conv_3x3 / dil_conv_5x5: 40/80/160 channels,
// dilation <= 2).  Same staging as wgrad_h3_kernel — per 64-pixel row segment dy [64][16*CT] and the KS activation rows
// [KS][64+(KS-1)d][16] go to LDS once and every tap reads its shifted window — but the accumulators are split the other
// way round: all four waves use every output-channel tile and each owns a QUARTER OF THE TAPS (7 of 25, 3 of 9), so a
// 40-channel conv keeps 3x7 = 21 accumulator tiles per wave with 83 % useful rows.  On the per-tap kernels these launches
// re-read dy and the activation once per tap (25x) and were bound by L2 bandwidth.
constexpr int HK_ZW = H3_KP + 4 * 2;       // widest patch row: 5x5, dilation 2

template <int KS, int CT, bool BATCH>
__global__ void __launch_bounds__(256, 2) wgrad_hk_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int TAPS = KS * KS, TPW = (TAPS + 3) / 4, HK = KS / 2;
  constexpr int BCO = 16 * CT, LY = BCO, YQ = BCO / 4;            // 48 and 80 are = 16 mod 32: conflict-free fragment reads
  constexpr int NYJ = (H3_KP * YQ + 255) / 256;
  constexpr int NZJ = (KS * HK_ZW * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ys[H3_KP * LY];
  __shared__ __attribute__((aligned(16))) float Zs[KS * HK_ZW * 16];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int zt = blk_x % p.nzt, yt = blk_x / p.nzt;
  const int co0 = yt * BCO, c0 = zt * 16;
  const int d = p.dil, ZW = H3_KP + (KS - 1) * d;
  const int spr = (p.OW + H3_KP - 1) / H3_KP;
  const int nseg = p.N * p.OH * spr;
  const int sbeg = blk_y * p.chunkP;
  int send = sbeg + p.chunkP; if (send > nseg) send = nseg;

  int yrow[NYJ], yqv[NYJ];
#pragma unroll
  for (int k = 0; k < NYJ; ++k) { const int slot = t + 256 * k; yrow[k] = slot / YQ; yqv[k] = slot - yrow[k] * YQ; }   // yrow >= 64: outside
  const int zq = t & 3, zc = c0 + 4 * zq, nremz = p.src.C - zc;
  int zr[NZJ], zj[NZJ];
#pragma unroll
  for (int k = 0; k < NZJ; ++k) {
    const int pix = (t + 256 * k) >> 2;
    zr[k] = pix / ZW; zj[k] = pix - zr[k] * ZW;         // zr >= KS marks a slot outside the patch
  }
  float4 za = make_float4(1.f, 1.f, 1.f, 1.f), zb = zero4();
  if (p.src.a && nremz > 0) { za = ld4(p.src.a + zc); zb = ld4(p.src.b + zc); }
  const bool zrelu = p.src.relu != 0;
  int zbase[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int tap = wave * TPW + j; if (tap > TAPS - 1) tap = TAPS - 1;      // surplus slots of the last wave recompute the last tap (discarded)
    zbase[j] = (((tap / KS) * HK_ZW) + kq + (tap % KS) * d) * 16 + li;
  }

  f32x4 acc[CT][TPW];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ry[NYJ], rz[NZJ];
  unsigned ymask = 0, zmask = 0;
  auto load_step = [&](int seg) {
    const int rowid = seg / spr, sx = seg - rowid * spr;
    const int n = rowid / p.OH, oh = rowid - n * p.OH;
    const int ow0 = sx * H3_KP;
    const long pp0 = (long)rowid * p.OW + ow0;
    ymask = 0; zmask = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      const int co = co0 + 4 * yqv[k];
      const bool ok = yrow[k] < H3_KP && co < p.Cout && ow0 + yrow[k] < p.OW;
      ry[k] = ld4(ok ? p.dy + (pp0 + yrow[k]) * p.lddy + co : p.dy);
      ymask |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      const int ih = oh + (zr[k] - HK) * d, iw = ow0 - HK * d + zj[k];
      const bool ok = zr[k] < KS && nremz > 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rz[k] = ld4(ok ? p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + zc : p.src.x);
      zmask |= (ok ? 1u : 0u) << k;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      float4 v = ry[k];
      const bool ok = (ymask >> k) & 1u;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (yrow[k] < H3_KP) lds_st4(&Ys[yrow[k] * LY + 4 * yqv[k]], v);
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      float4 v = rz[k];
      v.x = fmaf(za.x, v.x, zb.x); v.y = fmaf(za.y, v.y, zb.y); v.z = fmaf(za.z, v.z, zb.z); v.w = fmaf(za.w, v.w, zb.w);
      if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const bool ok = (zmask >> k) & 1u;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (zr[k] < KS) lds_st4(&Zs[(zr[k] * HK_ZW + zj[k]) * 16 + 4 * zq], v);
    }
  };

  if (sbeg < send) {
    load_step(sbeg);
    store_step();
    __syncthreads();
    const float* yw = &Ys[kq * LY + li];
    for (int seg = sbeg; seg < send; ++seg) {
      const bool more = seg + 1 < send;
      if (more) load_step(seg + 1);
      float yfA[CT], zfA[TPW], yfB[CT], zfB[TPW];
      auto rd = [&](int s4, float* yf, float* zf) {
#pragma unroll
        for (int i = 0; i < CT; ++i) yf[i] = yw[s4 * 4 * LY + i * 16];
#pragma unroll
        for (int j = 0; j < TPW; ++j) zf[j] = Zs[zbase[j] + s4 * 64];
      };
      auto mma = [&](const float* yf, const float* zf) {
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[i], zf[j], acc[i][j], 0, 0, 0);
      };
      rd(0, yfA, zfA);
#pragma unroll
      for (int s4 = 0; s4 < H3_KP / 4; s4 += 2) {
        rd(s4 + 1, yfB, zfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(yfA, zfA);
        __builtin_amdgcn_sched_barrier(0);
        if (s4 + 2 < H3_KP / 4) rd(s4 + 2, yfA, zfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(yfB, zfB);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (more) { store_step(); __syncthreads(); }
    }
  }
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * TAPS * C;
  const int c = c0 + li;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cow = co0 + i * 16 + kq * 4 + r;
      if (cow < p.Cout && c < C) {
        gfloat* o = wsb + (long)cow * TAPS * C + c;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int tap = wave * TPW + j;
          if (tap < TAPS) o[tap * C] = acc[i][j][r];
        }
      }
    }
}


// Split-bf16 form of wgrad_hk_kernel (same work decomposition, partial-tile layout and epilogue; arithmetic, LDS images and
// transposed fragment reads as in wgrad_h3b_kernel): every wave uses all CT output-channel tiles and owns a quarter of the taps.
constexpr int HKB_ZWP = 72;                 // patch row pitch in pixels: 64 + 4 * 2 (5x5, dilation 2), a multiple of 8
template <int KS, int CT, bool BATCH, int NP>
__global__ void __launch_bounds__(256, 2) wgrad_hkb_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  constexpr int TAPS = KS * KS, TPW = (TAPS + 3) / 4, HK = KS / 2;
  constexpr int BCO = 16 * CT, YQ = BCO / 4;
  constexpr int NYJ = (H3_KP * YQ + 255) / 256;
  constexpr int NZJ = (KS * HK_ZW * 4 + 255) / 256;
  constexpr int YIMG = H3_KP * 32 + 32, ZROW = HKB_ZWP * 32;       // bytes per dy tile image / per patch row (one plane)
  constexpr int YPL = CT * YIMG, ZPL = KS * ZROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  unsigned char* Yb = wsm;                                        // [NP][CT][64 px][16 co]
  unsigned char* Zb = wsm + NP * YPL;                             // [NP][KS rows][HKB_ZWP px][16 ci]
  unsigned* wmx = reinterpret_cast<unsigned*>(wsm + NP * (YPL + ZPL));      // NP = 2: [2][4] the waves' largest magnitudes of the segment being staged (wgrad_h3b_kernel)
  WgScale fsc = {0, 0};

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int zt = blk_x % p.nzt, yt = blk_x / p.nzt;
  const int co0 = yt * BCO, c0 = zt * 16;
  const int d = p.dil, ZW = H3_KP + (KS - 1) * d;
  const int spr = (p.OW + H3_KP - 1) / H3_KP;
  const int nseg = p.N * p.OH * spr;
  const int sbeg = blk_y * p.chunkP;
  int send = sbeg + p.chunkP; if (send > nseg) send = nseg;

  int yrow[NYJ], yqv[NYJ];
#pragma unroll
  for (int k = 0; k < NYJ; ++k) { const int slot = t + 256 * k; yrow[k] = slot / YQ; yqv[k] = slot - yrow[k] * YQ; }   // yrow >= 64: outside
  const int zq = t & 3, zc = c0 + 4 * zq, nremz = p.src.C - zc;
  int zr[NZJ], zj[NZJ];
#pragma unroll
  for (int k = 0; k < NZJ; ++k) {
    const int pix = (t + 256 * k) >> 2;
    zr[k] = pix / ZW; zj[k] = pix - zr[k] * ZW;         // zr >= KS marks a slot outside the patch
  }
  float4 za = make_float4(1.f, 1.f, 1.f, 1.f), zb = zero4();
  if (p.src.a && nremz > 0) { za = ld4(p.src.a + zc); zb = ld4(p.src.b + zc); }
  const bool zrelu = p.src.relu != 0;
  const int tq = li >> 2, tp = li & 3;
  const int lrow = 8 * kq + tq;
  int zrow_off[TPW], zshift[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int tap = wave * TPW + j; if (tap > TAPS - 1) tap = TAPS - 1;      // surplus slots of the last wave recompute the last tap (discarded)
    zrow_off[j] = (tap / KS) * ZROW; zshift[j] = (tap % KS) * d;
  }

  f32x4 acc[CT][TPW];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ry[NYJ], rz[NZJ];
  unsigned ymask = 0, zmask = 0;
  auto load_step = [&](int seg) {
    const int rowid = seg / spr, sx = seg - rowid * spr;
    const int n = rowid / p.OH, oh = rowid - n * p.OH;
    const int ow0 = sx * H3_KP;
    const long pp0 = (long)rowid * p.OW + ow0;
    ymask = 0; zmask = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      const int co = co0 + 4 * yqv[k];
      const bool ok = yrow[k] < H3_KP && co < p.Cout && ow0 + yrow[k] < p.OW;
      ry[k] = ld4(ok ? p.dy + (pp0 + yrow[k]) * p.lddy + co : p.dy);
      ymask |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      const int ih = oh + (zr[k] - HK) * d, iw = ow0 - HK * d + zj[k];
      const bool ok = zr[k] < KS && nremz > 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rz[k] = ld4(ok ? p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + zc : p.src.x);
      zmask |= (ok ? 1u : 0u) << k;
    }
  };
  auto ypro = [&](int k) {
    float4 v = ry[k];
    const bool ok = (ymask >> k) & 1u;
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
  };
  auto zpro = [&](int k) {
    float4 v = rz[k];
    v.x = fmaf(za.x, v.x, zb.x); v.y = fmaf(za.y, v.y, zb.y); v.z = fmaf(za.z, v.z, zb.z); v.w = fmaf(za.w, v.w, zb.w);
    if (zrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    const bool ok = (zmask >> k) & 1u;
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
  };
  auto prep2 = [&]() {                                               // NP = 2: masks and prologue in place, the wave's maxima to LDS (in front of the barrier)
    unsigned my = 0, mz = 0;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) { ry[k] = ypro(k); const unsigned b = absbits4(ry[k]); my = b > my ? b : my; }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) { rz[k] = zpro(k); const unsigned b = absbits4(rz[k]); mz = b > mz ? b : mz; }
    wg_publish_max(wmx, 4, wave, lane, my, mz);
  };
  auto rescale2 = [&]() {
    const float r = wg_rescale(wmx, 4, fsc);
    if (r != 1.f) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[i][j] *= r;
    }
  };
  auto store_step = [&]() {
    const float sy = NP == 2 ? wg_pow2(fsc.kfy) : 1.f, sz = NP == 2 ? wg_pow2(fsc.kfz) : 1.f;
#pragma unroll
    for (int k = 0; k < NYJ; ++k) {
      const float4 v = NP == 2 ? wg_mul4(ry[k], sy) : ypro(k);
      if (yrow[k] < H3_KP) {
        uint2 pl[NP];
        wg_split4<NP>(v, pl);
        const int off = (yqv[k] >> 2) * YIMG + wg_prow(yrow[k]) + 8 * (yqv[k] & 3);
#pragma unroll
        for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(Yb + m * YPL + off) = pl[m];
      }
    }
#pragma unroll
    for (int k = 0; k < NZJ; ++k) {
      const float4 v = NP == 2 ? wg_mul4(rz[k], sz) : zpro(k);
      if (zr[k] < KS) {
        uint2 pl[NP];
        wg_split4<NP>(v, pl);
        const int off = zr[k] * ZROW + wg_prow(zj[k]) + 8 * zq;
#pragma unroll
        for (int m = 0; m < NP; ++m) *reinterpret_cast<uint2*>(Zb + m * ZPL + off) = pl[m];
      }
    }
  };
  auto rd = [&](const unsigned char* base, int plane_bytes, int pix0, wg_bf16x8* f) {
    const int o0 = wg_prow(pix0 + lrow) + 8 * tp, o1 = wg_prow(pix0 + lrow + 4) + 8 * tp;
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o0));
      const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4*)(base + m * plane_bytes + o1));
      struct { wg_s16x4 a, b; } pr = {lo, hi};          // whole-register reinterpretation (see wgrad_h3b_kernel)
      f[m] = __builtin_bit_cast(wg_bf16x8, pr);
    }
  };
  auto mma = [&](f32x4 (&c)[CT][TPW], int j, const wg_bf16x8 (&y)[CT][NP], const wg_bf16x8* z) {
#define WG_TERM(YI, ZI) _Pragma("unroll") for (int i = 0; i < CT; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[i][YI], z[ZI], c[i][j], 0, 0, 0);
#define WG_TERMH(YI, ZI) _Pragma("unroll") for (int i = 0; i < CT; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, y[i][YI]), __builtin_bit_cast(f16x8, z[ZI]), c[i][j], 0, 0, 0);
    if constexpr (NP == 2) { WG_TERMH(1, 0) WG_TERMH(0, 1) WG_TERMH(0, 0) } else {
    if (NP == 3) { WG_TERM(2, 0) WG_TERM(0, 2) WG_TERM(1, 1) }
    WG_TERM(1, 0) WG_TERM(0, 1) WG_TERM(0, 0) }
#undef WG_TERM
#undef WG_TERMH
  };

  if (sbeg < send) {
    load_step(sbeg);
    if (NP == 2) { prep2(); __syncthreads(); rescale2(); }
    store_step();
    __syncthreads();
    for (int seg = sbeg; seg < send; ++seg) {
      const bool more = seg + 1 < send;
      if (more) load_step(seg + 1);
#pragma unroll
      for (int ks = 0; ks < H3_KP / 32; ++ks) {
        wg_bf16x8 yf[CT][NP];
#pragma unroll
        for (int i = 0; i < CT; ++i) rd(Yb + i * YIMG, YPL, ks * 32, yf[i]);
        wg_bf16x8 zf[2][NP];
        rd(Zb + zrow_off[0], ZPL, ks * 32 + zshift[0], zf[0]);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          if (j + 1 < TPW) rd(Zb + zrow_off[j + 1], ZPL, ks * 32 + zshift[j + 1], zf[(j + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc, j, yf, zf[j & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (NP == 2 && more) prep2();
      __syncthreads();
      if (more) { if (NP == 2) rescale2(); store_step(); __syncthreads(); }
    }
  }
  const int C = p.src.C;
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * TAPS * C;
  if (NP == 2) {
    const float iy = wg_pow2(254 - fsc.kfy), iz = wg_pow2(254 - fsc.kfz);
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[i][j] = acc[i][j] * iy * iz;
  }
  const int c = c0 + li;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cow = co0 + i * 16 + kq * 4 + r;
      if (cow < p.Cout && c < C) {
        gfloat* o = wsb + (long)cow * TAPS * C + c;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int tap = wave * TPW + j;
          if (tap < TAPS) o[tap * C] = acc[i][j][r];
        }
      }
    }
}
constexpr size_t wg_hkb_lds(int ks, int ct, int np) { return (size_t)np * ((size_t)ct * (H3_KP * 32 + 32) + (size_t)ks * HKB_ZWP * 32) + 64; }      // + the waves' maxima (NP = 2)

// Register-streaming weight gradient for the narrow cell convolutions (Cout, C <= 160; 1x1, dilated k x k, strided):
// no LDS staging and no barrier in the main loop.  A wave walks its own pixel range four pixels per MFMA k-step; lane
// (li, kq) loads its slice of dy (pixel kq) and of the activation (tap-shifted pixel kq) straight from global memory and
// uses the COMPONENTS of those vector loads as MFMA operands.  Two lane layouts per operand:
//   LAY 4: one float4 at channel 4*li        -> 4 operand tiles, tile e holds channels {4r+e}           (<= 64 channels)
//   LAY 3: one float2 at channel 2*li + one float at channel 32+li -> 3 tiles {2r}, {2r+1}, {32+r}      (<= 48 channels)
// so a 40-channel conv issues 3x3 MFMAs per k-step (83 % useful rows) instead of 4x4 (62 %).  Every load is unconditional
// (masked lanes read a safe address and are zeroed) and RS_U k-steps are in flight per wave.  The LDS-staged kernels
// above ran these launches at 23 TF/s, bound by their barrier/latency chains.
constexpr int RS_T = 64;             // LDS tile edge for the cross-wave combine (channels)
constexpr int RS_U = 4;              // k-steps per unrolled batch

template <int LAY> struct RsFrag { float v[LAY]; };

template <int LAY>
__device__ __forceinline__ RsFrag<LAY> rs_load(const float* base, int li, bool ok4, bool ok2, bool ok1) {
  RsFrag<LAY> f;
  if (LAY == 4) {
    const float4 x = ld4(base + (ok4 ? 4 * li : 0));
    f.v[0] = ok4 ? x.x : 0.f; f.v[1] = ok4 ? x.y : 0.f; f.v[2] = ok4 ? x.z : 0.f; f.v[3 % LAY] = ok4 ? x.w : 0.f;
  } else {
    typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
    const rs_f32x2 x = *(const __attribute__((address_space(1))) rs_f32x2*)(base + (ok2 ? 2 * li : 0));
    const float y = ((const gfloat*)base)[ok1 ? 32 + li : 0];
    f.v[0] = ok2 ? x.x : 0.f; f.v[1] = ok2 ? x.y : 0.f; f.v[2] = ok1 ? y : 0.f;
  }
  return f;
}
// channel (relative to the tile origin) held by component e, row/column index R of the MFMA tile
template <int LAY> __device__ __forceinline__ int rs_chan(int e, int R) { return LAY == 4 ? 4 * R + e : (e < 2 ? 2 * R + e : 32 + R); }

// [r5] F16 = the split-fp16 arithmetic (common.h) in the same register-streaming form: a batch of RS_U = 4 k-steps (16 pixels per wave) becomes ONE k-step of
// v_mfma_f32_16x16x16_f16 — the lane that loaded pixel 4 u + kq in k-step u supplies it as k-slot 4 kq + u of both operands (any bijection of the contraction
// index serves as long as the two operands agree) — so the loads are exactly the fp32 form's and 36 fp32 matrix instructions (1152 pipe cycles per 16 pixels)
// become 27 fp16 ones (432).  Every WAVE keeps its own running scales for dy and for the activation (it owns its accumulators until the fixed-order combine at
// the end): per batch the wave's largest magnitudes by four DPP steps and readlanes, the accumulators rescaled when a scale drops, unscaled before the combine.
typedef _Float16 wg_f16x4 __attribute__((ext_vector_type(4)));
template <int LA, int LB, bool BATCH, bool F16 = false>
__global__ void __launch_bounds__(256, 2) wgrad_rs_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  static_assert(!F16 || RS_U == 4, "one fp16 k-step = four pixel quads");
  int op = 0, blk_x = blockIdx.x, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_x = __builtin_amdgcn_readfirstlane(wk.y); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);          // by value: the descriptor lives in scalar registers
  __shared__ float tile[RS_T][RS_T + 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  int bx = blk_x;
  const int zt = bx % p.nzt; bx /= p.nzt;
  const int tap = bx % p.taps; const int yt = bx / p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int tsy = p.vecY, tsz = p.vecZ;        // tile strides in channels (multiples of 4), set by the host for this kind
  const int co0 = yt * tsy, c0 = zt * tsz;
  const int ncy = min(tsy, p.Cout - co0), ncz = min(tsz, p.src.C - c0);
  const bool y4 = 4 * li < ncy, y2 = 2 * li < min(ncy, 32), y1 = 32 + li < ncy;
  const bool z4 = 4 * li < ncz, z2 = 2 * li < min(ncz, 32), z1 = 32 + li < ncz;
  const int C = p.src.C;
  const int pbeg = blk_y * p.chunkP;
  int pend = pbeg + p.chunkP; if (pend > p.P) pend = p.P;
  const int span = (pend - pbeg + 3) / 4;                  // k-steps of the block
  const int per_wave = (span + 3) / 4;
  const int s_beg = wave * per_wave, s_end = min(span, s_beg + per_wave);
  RsFrag<LB> za, zb;
#pragma unroll
  for (int f = 0; f < LB; ++f) { za.v[f] = 1.f; zb.v[f] = 0.f; }
  if (p.src.a) { za = rs_load<LB>(p.src.a + c0, li, z4, z2, z1); zb = rs_load<LB>(p.src.b + c0, li, z4, z2, z1); }
  const bool zrelu = p.src.relu != 0;
  const int ohw = p.OH * p.OW;
  const float* ybase = p.dy + co0;
  const float* zbase = p.src.x + c0;
  const bool same = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;      // 1x1: the activation pixel is the dy pixel
  f32x4 acc[LA][LB];
#pragma unroll
  for (int e = 0; e < LA; ++e)
#pragma unroll
    for (int f = 0; f < LB; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  {
    // [r4] The loop used to spend 6-11 vector instructions per MFMA (r03 counters: 6892 VALU / 1148 MFMA per wave, 66-70 % issue stall; 52 of
    // the 84 VALU of the compute block were v_cndmask, the load blocks re-derived (n, oh, ow) by integer division for every k-step): with
    // 24 issue cycles free per 32-cycle fp32 MFMA the kernel was bound by VALU issue, not by HBM.  Now
    //   * this lane's pixel walks incrementally (4 pixels per k-step: offsets by addition, (n, oh, ow) by carries, no division);
    //   * channel-validity selects are gone: an operand element of a channel beyond the tile only feeds accumulator rows / columns that
    //     are never stored (the combine below masks them), and its load address is clamped as before;
    //   * pixel validity (tail of the range, zero padding) is applied to ONE operand only, the activation: 0 * dy adds nothing, and dy of a
    //     masked pixel is read from a valid address.
    // Same products, same summation order: results are bit-identical to the previous form for finite gradients.
    int pp = pbeg + 4 * s_beg + kq;
    int n_ = 0, oh_ = 0, ow_ = 0;
    if (!same) { n_ = pp / ohw; const int rem = pp - n_ * ohw; oh_ = rem / p.OW; ow_ = rem - oh_ * p.OW; }
    const int ylane = LA == 4 ? (y4 ? 4 * li : 0) : (y2 ? 2 * li : 0), ylane1 = (LA == 3 && y1) ? 32 + li : 0;
    const int zlane = LB == 4 ? (z4 ? 4 * li : 0) : (z2 ? 2 * li : 0), zlane1 = (LB == 3 && z1) ? 32 + li : 0;
    auto ldA = [&](const float* b) {
      RsFrag<LA> f;
      if (LA == 4) { const float4 x = ld4(b + ylane); f.v[0] = x.x; f.v[1] = x.y; f.v[2] = x.z; f.v[3 % LA] = x.w; }
      else {
        typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
        const rs_f32x2 x = *(const __attribute__((address_space(1))) rs_f32x2*)(b + ylane);
        f.v[0] = x.x; f.v[1] = x.y; f.v[2] = ((const gfloat*)b)[ylane1];
      }
      return f;
    };
    auto ldB = [&](const float* b) {
      RsFrag<LB> f;
      if (LB == 4) { const float4 x = ld4(b + zlane); f.v[0] = x.x; f.v[1] = x.y; f.v[2] = x.z; f.v[3 % LB] = x.w; }
      else {
        typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
        const rs_f32x2 x = *(const __attribute__((address_space(1))) rs_f32x2*)(b + zlane);
        f.v[0] = x.x; f.v[1] = x.y; f.v[2] = ((const gfloat*)b)[zlane1];
      }
      return f;
    };
    WgScale fsc = {0, 0};
    for (int s0 = s_beg; s0 < s_end; s0 += RS_U) {
      RsFrag<LA> dy4[RS_U]; RsFrag<LB> z4v[RS_U];
      bool zv[RS_U];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const bool pv_ = (s0 + u) < s_end && pp < pend;
        dy4[u] = ldA(ybase + (pv_ ? (long)pp * p.lddy : 0));
        long zoff = 0; bool okz = pv_;
        if (same) zoff = (long)pp * p.src.ld;
        else {
          const int ih = oh_ * p.stride - p.pad + kh * p.dil, iw = ow_ * p.stride - p.pad + kw * p.dil;
          okz = okz && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          zoff = ((long)(n_ * p.H + ih) * p.W + iw) * p.src.ld;
          ow_ += 4;                                          // the lane's next pixel: 4 further along the flattened (n, oh, ow) order
          while (ow_ >= p.OW) { ow_ -= p.OW; if (++oh_ >= p.OH) { oh_ = 0; ++n_; } }
        }
        z4v[u] = ldB(zbase + (okz ? zoff : 0));
        zv[u] = okz;
        pp += 4;
      }
      if constexpr (F16) {
        float my = 0.f, mz = 0.f;
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
          const bool ok = zv[u];
#pragma unroll
          for (int f = 0; f < LB; ++f) {
            float x = fmaf(za.v[f], z4v[u].v[f], zb.v[f]);
            if (zrelu) x = fmaxf(x, 0.f);
            x = ok ? x : 0.f;
            z4v[u].v[f] = x;
            mz = fmaxf(mz, fabsf(x));
          }
#pragma unroll
          for (int e = 0; e < LA; ++e) my = fmaxf(my, fabsf(dy4[u].v[e]));
        }
        const int wy = f16_scale_field(wave_umax(__float_as_uint(my))), wz = f16_scale_field(wave_umax(__float_as_uint(mz)));
        int sh = 0;
        if (fsc.kfy == 0) fsc.kfy = wy; else if (wy < fsc.kfy) { sh += wy - fsc.kfy; fsc.kfy = wy; }
        if (fsc.kfz == 0) fsc.kfz = wz; else if (wz < fsc.kfz) { sh += wz - fsc.kfz; fsc.kfz = wz; }
        if (sh != 0) {                          // (wave-uniform) a larger batch: the sums move to the coarser scale, exactly
          const int rf = 127 + sh;
          const float r = rf > 0 ? wg_pow2(rf) : 0.f;
#pragma unroll
          for (int e = 0; e < LA; ++e)
#pragma unroll
            for (int f = 0; f < LB; ++f) acc[e][f] *= r;
        }
        const float sy = wg_pow2(fsc.kfy), sz = wg_pow2(fsc.kfz);
        wg_f16x4 yh[LA], yl[LA], zh[LB], zl[LB];
#pragma unroll
        for (int e = 0; e < LA; ++e) {
          uint2 pl[2];
          split4h(make_float4(dy4[0].v[e] * sy, dy4[1].v[e] * sy, dy4[2].v[e] * sy, dy4[3].v[e] * sy), pl);
          yh[e] = __builtin_bit_cast(wg_f16x4, pl[0]); yl[e] = __builtin_bit_cast(wg_f16x4, pl[1]);
        }
#pragma unroll
        for (int f = 0; f < LB; ++f) {
          uint2 pl[2];
          split4h(make_float4(z4v[0].v[f] * sz, z4v[1].v[f] * sz, z4v[2].v[f] * sz, z4v[3].v[f] * sz), pl);
          zh[f] = __builtin_bit_cast(wg_f16x4, pl[0]); zl[f] = __builtin_bit_cast(wg_f16x4, pl[1]);
        }
#pragma unroll
        for (int e = 0; e < LA; ++e)
#pragma unroll
          for (int f = 0; f < LB; ++f) {
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x16f16(yl[e], zh[f], acc[e][f], 0, 0, 0);
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x16f16(yh[e], zl[f], acc[e][f], 0, 0, 0);
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x16f16(yh[e], zh[f], acc[e][f], 0, 0, 0);
          }
      } else {
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        RsFrag<LB> v = z4v[u];
        const bool ok = zv[u];
#pragma unroll
        for (int f = 0; f < LB; ++f) {
          float x = fmaf(za.v[f], v.v[f], zb.v[f]);
          if (zrelu) x = fmaxf(x, 0.f);
          v.v[f] = ok ? x : 0.f;
        }
#pragma unroll
        for (int e = 0; e < LA; ++e)
#pragma unroll
          for (int f = 0; f < LB; ++f)
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy4[u].v[e], v.v[f], acc[e][f], 0, 0, 0);
      }
      }
    }
    if (F16) {                                  // this wave's two scales leave its sums
      const float iy = wg_pow2(254 - fsc.kfy), iz = wg_pow2(254 - fsc.kfz);
#pragma unroll
      for (int e = 0; e < LA; ++e)
#pragma unroll
        for (int f = 0; f < LB; ++f) acc[e][f] = acc[e][f] * iy * iz;
    }
  }
  // combine the four waves in a fixed order; acc[e][f][r] = dW[co0 + chanA(e, 4*kq + r)][c0 + chanB(f, li)]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int e = 0; e < LA; ++e)
#pragma unroll
        for (int f = 0; f < LB; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &tile[rs_chan<LA>(e, 4 * kq + r)][rs_chan<LB>(f, li)];
            *d = (w == 0) ? acc[e][f][r] : *d + acc[e][f][r];
          }
    }
    __syncthreads();
  }
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * p.taps * C;
  for (int idx = t; idx < RS_T * RS_T; idx += 256) {
    const int r = idx / RS_T, cc = idx - r * RS_T;
    if (r < ncy && cc < ncz) wsb[((long)(co0 + r) * p.taps + tap) * C + c0 + cc] = tile[r][cc];
  }
}

// Weight gradient of a k x k conv with a HANDFUL of input channels (stem0: 3 -> 64, 3x3, stride 2; ADD.py:153-157): KH*KW*C <= 32 patch
// values per output pixel.  On the generic kernel this launch re-read dy once per tap (0.37 ms, 0.9 TB/s).  Here it is ONE pass in the
// register-streaming form of wgrad_rs_kernel: lane (li, kq) of a k-step = 4 pixels loads dy of pixel kq as one float4 at channel 4 li
// (component e = A operand of output-channel tile {4r + e}) and GATHERS patch values e0 = li and 16 + li of that pixel (tap e / C,
// channel e % C, prologue and zero padding applied) as the B operands of the two column tiles: 8 MFMAs per 4 pixels, dy read once.
template <bool BATCH>
__global__ void __launch_bounds__(256, 2) wgrad_st_kernel(const WgK pv, const WgK* __restrict__ ops, const int4* __restrict__ work) {
  int op = 0, blk_y = blockIdx.y;
  if (BATCH) {
    const int4 wk = work[blockIdx.x];
    op = __builtin_amdgcn_readfirstlane(wk.x); blk_y = __builtin_amdgcn_readfirstlane(wk.z);
  }
  const WgK p = wg_desc<BATCH>(pv, ops, op);
  __shared__ float tile[RS_T][33];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int C = p.src.C, NE = p.taps * C;                   // patch values per pixel (<= 32)
  const bool y4 = 4 * li < p.Cout;
  const int pbeg = blk_y * p.chunkP;
  int pend = pbeg + p.chunkP; if (pend > p.P) pend = p.P;
  const int span = (pend - pbeg + 3) / 4, per_wave = (span + 3) / 4;
  const int s_beg = wave * per_wave, s_end = min(span, s_beg + per_wave);
  // this lane's two patch elements: tap and channel, prologue coefficients
  int ekh[2], ekw[2], ec[2]; bool eok[2]; float ea[2], eb[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int e = 16 * f + li;
    eok[f] = e < NE;
    const int tap = eok[f] ? e / C : 0;
    ec[f] = eok[f] ? e - tap * C : 0;
    ekh[f] = (tap / p.KW) * p.dil - p.pad; ekw[f] = (tap % p.KW) * p.dil - p.pad;
    ea[f] = 1.f; eb[f] = 0.f;
    if (p.src.a && eok[f]) { ea[f] = ((const gfloat*)p.src.a)[ec[f]]; eb[f] = ((const gfloat*)p.src.b)[ec[f]]; }
  }
  const bool zrelu = p.src.relu != 0;
  const int ohw = p.OH * p.OW;
  f32x4 acc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s0 = s_beg; s0 < s_end; s0 += RS_U) {
    RsFrag<4> dy4[RS_U]; float zv[RS_U][2];
#pragma unroll
    for (int u = 0; u < RS_U; ++u) {
      const int pp = pbeg + 4 * (s0 + u) + kq;
      const bool pv_ = (s0 + u) < s_end && pp < pend;
      dy4[u] = rs_load<4>(p.dy + (pv_ ? (long)pp * p.lddy : 0), li, pv_ && y4, false, false);
      const int n = pp / ohw, rem = pp - n * ohw, oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int ih = oh * p.stride + ekh[f], iw = ow * p.stride + ekw[f];
        const bool ok = pv_ && eok[f] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        float x = ((const gfloat*)p.src.x)[ok ? ((long)(n * p.H + ih) * p.W + iw) * p.src.ld + ec[f] : 0];
        x = fmaf(ea[f], x, eb[f]);
        if (zrelu) x = fmaxf(x, 0.f);
        zv[u][f] = ok ? x : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RS_U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy4[u].v[e], zv[u][f], acc[e][f], 0, 0, 0);
  }
  // combine the four waves in a fixed order; acc[e][f][r] = dW[4 (4 kq + r) + e][16 f + li]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &tile[4 * (4 * kq + r) + e][16 * f + li];
            *d = (w == 0) ? acc[e][f][r] : *d + acc[e][f][r];
          }
    }
    __syncthreads();
  }
  gfloat* wsb = (gfloat*)p.ws + (long)blk_y * p.Cout * NE;
  for (int idx = t; idx < RS_T * 32; idx += 256) {
    const int r = idx >> 5, cc = idx & 31;
    if (r < p.Cout && cc < NE) wsb[(long)r * NE + cc] = tile[r][cc];
  }
}

__global__ void wgrad_reduce_kernel(const float* ws, int splits, int Cout, int taps, int C, float* dw, int ldw,
                                    int cin_total, int w_choff, int accumulate) {
  long n = (long)Cout * taps * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent chains keep the loads in flight
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      s0 += ws[(long)k * n + i]; s1 += ws[(long)(k + 1) * n + i]; s2 += ws[(long)(k + 2) * n + i]; s3 += ws[(long)(k + 3) * n + i];
    }
    for (; k < splits; ++k) s0 += ws[(long)k * n + i];
    float s = (s0 + s1) + (s2 + s3);
    int c = (int)(i % C); long r = i / C; int tap = (int)(r % taps); int co = (int)(r / taps);
    float* d = dw + (long)co * ldw + (long)tap * cin_total + w_choff + c;
    *d = accumulate ? *d + s : s;
  }
}

// many slices, few elements: one wave per element, lanes stride over the slices, fixed-order butterfly
__global__ void __launch_bounds__(256) wgrad_reduce_wave_kernel(const float* ws, int splits, int Cout, int taps, int C, float* dw, int ldw,
                                                                int cin_total, int w_choff, int accumulate) {
  const long n = (long)Cout * taps * C;
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  float s = 0.f;
  for (int k = lane; k < splits; k += 64) s += ws[(long)k * n + i];
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) {
    int c = (int)(i % C); long r = i / C; int tap = (int)(r % taps); int co = (int)(r / taps);
    float* d = dw + (long)co * ldw + (long)tap * cin_total + w_choff + c;
    *d = accumulate ? *d + s : s;
  }
}

// batched reduce: rwork[b] = (op, first element, mode): mode 1 = many slices, few elements: 64 consecutive elements x 4 slice
// groups per block (every load a coalesced 256-byte segment; fixed-order combine in LDS), 0 = thread per element
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(const WgK* __restrict__ ops, const int4* __restrict__ rwork) {
  __shared__ float part[4][64];
  const int4 wk = rwork[blockIdx.x];
  const WgK p = wg_desc<true>(ops[0], ops, __builtin_amdgcn_readfirstlane(wk.x));
  const int C = p.src.C;
  const gfloat* ws = (const gfloat*)p.ws;
  const long n = (long)p.Cout * p.taps * C;
  long i; float s;
  bool writer;
  if (wk.z) {
    const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
    i = (long)wk.y + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int k = rg;
      for (; k + 12 < p.splits; k += 16) {
        s0 += ws[(long)k * n + i]; s1 += ws[(long)(k + 4) * n + i]; s2 += ws[(long)(k + 8) * n + i]; s3 += ws[(long)(k + 12) * n + i];
      }
      for (; k < p.splits; k += 4) s0 += ws[(long)k * n + i];
    }
    part[rg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    writer = rg == 0 && i < n;
  } else {
    i = (long)wk.y + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < p.splits; k += 4) {
      s0 += ws[(long)k * n + i]; s1 += ws[(long)(k + 1) * n + i]; s2 += ws[(long)(k + 2) * n + i]; s3 += ws[(long)(k + 3) * n + i];
    }
    for (; k < p.splits; ++k) s0 += ws[(long)k * n + i];
    s = (s0 + s1) + (s2 + s3);
    writer = true;
  }
  if (writer) {
    int c = (int)(i % C); long r = i / C; int tap = (int)(r % p.taps); int co = (int)(r / p.taps);
    gfloat* d = (gfloat*)p.dw + (long)co * p.ldw + (long)tap * p.cin_total + p.w_choff + c;
    *d = p.accumulate ? *d + s : s;
  }
}

int pick_cty(int Cout) {
  const int cands[5] = {8, 5, 4, 3, 2};
  int best = 2; long bc = -1;
  for (int k = 0; k < 5; ++k) { long cols = (long)cdiv(Cout, 16 * cands[k]) * 16 * cands[k]; if (bc < 0 || cols < bc) { bc = cols; best = cands[k]; } }
  return best;
}
int pick_ctz(int C) {
  const int cands[4] = {5, 4, 3, 1};
  int best = 1; long bc = -1;
  for (int k = 0; k < 4; ++k) { long cols = (long)cdiv(C, 16 * cands[k]) * 16 * cands[k]; if (bc < 0 || cols < bc) { bc = cols; best = cands[k]; } }
  return best;
}
// 0: pixel-split kernel; 1: output-split 128x64 (wide heads); 2: output-split 96x96 (80-channel cells);
// 3: output-split 64x64 (64-channel stem); 4: output-split 128x128 (wide heads with >= 128 input channels)
int os_kind(int Cout, int C) {
  if (Cout >= 128 && Cout % 128 == 0 && C >= 48) return 1;
  if (Cout > 64 && Cout <= 96 && C > 64 && C <= 96) return 2;
  if (Cout > 48 && Cout <= 64 && C > 48 && C <= 64) return 3;
  return 0;
}
void pick_tiles(int Cout, int C, int* cty, int* ctz) {
  *cty = pick_cty(Cout); *ctz = pick_ctz(C);
  if (*cty == 8 && *ctz == 5) *ctz = 4;   // 8x5 accumulator tiles would not leave room for the staging registers
  switch (os_kind(Cout, C)) {
    case 1: *cty = 8; *ctz = 4; break;
    case 4: *cty = 8; *ctz = 8; break;
    case 2: *cty = 6; *ctz = 6; break;
    case 3: *cty = 4; *ctz = 4; break;
    default: break;
  }
}
bool use_output_split(int Cout, int C) { return os_kind(Cout, C) != 0; }
// 5: halo-patch kernel (3x3, stride 1, 'same' padding, wide): tiles are (64*NT co) x (16 c), the nine taps live in the block
bool h3_ok(const addk_conv_wgrad_args* a) {
  return (addk_get_fast_paths() & ADDK_FAST_WGRAD3) && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == a->dil && a->dil >= 1 && a->dil <= 18 &&
         a->OH == a->H && a->OW == a->W && a->W >= 64 && a->Cout % 64 == 0 && a->src.C >= 16 &&
         aligned16(a->dy) && a->lddy % 4 == 0 && src_vec_ok(a->src) && (long)a->N * a->H * a->W >= 8192;
}
// 9: the split-bf16 halo-patch arithmetic for the wide 1x1 heads (wgrad_h1b_kernel), when the split kernels are in use
inline bool h3b_runs(int cty);
bool h1_ok(const addk_conv_wgrad_args* a) {
  return (addk_get_fast_paths() & ADDK_FAST_WGRAD3) && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0 && a->OH == a->H && a->OW == a->W &&
         a->W >= 64 && a->Cout % 128 == 0 && a->src.C >= 64 && a->src.C % 4 == 0 && aligned16(a->dy) && a->lddy % 4 == 0 && src_vec_ok(a->src) &&
         (!a->src.a || (aligned16(a->src.a) && aligned16(a->src.b))) && (long)a->N * a->H * a->W >= 8192 && h3b_runs(8);
}
// 6: register-streaming kernel for the narrow cell convolutions
bool rs_ok(const addk_conv_wgrad_args* a) {
  return (addk_get_fast_paths() & ADDK_FAST_WGRAD_RS) && a->Cout <= 160 && a->Cout >= 16 && a->src.C >= 16 && a->KH * a->KW <= 25 &&
         aligned16(a->dy) && a->lddy % 4 == 0 && a->Cout % 4 == 0 && src_vec_ok(a->src) && (long)a->N * a->OH * a->OW >= 4096;
}
// tile stride of an operand: 40-channel tiles (3-component layout) when the channel count is a multiple of 40 or fits 48,
// else up to 64 channels (4-component layout)
inline int rs_tile(int Cn) {
  if (Cn <= 48) return Cn;
  if (Cn % 40 == 0) return 40;
  if (Cn % 48 == 0) return 48;
  const int nt = cdiv(Cn, RS_T); return cdiv(cdiv(Cn, nt), 4) * 4;
}
inline int rs_lay(int tile) { return tile <= 48 ? 3 : 4; }
// 7: halo-patch kernel with the taps split across waves (the cells' dilated 3x3 / 5x5 convolutions)
// output-channel tiles per block: 3 for the 5x5 (7 taps per wave -> 21 accumulator tiles), up to 5 for the 3x3 (3 taps per wave)
inline int hk_ct(int Cout, int ks) { return (ks == 5 || Cout <= 48) ? 3 : 5; }
inline int hk_tiles(int Cout, int C, int ks) { return cdiv(Cout, 16 * hk_ct(Cout, ks)) * cdiv(C, 16); }
bool hk_ok(const addk_conv_wgrad_args* a) {
  return (addk_get_fast_paths() & ADDK_FAST_WGRAD_RS) && a->KH == a->KW && (a->KH == 3 || a->KH == 5) && a->stride == 1 &&
         a->dil >= 1 && a->dil <= 2 && a->pad == a->dil * (a->KH / 2) && a->OH == a->H && a->OW == a->W &&
         a->Cout >= 32 && a->Cout <= 160 && a->Cout % 4 == 0 && a->src.C >= 16 && a->OW >= 32 &&
         aligned16(a->dy) && a->lddy % 4 == 0 && src_vec_ok(a->src) && (long)a->N * a->H * a->W >= 4096;
}
// 8: few input channels (stem0), all taps x channels in two column tiles of one workgroup (wgrad_st_kernel)
bool st_ok(const addk_conv_wgrad_args* a) {
  return (addk_get_fast_paths() & ADDK_FAST_WGRAD_RS) && a->src.C <= 4 && a->KH * a->KW * a->src.C <= 32 && a->Cout <= 64 && a->Cout % 4 == 0 &&
         aligned16(a->dy) && a->lddy % 4 == 0 && (long)a->N * a->OH * a->OW >= 65536;
}
int kind_of(const addk_conv_wgrad_args* a) { return st_ok(a) ? 8 : h3_ok(a) ? 5 : h1_ok(a) ? 9 : hk_ok(a) ? 7 : rs_ok(a) ? 6 : os_kind(a->Cout, a->src.C); }
// Halo-patch scheduling.  A block runs `steps` row segments; blocks are dispatched in grid order as CU slots free up
// (2 resident blocks per CU at NT=2, 3 at NT=1), so what matters is that the LAST round of blocks is nearly full:
// pick the segment count per block that minimises  ceil(blocks / slots) * (steps + start-up)  over the whole launch.
struct H3Op { int tiles; long nseg; };
// precision of the split-bf16 weight-gradient kernels for a launch of cty output-channel tiles per workgroup (0: fp32 kernels); tail_x3 (mode 3): the
// 128-channel blocks are the exit heads (decoder, ASPP) -> three terms; stem1's 64-channel blocks keep six
inline int wg_np_of(int cty) { const int m = addk_get_conv_precision(); return m == 2 ? 3 : m == 1 ? 2 : m == 3 ? (cty == 8 ? 2 : 3) : 0; }
inline bool h3b_runs(int cty) { return wg_np_of(cty) && wgrad_split_enabled() && (cty == 8 || wgrad_split_narrow()); }
// input-channel tiles per workgroup of the 3x3 halo-patch kernel: 2 (512 threads sharing one staged dy tile) on the split-bf16 kernel when the channels fill
// whole pairs of tiles (a half-empty pair costs what the shared dy saves: 304 and 400 channels measured equal, 256 -7 %, stem1's 64 -> 64 -14 %)
inline int h3_ng(int Cout, int C) {
  // (re-measured with the split-fp16 arithmetic: the shared dy tile still wins, 2.2 vs 2.38 ms per step: profiles/r05_wgrad_h3b_f16_pipelined.txt)
  return (C % 32 == 0 && h3b_runs(Cout % 128 == 0 ? 8 : 4)) ? 2 : 1;
}
inline int h3_tiles(int Cout, int C) { const int nt = Cout % 128 == 0 ? 2 : 1; return (Cout / (64 * nt)) * cdiv(C, 16); }
// at most 32 workspace slices, or as many as it takes for the op alone to offer one block per slot (few-tile convs: stem1)
inline int h3_max_splits(int tiles) { const int s = cdiv(768, tiles); return s > 32 ? s : 32; }
inline int h3_splits(long nseg, int steps, int tiles, int cap_tiles = 0) {      // cap_tiles: the tile count the workspace was sized with
  int sp = cdiv(nseg, steps);
  const int cap = h3_max_splits(cap_tiles ? cap_tiles : tiles);
  if (sp > cap) sp = cap;
  return cdiv(nseg, cdiv(nseg, sp));
}
int h3_pick_steps(const H3Op* ops, int n, int nt, int ng = 1) {
  const long slots = ng == 2 ? 256 : nt == 2 ? 512 : 768;
  int best = 64; double best_cost = -1.0;
  for (int steps = 8; steps <= 256; ++steps) {
    long blocks = 0; long longest = 0;
    for (int i = 0; i < n; ++i) {
      const int sp = h3_splits(ops[i].nseg, steps, ops[i].tiles);
      blocks += (long)ops[i].tiles * sp;
      const long ch = cdiv(ops[i].nseg, sp);
      if (ch > longest) longest = ch;
    }
    const double cost = (double)cdiv(blocks, slots) * ((double)longest + 1.5);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = steps; }
  }
  return best;
}
// `budget` = workgroups this conv should contribute.  A lone launch needs ~1536 of them to fill the chip even if that
// leaves a block a single 64-pixel step; inside a batch the other convs provide the parallelism, so each block gets
// >= 8 steps and the per-block epilogue (cross-wave combine + partial tile written to the workspace) is amortised.
int pick_splits(long P, int tiles, int budget = 1536, int min_steps = 1) {
  long maxs = cdiv(P, (long)min_steps * KP);
  long want = cdiv(budget, tiles);
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

}  // namespace

#ifdef ADDK_WG_DIAG
// the eight counters summed over the waves since the last call; resets them
extern "C" int addk_wg_diag(unsigned long long* out8) {
  unsigned long long h[64][8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg_diag), sizeof h) != hipSuccess) return ADDK_ERR_INVALID;
  for (int k = 0; k < 8; ++k) { out8[k] = 0; for (int i = 0; i < 64; ++i) out8[k] += h[i][k]; }
#ifdef ADDK_WG_DIAG2
  { unsigned long long h2[64][2]; (void)hipMemcpyFromSymbol(h2, HIP_SYMBOL(g_wg_diag2), sizeof h2); unsigned long long a = 0, b = 0;
    for (int i = 0; i < 64; ++i) { a += h2[i][0]; b += h2[i][1]; h2[i][0] = h2[i][1] = 0; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_diag2), h2, sizeof h2);
    fprintf(stderr, "      inside split + LDS stores: waiting for the loads %.1f %% of wave life, dy part %.1f %%\n", 100.0 * a / out8[0], 100.0 * b / out8[0]); }
#endif
  for (int i = 0; i < 64; ++i) for (int k = 0; k < 8; ++k) h[i][k] = 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_diag), h, sizeof h) == hipSuccess ? ADDK_OK : ADDK_ERR_INVALID;
}
#endif
extern "C" int64_t addk_conv_wgrad_ws(int64_t P, int32_t Cout, int32_t C, int32_t taps) {
  int cty, ctz; pick_tiles(Cout, C, &cty, &ctz);
  int tiles = cdiv(Cout, 16 * cty) * taps * cdiv(C, 16 * ctz);
  int splits = pick_splits(P, tiles);
  if ((taps == 9 || taps == 25) && Cout >= 32 && Cout <= 160 && C >= 16) {      // cells' dilated convs: wgrad_hk_kernel
    const int hs = h3_max_splits(hk_tiles(Cout, C, taps == 9 ? 3 : 5));
    if (hs > splits) splits = hs;
  }
  if (taps == 9 && Cout % 64 == 0 && C >= 16) {      // the halo-patch kernel may be chosen
    const int hs = h3_max_splits(h3_tiles(Cout, C));
    if (hs > splits) splits = hs;
  }
  if (C <= 4 && taps * C <= 32 && Cout <= 64 && splits < 1024) splits = 1024;      // few input channels: wgrad_st_kernel slices the pixels only (st_ok)
  return (int64_t)splits * Cout * taps * C;
}

static int wg_fill(const addk_conv_wgrad_args* a, WgK& k, int& cty, int& ctz, int& tiles, bool check_ws, int budget = 1536, int min_steps = 1, int h3_steps = 0) {
  ADDK_REQUIRE(a && a->dy && a->src.x && a->dw && (a->ws || !check_ws), "conv_wgrad: null pointer");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->Cout > 0 && a->src.C > 0, "conv_wgrad: empty shape");
  ADDK_REQUIRE(a->lddy >= a->Cout && a->src.ld >= a->src.C, "conv_wgrad: short stride");
  ADDK_REQUIRE(a->w_choff + a->src.C <= a->cin_total && a->ldw >= a->KH * a->KW * a->cin_total, "conv_wgrad: weight layout");
  ADDK_REQUIRE((a->src.a == nullptr) == (a->src.b == nullptr), "conv_wgrad: a/b must come together");
  k.dy = a->dy; k.lddy = a->lddy; k.Cout = a->Cout;
  k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.KH = a->KH; k.KW = a->KW; k.stride = a->stride; k.pad = a->pad; k.dil = a->dil;
  k.src = a->src; k.ws = a->ws;
  pick_tiles(a->Cout, a->src.C, &cty, &ctz);
  k.taps = a->KH * a->KW; k.nyt = cdiv(a->Cout, 16 * cty); k.nzt = cdiv(a->src.C, 16 * ctz);
  ADDK_REQUIRE((long)a->N * a->OH * a->OW < (1L << 30) && (long)a->N * a->H * a->W < (1L << 30), "conv_wgrad: tensor too large for 32-bit pixel indexing");
  k.P = a->N * a->OH * a->OW;
  tiles = k.nyt * k.taps * k.nzt;
  k.splits = pick_splits(k.P, tiles, budget, min_steps);
  ADDK_REQUIRE(!check_ws || kind_of(a) == 5 || kind_of(a) == 7 || a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  k.chunkP = cdiv(cdiv(k.P, k.splits), KP) * KP;
  const int kd = kind_of(a);      // ONE decision for geometry and launch: a shape several kernels accept (1x1, Cout = 128: h1 and rs) must not get the geometry of one and the launch of another
  if (kd == 5) {      // halo-patch kernel: pixel range in 64-pixel row segments, never more slices than the workspace bound
    const int nt = a->Cout % 128 == 0 ? 2 : 1, ng = h3_ng(a->Cout, a->src.C);
    cty = 4 * nt; ctz = ng;
    k.nyt = a->Cout / (64 * nt); k.nzt = cdiv(a->src.C, 16 * ng);
    tiles = k.nyt * k.nzt;
    const long nseg = (long)a->N * a->OH * cdiv(a->OW, H3_KP);
    if (h3_steps <= 0) { H3Op o{tiles, nseg}; h3_steps = h3_pick_steps(&o, 1, nt, ng); }
    k.splits = h3_splits(nseg, h3_steps, tiles, h3_tiles(a->Cout, a->src.C));
    k.chunkP = cdiv(nseg, k.splits);
    ADDK_REQUIRE(!check_ws || a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  }
  if (kd == 9) {      // wide 1x1 heads on the split-bf16 kernel: 128 output x 64 input channels per workgroup, 64-pixel row segments
    cty = 8; ctz = 4;
    k.nyt = a->Cout / 128; k.nzt = cdiv(a->src.C, 16 * H1_TP);
    tiles = k.nyt * k.nzt;
    const long nseg = (long)a->N * a->OH * cdiv(a->OW, H3_KP);
    if (h3_steps <= 0) { H3Op o{tiles, nseg}; h3_steps = h3_pick_steps(&o, 1, 2); }
    long cap = addk_conv_wgrad_ws(k.P, a->Cout, a->src.C, 1) / ((long)a->Cout * a->src.C);      // slices the workspace was sized for
    if (cap < 1) cap = 1;
    long sp = cdiv(nseg, h3_steps); if (sp > cap) sp = cap; if (sp < 1) sp = 1;
    k.chunkP = cdiv(nseg, sp);
    k.splits = cdiv(nseg, k.chunkP);
    ADDK_REQUIRE(!check_ws || a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  }
  k.vecY = aligned16(a->dy) && a->lddy % 4 == 0 && a->Cout % 4 == 0;
  k.vecZ = src_vec_ok(a->src);
  if (kd == 8) {      // few input channels: one workgroup holds every (tap, channel) column; 1024-pixel slices, as many as the workspace bound allows
    cty = 4; ctz = 2;
    k.nyt = 1; k.nzt = 1; tiles = 1;
    int sp = cdiv(k.P, 1024);
    if (sp > 1024) sp = 1024;
    k.chunkP = cdiv(cdiv(k.P, sp), 4) * 4;
    k.splits = cdiv(k.P, k.chunkP);
    ADDK_REQUIRE(!check_ws || a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  } else
  if (kd == 7) {      // halo-patch kernel, taps split across waves: 64-pixel row segments like wgrad_h3
    const int ct = hk_ct(a->Cout, a->KH);
    cty = ct; ctz = a->KH;
    k.nyt = cdiv(a->Cout, 16 * ct); k.nzt = cdiv(a->src.C, 16);
    tiles = k.nyt * k.nzt;
    const long nseg = (long)a->N * a->OH * cdiv(a->OW, H3_KP);
    if (h3_steps <= 0) { H3Op o{tiles, nseg}; h3_steps = h3_pick_steps(&o, 1, ct == 3 ? 1 : 2); }
    k.splits = h3_splits(nseg, h3_steps, tiles);
    k.chunkP = cdiv(nseg, k.splits);
    ADDK_REQUIRE(!check_ws || a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  } else
  if (kd == 6) {      // register-streaming kernel: <= 64-channel tiles, 2048-pixel chunks (128 k-steps per wave), vecY/vecZ carry the tile strides
    const int cap = pick_splits(k.P, tiles);
    k.vecY = rs_tile(a->Cout); k.vecZ = rs_tile(a->src.C);
    cty = rs_lay(k.vecY); ctz = rs_lay(k.vecZ);
    k.nyt = cdiv(a->Cout, k.vecY); k.nzt = cdiv(a->src.C, k.vecZ);
    tiles = k.nyt * k.taps * k.nzt;
    const int rs_chunk = 2048;
    int sp = cdiv(k.P, rs_chunk);
    if (sp > cap) sp = cap;
    if (sp < 1) sp = 1;
    k.chunkP = cdiv(cdiv(k.P, sp), 4) * 4;
    k.splits = cdiv(k.P, k.chunkP);
  }
  k.dw = a->dw; k.ldw = a->ldw; k.cin_total = a->cin_total; k.w_choff = a->w_choff; k.accumulate = a->accumulate;
  return 0;
}

static int wg_launch(int kind, int cty, int ctz, dim3 grid, hipStream_t st, const WgK& k, const WgK* ops, const int4* work) {
  bool done = false;
#define ADDK_OS(K_, TY_, TZ_) \
  if (kind == K_) { \
    if (ops) hipLaunchKernelGGL((wgrad_os_kernel<TY_, TZ_, true>), grid, dim3(256), 0, st, k, ops, work); \
    else hipLaunchKernelGGL((wgrad_os_kernel<TY_, TZ_, false>), grid, dim3(256), 0, st, k, ops, work); \
    done = true; }
  ADDK_OS(1, 4, 2) ADDK_OS(2, 3, 3) ADDK_OS(3, 2, 2) ADDK_OS(4, 4, 4)
#undef ADDK_OS
  const int hk_np = addk_get_conv_precision() >= 2 ? 3 : addk_get_conv_precision() == 1 ? 2 : 0;      // (tail_x3: the cells' convs keep six terms)
#define ADDK_HKB_(K_, C_, B_, P_) { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hkb_kernel<K_, C_, B_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL((wgrad_hkb_kernel<K_, C_, B_, P_>), grid, dim3(256), wg_hkb_lds(K_, C_, P_), st, k, ops, work); done = true; }
#define ADDK_HKB(K_, C_) \
  if (kind == 7 && ctz == K_ && cty == C_ && hk_np && wgrad_split_enabled() && wgrad_split_narrow()) { \
    if (ops) { if (hk_np == 3) ADDK_HKB_(K_, C_, true, 3) else ADDK_HKB_(K_, C_, true, 2) } \
    else { if (hk_np == 3) ADDK_HKB_(K_, C_, false, 3) else ADDK_HKB_(K_, C_, false, 2) } }
  ADDK_HKB(3, 3) ADDK_HKB(3, 5) ADDK_HKB(5, 3)
#undef ADDK_HKB
#undef ADDK_HKB_
#define ADDK_HK(K_, C_) \
  if (!done && kind == 7 && ctz == K_ && cty == C_) { \
    if (ops) hipLaunchKernelGGL((wgrad_hk_kernel<K_, C_, true>), grid, dim3(256), 0, st, k, ops, work); \
    else hipLaunchKernelGGL((wgrad_hk_kernel<K_, C_, false>), grid, dim3(256), 0, st, k, ops, work); \
    done = true; }
  ADDK_HK(3, 3) ADDK_HK(3, 5) ADDK_HK(5, 3)
#undef ADDK_HK
  if (kind == 8) {
    if (ops) hipLaunchKernelGGL((wgrad_st_kernel<true>), grid, dim3(256), 0, st, k, ops, work);
    else hipLaunchKernelGGL((wgrad_st_kernel<false>), grid, dim3(256), 0, st, k, ops, work);
    done = true;
  }
  const bool rs_f16 = addk_get_conv_precision() == 1;      // f16x3: the narrow cell convs' weight gradients on the fp16 matrix pipe too (ABAB: step 29.57 -> 29.2 ms)
#define ADDK_RS(A_, B_) \
  if (kind == 6 && cty == A_ && ctz == B_) { \
    if (rs_f16) { if (ops) hipLaunchKernelGGL((wgrad_rs_kernel<A_, B_, true, true>), grid, dim3(256), 0, st, k, ops, work); \
                  else hipLaunchKernelGGL((wgrad_rs_kernel<A_, B_, false, true>), grid, dim3(256), 0, st, k, ops, work); } \
    else if (ops) hipLaunchKernelGGL((wgrad_rs_kernel<A_, B_, true>), grid, dim3(256), 0, st, k, ops, work); \
    else hipLaunchKernelGGL((wgrad_rs_kernel<A_, B_, false>), grid, dim3(256), 0, st, k, ops, work); \
    done = true; }
  ADDK_RS(3, 3) ADDK_RS(3, 4) ADDK_RS(4, 3) ADDK_RS(4, 4)
#undef ADDK_RS
  // 3x3 convolutions in a split-bf16 mode: the transposed-read kernel — the 128-channel blocks (decoder, ASPP) and the
  // 64-channel blocks (stem1: 0.76 -> 0.61 ms alone; ADDK_WGRAD_SPLIT_NARROW=0 keeps stem1 and the cells' dilated convs on fp32)
  const int wg_np = wg_np_of(cty);
  if (kind == 5 && ctz == 2 && !h3b_runs(cty)) { addk_set_error("conv_wgrad: launch prepared for the split-bf16 kernel, but the precision mode / ADDK_WGRAD_SPLIT changed since"); return ADDK_ERR_INVALID; }
#define ADDK_H3B_(N_, B_, P_, G_) { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_h3b_kernel<N_, B_, P_, G_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL((wgrad_h3b_kernel<N_, B_, P_, G_>), grid, dim3(256 * G_), wg_h3b_lds(N_, P_, G_), st, k, ops, work); done = true; }
#define ADDK_H3B(N_, G_) \
    if (ops) { if (wg_np == 3) ADDK_H3B_(N_, true, 3, G_) else ADDK_H3B_(N_, true, 2, G_) } \
    else { if (wg_np == 3) ADDK_H3B_(N_, false, 3, G_) else ADDK_H3B_(N_, false, 2, G_) }
  if (kind == 5 && cty == 8 && h3b_runs(cty)) { if (ctz == 2) { ADDK_H3B(2, 2) } else { ADDK_H3B(2, 1) } }
  if (kind == 5 && cty == 4 && h3b_runs(cty)) { if (ctz == 2) { ADDK_H3B(1, 2) } else { ADDK_H3B(1, 1) } }
#undef ADDK_H3B
#undef ADDK_H3B_
  if (kind == 9) {
    if (!h3b_runs(8)) { addk_set_error("conv_wgrad: launch prepared for the split-bf16 kernel, but the precision mode / ADDK_WGRAD_SPLIT changed since"); return ADDK_ERR_INVALID; }
#define ADDK_H1B_(B_, P_) { \
      static bool attr = false; \
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_h1b_kernel<B_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
      hipLaunchKernelGGL((wgrad_h1b_kernel<B_, P_>), grid, dim3(256), wg_h1b_lds(P_), st, k, ops, work); done = true; }
    if (ops) { if (wg_np == 3) ADDK_H1B_(true, 3) else ADDK_H1B_(true, 2) }
    else { if (wg_np == 3) ADDK_H1B_(false, 3) else ADDK_H1B_(false, 2) }
#undef ADDK_H1B_
  }
#define ADDK_H3(NT_) \
  if (!done && kind == 5 && cty == 4 * NT_) { \
    if (ops) hipLaunchKernelGGL((wgrad_h3_kernel<NT_, true>), grid, dim3(256), 0, st, k, ops, work); \
    else hipLaunchKernelGGL((wgrad_h3_kernel<NT_, false>), grid, dim3(256), 0, st, k, ops, work); \
    done = true; }
  ADDK_H3(1) ADDK_H3(2)
#undef ADDK_H3
#define ADDK_CASE(Y_, Z_) \
  if (!done && cty == Y_ && ctz == Z_) { \
    if (ops) hipLaunchKernelGGL((wgrad_kernel<Y_, Z_, true>), grid, dim3(256), 0, st, k, ops, work); \
    else hipLaunchKernelGGL((wgrad_kernel<Y_, Z_, false>), grid, dim3(256), 0, st, k, ops, work); \
    done = true; }
  ADDK_CASE(2, 1) ADDK_CASE(2, 3) ADDK_CASE(2, 4) ADDK_CASE(2, 5)
  ADDK_CASE(3, 1) ADDK_CASE(3, 3) ADDK_CASE(3, 4) ADDK_CASE(3, 5)
  ADDK_CASE(4, 1) ADDK_CASE(4, 3) ADDK_CASE(4, 4) ADDK_CASE(4, 5)
  ADDK_CASE(5, 1) ADDK_CASE(5, 3) ADDK_CASE(5, 4) ADDK_CASE(5, 5)
  ADDK_CASE(8, 1) ADDK_CASE(8, 3) ADDK_CASE(8, 4)
#undef ADDK_CASE
  if (!done) { addk_set_error("conv_wgrad: no tile config"); return ADDK_ERR_UNSUPPORTED; }
  return addk_check_launch("conv_wgrad");
}

extern "C" int addk_conv_wgrad(const addk_conv_wgrad_args* a, void* stream) {
  WgK k; int cty, ctz, tiles;
  int rc = wg_fill(a, k, cty, ctz, tiles, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  rc = wg_launch(kind_of(a), cty, ctz, dim3(tiles, k.splits), st, k, nullptr, nullptr);
  if (rc) return rc;
  long n = (long)a->Cout * k.taps * a->src.C;
  if (k.splits > 16 && n <= 65536) {
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, a->ws, k.splits, a->Cout, k.taps, a->src.C,
                       a->dw, a->ldw, a->cin_total, a->w_choff, a->accumulate);
  } else {
    int rb = cdiv(n, 256); if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, a->ws, k.splits, a->Cout, k.taps, a->src.C,
                       a->dw, a->ldw, a->cin_total, a->w_choff, a->accumulate);
  }
  return addk_check_launch("conv_wgrad_reduce");
}

// ---- batched weight gradients -------------------------------------------------------------------------------------
extern "C" int addk_conv_wgrad_config(const addk_conv_wgrad_args* a, int32_t* cfg) {
  WgK k; int cty, ctz, tiles;
  int rc = wg_fill(a, k, cty, ctz, tiles, false);
  if (rc) return rc;
  cfg[0] = kind_of(a); cfg[1] = cty; cfg[2] = ctz; cfg[3] = tiles * k.splits;
  return 0;
}

extern "C" int64_t addk_conv_wgrad_batch_prepare(const addk_conv_wgrad_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta) {
  if (!a || n <= 0 || !meta) { addk_set_error("wgrad_batch_prepare: bad args"); return ADDK_ERR_INVALID; }
  long nblocks = 0, nrblocks = 0;
  int kind0 = -1, cty0 = 0, ctz0 = 0;
  int budget = 8192 / n; if (budget < 32) budget = 32; if (budget > 1536) budget = 1536;
  const int min_steps = n >= 4 ? 8 : 1;
  int h3_steps = 0;
  const int kd0 = kind_of(&a[0]);
  if (kd0 == 7) {
    H3Op* ho = (H3Op*)malloc(sizeof(H3Op) * n);
    for (int i = 0; i < n; ++i) { ho[i].tiles = hk_tiles(a[i].Cout, a[i].src.C, a[i].KH); ho[i].nseg = (long)a[i].N * a[i].OH * cdiv(a[i].OW, H3_KP); }
    h3_steps = h3_pick_steps(ho, n, hk_ct(a[0].Cout, a[0].KH) == 3 ? 1 : 2);
    free(ho);
  }
  if (kd0 == 9) {
    H3Op* ho = (H3Op*)malloc(sizeof(H3Op) * n);
    for (int i = 0; i < n; ++i) { ho[i].tiles = (a[i].Cout / 128) * cdiv(a[i].src.C, 16 * H1_TP); ho[i].nseg = (long)a[i].N * a[i].OH * cdiv(a[i].OW, H3_KP); }
    h3_steps = h3_pick_steps(ho, n, 2);
    free(ho);
  }
  if (kd0 == 5) {
    H3Op* ho = (H3Op*)malloc(sizeof(H3Op) * n);
    const int nt = a[0].Cout % 128 == 0 ? 2 : 1, ng = h3_ng(a[0].Cout, a[0].src.C);
    for (int i = 0; i < n; ++i) {
      const int nti = a[i].Cout % 128 == 0 ? 2 : 1;
      ho[i].tiles = (a[i].Cout / (64 * nti)) * cdiv(a[i].src.C, 16 * h3_ng(a[i].Cout, a[i].src.C));
      ho[i].nseg = (long)a[i].N * a[i].OH * cdiv(a[i].OW, H3_KP);
    }
    h3_steps = h3_pick_steps(ho, n, nt, ng);
    free(ho);
  }
  for (int i = 0; i < n; ++i) {
    WgK k; int cty, ctz, tiles;
    int rc = wg_fill(&a[i], k, cty, ctz, tiles, host_blob != nullptr, budget, min_steps, h3_steps);
    if (rc) return rc;
    int kind = kind_of(&a[i]);
    if (i == 0) { kind0 = kind; cty0 = cty; ctz0 = ctz; }
    if (kind != kind0 || cty != cty0 || ctz != ctz0) { addk_set_error("wgrad_batch_prepare: mixed tile configurations"); return ADDK_ERR_INVALID; }
    nblocks += (long)tiles * k.splits;
    long ne = (long)a[i].Cout * k.taps * a[i].src.C;
    nrblocks += (k.splits > 16 && ne <= 65536) ? cdiv(ne, 64) : cdiv(ne, 256);
  }
  const int64_t off_work = ((int64_t)n * sizeof(WgK) + 15) / 16 * 16;
  const int64_t off_rwork = off_work + nblocks * (int64_t)sizeof(int4);
  const int64_t total = off_rwork + nrblocks * (int64_t)sizeof(int4);
  meta[0] = kind0; meta[1] = cty0; meta[2] = ctz0; meta[3] = n; meta[4] = off_work; meta[5] = nblocks; meta[6] = off_rwork; meta[7] = nrblocks;
  if (!host_blob) return total;
  if (blob_bytes < total) { addk_set_error("wgrad_batch_prepare: blob too small"); return ADDK_ERR_INVALID; }
  WgK* ops = reinterpret_cast<WgK*>(host_blob);
  int4* work = reinterpret_cast<int4*>(reinterpret_cast<char*>(host_blob) + off_work);
  int4* rwork = reinterpret_cast<int4*>(reinterpret_cast<char*>(host_blob) + off_rwork);
  long b = 0, rb = 0;
  for (int i = 0; i < n; ++i) {
    int cty, ctz, tiles;
    wg_fill(&a[i], ops[i], cty, ctz, tiles, true, budget, min_steps, h3_steps);
    if (kind0 == 6) {
      // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so within a group of 8 pixel chunks the tiles
      // (taps x channel blocks) of one chunk are placed 8 apart: they share an XCD and its L2 serves the 9/25 re-reads
      // of that chunk's dy / activation rows.
      const int sp = ops[i].splits;
      for (int g0 = 0; g0 < sp; g0 += 8) {
        const int gn = sp - g0 < 8 ? sp - g0 : 8;
        for (int x = 0; x < tiles; ++x)
          for (int yy = 0; yy < gn; ++yy) work[b++] = make_int4(i, x, g0 + yy, 0);
      }
    } else
    for (int y = 0; y < ops[i].splits; ++y)
      for (int x = 0; x < tiles; ++x) work[b++] = make_int4(i, x, y, 0);
    long ne = (long)a[i].Cout * ops[i].taps * a[i].src.C;
    if (ops[i].splits > 16 && ne <= 65536) { for (long e = 0; e < ne; e += 64) rwork[rb++] = make_int4(i, (int)e, 1, 0); }
    else { for (long e = 0; e < ne; e += 256) rwork[rb++] = make_int4(i, (int)e, 0, 0); }
  }
  return total;
}

extern "C" int addk_conv_wgrad_batch_run(const void* dev_blob, const int64_t* meta, void* stream) {
  ADDK_REQUIRE(dev_blob && meta && meta[3] > 0 && meta[5] > 0 && meta[7] > 0, "wgrad_batch_run: bad args");
  const WgK* ops = reinterpret_cast<const WgK*>(dev_blob);
  const int4* work = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(dev_blob) + meta[4]);
  const int4* rwork = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(dev_blob) + meta[6]);
  hipStream_t st = (hipStream_t)stream;
  WgK dummy{};
  int rc = wg_launch((int)meta[0], (int)meta[1], (int)meta[2], dim3((unsigned)meta[5], 1), st, dummy, ops, work);
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)meta[7]), dim3(256), 0, st, ops, rwork);
  return addk_check_launch("conv_wgrad_batch_reduce");
}
