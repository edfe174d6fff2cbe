// Weight gradient of the dense convolution for one source:
//     dW[co][tap][c] = sum_p dy[p, co] * Z[p @ tap, c],   Z = relu?(a*x+b) recomputed on the fly.
// GEMM with M = output channels (tile 16*CTY), N = input channels of one tap (tile 16*CTZ) and the
// reduction over pixels.  A block owns one (co tile, tap, c tile) and one slice of the pixel range;
// its four waves each take 16 of the 64 pixels staged per step (both tiles are staged in their memory
// order [pixel][channel]; MFMA fragments are read with ds_read_b32, rows padded so that the two pixel
// rows a 32-lane group touches fall on disjoint banks).  Wave partials are combined through LDS in a
// fixed order and the per-slice tiles go to a workspace that addk reduces deterministically into dW.
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
};

constexpr int KP = 64;
constexpr int ldpad(int bc) { return (bc % 32 == 16) ? bc : bc + 16; }

template <int CTY, int CTZ>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgK p) {
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
  int bx = blockIdx.x;
  const int zt = bx % p.nzt; bx /= p.nzt;
  const int tap = bx % p.taps; const int yt = bx / p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = yt * BCY, c0 = zt * BCZ;
  const int ohw = p.OH * p.OW;
  const int pbeg = blockIdx.y * p.chunkP;
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
      if (row < KP) st4(&Ys[row * LY + 4 * q], ry[j]);
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
      if (row < KP) st4(&Zs[row * LZ + 4 * q], v);
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
  float* wsb = p.ws + (long)blockIdx.y * p.Cout * p.taps * C;
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
template <int TY, int TZ>
__global__ void __launch_bounds__(256) wgrad_os_kernel(const WgK p) {
  constexpr int BCY = 32 * TY, BCZ = 32 * TZ;
  constexpr int LY = ldpad(BCY), LZ = ldpad(BCZ);
  constexpr int NYJ = (KP * BCY / 4 + 255) / 256;
  constexpr int NZJ = (KP * BCZ / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[KP * LY + KP * LZ];
  float* Ys = lds;
  float* Zs = lds + KP * LY;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int wy = wave >> 1, wz = wave & 1;
  int bx = blockIdx.x;
  const int zt = bx % p.nzt; bx /= p.nzt;
  const int tap = bx % p.taps; const int yt = bx / p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = yt * BCY, c0 = zt * BCZ;
  const int ohw = p.OH * p.OW;
  const int pbeg = blockIdx.y * p.chunkP;
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
      if (row < KP) st4(&Ys[row * LY + 4 * q], ry[j]);
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
      if (row < KP) st4(&Zs[row * LZ + 4 * q], v);
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
  float* wsb = p.ws + (long)blockIdx.y * p.Cout * p.taps * C;
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
void pick_tiles(int Cout, int C, int* cty, int* ctz) {
  *cty = pick_cty(Cout); *ctz = pick_ctz(C);
  if (*cty == 8 && *ctz == 5) *ctz = 4;   // 8x5 accumulator tiles would not leave room for the staging registers
  if (Cout >= 128 && Cout % 128 == 0 && C >= 48) {   // wide heads: output-split kernel, 128 x 64 tiles
    *cty = 8; *ctz = 4;
  }
}
bool use_output_split(int Cout, int C) { return Cout >= 128 && Cout % 128 == 0 && C >= 48; }
int pick_splits(long P, int tiles) {
  long maxs = cdiv(P, KP);                // at least one staged step per block
  long want = cdiv(1536, tiles);          // ~6 resident blocks per CU keep enough loads in flight for the HBM-bound shapes
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

}  // namespace

extern "C" int64_t addk_conv_wgrad_ws(int64_t P, int32_t Cout, int32_t C, int32_t taps) {
  int cty, ctz; pick_tiles(Cout, C, &cty, &ctz);
  int tiles = cdiv(Cout, 16 * cty) * taps * cdiv(C, 16 * ctz);
  return (int64_t)pick_splits(P, tiles) * Cout * taps * C;
}

extern "C" int addk_conv_wgrad(const addk_conv_wgrad_args* a, void* stream) {
  ADDK_REQUIRE(a && a->dy && a->src.x && a->dw && a->ws, "conv_wgrad: null pointer");
  ADDK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->Cout > 0 && a->src.C > 0, "conv_wgrad: empty shape");
  ADDK_REQUIRE(a->lddy >= a->Cout && a->src.ld >= a->src.C, "conv_wgrad: short stride");
  ADDK_REQUIRE(a->w_choff + a->src.C <= a->cin_total && a->ldw >= a->KH * a->KW * a->cin_total, "conv_wgrad: weight layout");
  ADDK_REQUIRE((a->src.a == nullptr) == (a->src.b == nullptr), "conv_wgrad: a/b must come together");
  WgK k;
  k.dy = a->dy; k.lddy = a->lddy; k.Cout = a->Cout;
  k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.KH = a->KH; k.KW = a->KW; k.stride = a->stride; k.pad = a->pad; k.dil = a->dil;
  k.src = a->src; k.ws = a->ws;
  int cty, ctz; pick_tiles(a->Cout, a->src.C, &cty, &ctz);
  k.taps = a->KH * a->KW; k.nyt = cdiv(a->Cout, 16 * cty); k.nzt = cdiv(a->src.C, 16 * ctz);
  ADDK_REQUIRE((long)a->N * a->OH * a->OW < (1L << 30) && (long)a->N * a->H * a->W < (1L << 30), "conv_wgrad: tensor too large for 32-bit pixel indexing");
  k.P = a->N * a->OH * a->OW;
  const int tiles = k.nyt * k.taps * k.nzt;
  k.splits = pick_splits(k.P, tiles);
  ADDK_REQUIRE(a->ws_floats >= (int64_t)k.splits * a->Cout * k.taps * a->src.C, "conv_wgrad: workspace too small");
  k.chunkP = cdiv(cdiv(k.P, k.splits), KP) * KP;
  k.vecY = aligned16(a->dy) && a->lddy % 4 == 0 && a->Cout % 4 == 0;
  k.vecZ = src_vec_ok(a->src);
  dim3 grid(tiles, k.splits);
  hipStream_t st = (hipStream_t)stream;
  bool done = false;
  if (use_output_split(a->Cout, a->src.C)) {
    hipLaunchKernelGGL((wgrad_os_kernel<4, 2>), grid, dim3(256), 0, st, k);
    done = true;
  }
#define ADDK_CASE(Y_, Z_) \
  if (!done && cty == Y_ && ctz == Z_) { hipLaunchKernelGGL((wgrad_kernel<Y_, Z_>), grid, dim3(256), 0, st, k); done = true; }
  ADDK_CASE(2, 1) ADDK_CASE(2, 3) ADDK_CASE(2, 4) ADDK_CASE(2, 5)
  ADDK_CASE(3, 1) ADDK_CASE(3, 3) ADDK_CASE(3, 4) ADDK_CASE(3, 5)
  ADDK_CASE(4, 1) ADDK_CASE(4, 3) ADDK_CASE(4, 4) ADDK_CASE(4, 5)
  ADDK_CASE(5, 1) ADDK_CASE(5, 3) ADDK_CASE(5, 4) ADDK_CASE(5, 5)
  ADDK_CASE(8, 1) ADDK_CASE(8, 3) ADDK_CASE(8, 4)
#undef ADDK_CASE
  if (!done) { addk_set_error("conv_wgrad: no tile config"); return ADDK_ERR_UNSUPPORTED; }
  int rc = addk_check_launch("conv_wgrad");
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
