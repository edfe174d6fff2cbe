// Global average pool, 3x3 pooling primitives, NCHW<->NHWC boundary transforms.
#include "common.h"

namespace {

struct GapK {
  addk_src src; int N; long HW;
  float* y; int ldy; float* ws; int rows;
  const float* dy; int lddy; float* g; int ldg; int accumulate; double* dab;
  int nq, npl, vec;
  int Cfull, c0;        // channel chunk [c0, c0 + src.C) of a Cfull-channel tensor (chunks of <= 1024 channels per launch)
};

// partial sums: ws[(n*rows + r)*C + c] = sum over the r-th pixel slice of image n of relu?(a*x+b)
__global__ void __launch_bounds__(256) gap_partial_kernel(const GapK p) {
  extern __shared__ float redt[];      // [C4]
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q, C = p.src.C, nrem = C - c;
  const int n = blockIdx.y, r = blockIdx.x;
  float4 s = zero4();
  if (active) {
    const float* b = p.src.x + (long)n * p.HW * p.src.ld + c;
    for (long i = (long)r * p.npl + pl; i < p.HW; i += (long)p.rows * p.npl) {
      float4 v = prologue4(ld4g(b + i * p.src.ld, nrem, p.vec), p.src.a, p.src.b, c, nrem, p.src.relu != 0, p.vec);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  for (int k = 0; k < p.npl; ++k) {
    if (active && pl == k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) redt[c + e] = (k == 0) ? get4(s, e) : redt[c + e] + get4(s, e);
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < C; i += 256) p.ws[((long)n * p.rows + r) * p.Cfull + p.c0 + i] = redt[i];
}

// one wave per (image, channel): lanes stride over the partial rows, fixed-order butterfly
__global__ void __launch_bounds__(256) gap_final_kernel(const float* ws, int N, int rows, int C, float inv, float* y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += ws[((long)n * rows + r) * C + c];
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) y[(long)n * ldy + c] = (float)(s * inv);
}

__global__ void __launch_bounds__(256) gap_bwd_kernel(const GapK p) {
  extern __shared__ double redd[];     // [C4][2]
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q, C = p.src.C, nrem = C - c;
  const long P = (long)p.N * p.HW;
  const float inv = 1.f / (float)p.HW;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = zero4();
  double sA[4] = {0.0, 0.0, 0.0, 0.0}, sB[4] = {0.0, 0.0, 0.0, 0.0};
  if (active && p.src.a) { av = ld4g(p.src.a + c, nrem, p.vec); bv = ld4g(p.src.b + c, nrem, p.vec); }
  if (active) {
    for (long pp = (long)blockIdx.x * p.npl + pl; pp < P; pp += (long)gridDim.x * p.npl) {
      int n = (int)(pp / p.HW);
      float4 d = ld4g(p.dy + (long)n * p.lddy + c, nrem, false);
      d.x *= inv; d.y *= inv; d.z *= inv; d.w *= inv;
      float4 x = ld4g(p.src.x + pp * p.src.ld + c, nrem, p.vec);
      if (p.src.relu) {
        if (!(fmaf(av.x, x.x, bv.x) > 0.f)) d.x = 0.f;
        if (!(fmaf(av.y, x.y, bv.y) > 0.f)) d.y = 0.f;
        if (!(fmaf(av.z, x.z, bv.z) > 0.f)) d.z = 0.f;
        if (!(fmaf(av.w, x.w, bv.w) > 0.f)) d.w = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { sA[e] += (double)get4(d, e) * (double)get4(x, e); sB[e] += (double)get4(d, e); }
      float4 gv = make_float4(d.x * av.x, d.y * av.y, d.z * av.z, d.w * av.w);
      float* gp = p.g + pp * p.ldg + c;
      if (p.accumulate) { float4 o = ld4g(gp, nrem, p.vec); gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
      st4g(gp, gv, nrem, p.vec);
    }
  }
  if (p.dab) {
    for (int r = 0; r < p.npl; ++r) {
      if (active && pl == r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double* d = &redd[(c + e) * 2];
          d[0] = (r == 0) ? sA[e] : d[0] + sA[e];
          d[1] = (r == 0) ? sB[e] : d[1] + sB[e];
        }
      }
      __syncthreads();
    }
    for (int k = threadIdx.x; k < C * 2; k += 256) p.dab[((long)blockIdx.x * p.Cfull + p.c0) * 2 + k] = redd[k];
  }
}

// ---- 3x3 pooling, pad 1 (registry primitives 1 and 2; cold) ----
struct PoolK {
  addk_src src; int N, H, W, OH, OW, stride, mode;
  float* y; int ldy; const float* dy; int lddy; float* g; int ldg; int accumulate;
};

__device__ __forceinline__ float lazy1(const addk_src& s, const float* px, int c) {
  float v = px[c];
  if (s.a) v = fmaf(s.a[c], v, s.b[c]);
  if (s.relu) v = fmaxf(v, 0.f);
  return v;
}

__global__ void pool3_fwd_kernel(const PoolK p) {
  const int C = p.src.C;
  long total = (long)p.N * p.OH * p.OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long pp = i / C;
    int ow = (int)(pp % p.OW); long r = pp / p.OW; int oh = (int)(r % p.OH); int n = (int)(r / p.OH);
    float acc = p.mode == 0 ? -INFINITY : 0.f; int cnt = 0;
    for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
      int ih = oh * p.stride - 1 + kh, iw = ow * p.stride - 1 + kw;
      if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
        float v = lazy1(p.src, p.src.x + ((long)(n * p.H + ih) * p.W + iw) * p.src.ld, c);
        if (p.mode == 0) acc = fmaxf(acc, v); else acc += v;
        ++cnt;
      }
    }
    p.y[pp * p.ldy + c] = p.mode == 0 ? acc : acc / (float)cnt;
  }
}

// gather form: input pixel enumerates the windows that contain it
__global__ void pool3_bwd_kernel(const PoolK p) {
  const int C = p.src.C;
  long total = (long)p.N * p.H * p.W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long pp = i / C;
    int iw = (int)(pp % p.W); long r = pp / p.W; int ih = (int)(r % p.H); int n = (int)(r / p.H);
    const float* xin = p.src.x + pp * p.src.ld;
    float xv = lazy1(p.src, xin, c);
    float gsum = 0.f;
    for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
      int th = ih + 1 - kh, tw = iw + 1 - kw;
      if (th < 0 || tw < 0 || th % p.stride || tw % p.stride) continue;
      int oh = th / p.stride, ow = tw / p.stride;
      if (oh >= p.OH || ow >= p.OW) continue;
      float d = p.dy[((long)(n * p.OH + oh) * p.OW + ow) * p.lddy + c];
      if (p.mode == 0) {
        // the first maximal element in window scan order receives the gradient (ATen max_pool2d semantics)
        float best = -INFINITY; int bh = -1, bw = -1;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
          int jh = oh * p.stride - 1 + a, jw = ow * p.stride - 1 + b;
          if ((unsigned)jh < (unsigned)p.H && (unsigned)jw < (unsigned)p.W) {
            float v = lazy1(p.src, p.src.x + ((long)(n * p.H + jh) * p.W + jw) * p.src.ld, c);
            if (v > best || bh < 0) { best = v; bh = jh; bw = jw; }
          }
        }
        if (bh == ih && bw == iw) gsum += d;
      } else {
        int h0 = max(oh * p.stride - 1, 0), h1 = min(oh * p.stride + 1, p.H - 1);
        int w0 = max(ow * p.stride - 1, 0), w1 = min(ow * p.stride + 1, p.W - 1);
        gsum += d / (float)((h1 - h0 + 1) * (w1 - w0 + 1));
      }
    }
    float a = p.src.a ? p.src.a[c] : 1.f;
    bool m = !p.src.relu || xv > 0.f;
    float gv = m ? gsum * a : 0.f;
    float* gp = p.g + pp * p.ldg + c;
    *gp = p.accumulate ? *gp + gv : gv;
  }
}

// ---- layout ----
__global__ void nchw_to_nhwc_kernel(const float* x, int N, int C, long HW, float* y, int ldy) {
  long total = (long)N * HW;
  for (long pp = (long)blockIdx.x * blockDim.x + threadIdx.x; pp < total; pp += (long)gridDim.x * blockDim.x) {
    int n = (int)(pp / HW); long i = pp - (long)n * HW;
    const float* xp = x + (long)n * C * HW + i;
    float* yp = y + pp * ldy;
    for (int c = 0; c < C; ++c) yp[c] = xp[(long)c * HW];
    for (int c = C; c < ldy; ++c) yp[c] = 0.f;       // padding channels stay finite (conv float4 loads touch them)
  }
}

__global__ void nhwc_to_nchw_kernel(const addk_src s, int N, long HW, float* y) {
  long total = (long)N * HW;
  const int C = s.C;
  for (long pp = (long)blockIdx.x * blockDim.x + threadIdx.x; pp < total; pp += (long)gridDim.x * blockDim.x) {
    int n = (int)(pp / HW); long i = pp - (long)n * HW;
    const float* xp = s.x + pp * s.ld;
    float* yp = y + (long)n * C * HW + i;
    for (int c = 0; c < C; ++c) yp[(long)c * HW] = lazy1(s, xp, c);
  }
}

// gradient of nhwc_to_nchw: one block per pixel range, thread per pixel, channel loop; dab partial via per-thread
// channel loop + block reduction per channel (C small at module boundaries; used by standalone modules / tests)
__global__ void __launch_bounds__(256) nchw_grad_kernel(const float* dy, const addk_src s, int N, long HW, float* g, int ldg,
                                                        int accumulate, double* dab, int rows) {
  __shared__ double red[256][2];
  const long total = (long)N * HW;
  const int C = s.C;
  for (int c = 0; c < C; ++c) {
    float a = s.a ? s.a[c] : 1.f, b = s.b ? s.b[c] : 0.f;
    double sA = 0.0, sB = 0.0;
    for (long pp = (long)blockIdx.x * 256 + threadIdx.x; pp < total; pp += (long)rows * 256) {
      int n = (int)(pp / HW); long i = pp - (long)n * HW;
      float d = dy[((long)n * C + c) * HW + i];
      float x = s.x[pp * s.ld + c];
      if (s.relu && !(fmaf(a, x, b) > 0.f)) d = 0.f;
      sA += (double)d * (double)x; sB += (double)d;
      float* gp = g + pp * ldg + c;
      float gv = d * a;
      *gp = accumulate ? *gp + gv : gv;
    }
    if (dab) {
      red[threadIdx.x][0] = sA; red[threadIdx.x][1] = sB;
      __syncthreads();
      for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { red[threadIdx.x][0] += red[threadIdx.x + k][0]; red[threadIdx.x][1] += red[threadIdx.x + k][1]; }
        __syncthreads();
      }
      if (threadIdx.x == 0) { dab[((long)blockIdx.x * C + c) * 2] = red[0][0]; dab[((long)blockIdx.x * C + c) * 2 + 1] = red[0][1]; }
      __syncthreads();
    }
  }
}

int rows_for(long P, int C) {
  EwMap m = ew_map(C);
  long r = P / ((long)m.npl * 2);
  if (r < 1) r = 1;
  if (r > 1024) r = 1024;
  return (int)r;
}

}  // namespace

// Channel chunk c0.. of `src` (pointers advanced; a chunk is at most 1024 channels = 256 threads x 4)
static addk_src gap_chunk(const addk_src& s, int c0, int n) {
  addk_src c = s;
  c.x = s.x + c0; c.C = n;
  if (s.a) { c.a = s.a + c0; c.b = s.b + c0; }
  return c;
}

extern "C" int addk_gap_fwd(const addk_src* src, int32_t N, int32_t HW, float* y, int32_t ldy, float* ws, int32_t mean, void* stream) {
  ADDK_REQUIRE(src && src->x && y && ws && N > 0 && HW > 0 && src->C > 0 && ldy >= src->C, "gap_fwd: bad args");
  ADDK_REQUIRE((src->a == nullptr) == (src->b == nullptr), "gap_fwd: a/b must come together");
  hipStream_t st = (hipStream_t)stream;
  const int rows = rows_for(HW, src->C);
  for (int c0 = 0; c0 < src->C; c0 += 1024) {          // F=40 with a level-3 last stage has 1600 ASPP input channels
    const int n = src->C - c0 < 1024 ? src->C - c0 : 1024;
    GapK k{};
    k.src = gap_chunk(*src, c0, n); k.N = N; k.HW = HW; k.y = y; k.ldy = ldy; k.ws = ws; k.Cfull = src->C; k.c0 = c0;
    EwMap m = ew_map(n); k.nq = m.nq; k.npl = m.npl; k.vec = src_vec_ok(k.src);
    k.rows = rows;
    hipLaunchKernelGGL(gap_partial_kernel, dim3(k.rows, N), dim3(256), (size_t)m.nq * 4 * sizeof(float), st, k);
    int rc = addk_check_launch("gap_partial");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(gap_final_kernel, dim3(cdiv((long)N * src->C, 4)), dim3(256), 0, st, ws, N, rows, src->C, mean ? 1.f / (float)HW : 1.f, y, ldy);
  return addk_check_launch("gap_final");
}

extern "C" int addk_gap_bwd(const addk_src* src, int32_t N, int32_t HW, const float* dy, int32_t lddy, float* g, int32_t ldg,
                            int32_t accumulate, double* dab, void* stream) {
  ADDK_REQUIRE(src && src->x && dy && g && N > 0 && HW > 0 && src->C > 0 && ldg >= src->C && lddy >= src->C, "gap_bwd: bad args");
  const int rows = rows_for((long)N * HW, src->C);
  for (int c0 = 0; c0 < src->C; c0 += 1024) {
    const int n = src->C - c0 < 1024 ? src->C - c0 : 1024;
    GapK k{};
    k.src = gap_chunk(*src, c0, n); k.N = N; k.HW = HW; k.dy = dy + c0; k.lddy = lddy; k.g = g + c0; k.ldg = ldg; k.accumulate = accumulate;
    k.dab = (double*)dab; k.Cfull = src->C; k.c0 = c0;
    EwMap m = ew_map(n); k.nq = m.nq; k.npl = m.npl;
    k.vec = src_vec_ok(k.src) && aligned16(k.g) && ldg % 4 == 0;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(rows), dim3(256), (size_t)m.nq * 8 * sizeof(double), (hipStream_t)stream, k);
    int rc = addk_check_launch("gap_bwd");
    if (rc) return rc;
  }
  return 0;
}

static int pool_fill(PoolK& k, const addk_src* src, int N, int H, int W, int OH, int OW, int stride, int mode) {
  ADDK_REQUIRE(src && src->x && src->C > 0 && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && stride > 0 && (mode == 0 || mode == 1), "pool3: bad args");
  k.src = *src; k.N = N; k.H = H; k.W = W; k.OH = OH; k.OW = OW; k.stride = stride; k.mode = mode;
  return 0;
}

extern "C" int addk_pool3_fwd(const addk_src* src, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t stride, int32_t mode,
                              float* y, int32_t ldy, void* stream) {
  PoolK k{};
  int rc = pool_fill(k, src, N, H, W, OH, OW, stride, mode);
  if (rc) return rc;
  ADDK_REQUIRE(y && ldy >= src->C, "pool3_fwd: bad output");
  k.y = y; k.ldy = ldy;
  long total = (long)N * OH * OW * src->C; long b = cdiv(total, 256); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(pool3_fwd_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, k);
  return addk_check_launch("pool3_fwd");
}

extern "C" int addk_pool3_bwd(const addk_src* src, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t stride, int32_t mode,
                              const float* dy, int32_t lddy, float* g, int32_t ldg, int32_t accumulate, void* stream) {
  PoolK k{};
  int rc = pool_fill(k, src, N, H, W, OH, OW, stride, mode);
  if (rc) return rc;
  ADDK_REQUIRE(dy && g && lddy >= src->C && ldg >= src->C, "pool3_bwd: bad args");
  k.dy = dy; k.lddy = lddy; k.g = g; k.ldg = ldg; k.accumulate = accumulate;
  long total = (long)N * H * W * src->C; long b = cdiv(total, 256); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(pool3_bwd_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, k);
  return addk_check_launch("pool3_bwd");
}

extern "C" int addk_nchw_to_nhwc(const float* x, int32_t N, int32_t C, int64_t HW, float* y, int32_t ldy, void* stream) {
  ADDK_REQUIRE(x && y && N > 0 && C > 0 && HW > 0 && ldy >= C, "nchw_to_nhwc: bad args");
  long b = cdiv((long)N * HW, 256); if (b > 8192) b = 8192;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, N, C, (long)HW, y, ldy);
  return addk_check_launch("nchw_to_nhwc");
}

extern "C" int addk_nhwc_to_nchw(const addk_src* src, int32_t N, int64_t HW, float* y, void* stream) {
  ADDK_REQUIRE(src && src->x && y && N > 0 && HW > 0 && src->C > 0, "nhwc_to_nchw: bad args");
  long b = cdiv((long)N * HW, 256); if (b > 8192) b = 8192;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, *src, N, (long)HW, y);
  return addk_check_launch("nhwc_to_nchw");
}

extern "C" int addk_nchw_grad_to_nhwc(const float* dy, const addk_src* src, int32_t N, int64_t HW, float* g, int32_t ldg,
                                      int32_t accumulate, double* dab, void* stream) {
  ADDK_REQUIRE(dy && src && src->x && g && N > 0 && HW > 0 && src->C > 0 && ldg >= src->C, "nchw_grad_to_nhwc: bad args");
  int rows = rows_for((long)N * HW, src->C);
  hipLaunchKernelGGL(nchw_grad_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, *src, N, (long)HW, g, ldg, accumulate, (double*)dab, rows);
  return addk_check_launch("nchw_grad_to_nhwc");
}
