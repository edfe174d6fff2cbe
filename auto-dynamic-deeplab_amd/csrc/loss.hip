// Softmax cross-entropy on NCHW logits (nn.CrossEntropyLoss(weight, ignore_index=255), train.py:70,231),
// normalised Shannon entropy of the prediction (operations.py:161-170), argmax and the confusion matrix
// of the evaluator (utils/metrics.py:34-43).  One thread per pixel; channel planes are contiguous along
// W so every per-channel access of a wave is one coalesced 256-B segment.
#include "common.h"

namespace {


__device__ __forceinline__ float block_sum(float v, float* sh) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
  return s;
}

__global__ void __launch_bounds__(256) ce_count_kernel(const int64_t* target, long n, const float* cw, int ignore, int nc, float* ws) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long t = target[i];
    if (t != ignore && t >= 0 && t < nc) s += cw ? cw[t] : 1.f;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) sum_partials_kernel(const float* ws, int n, float scale, const float* denom, float* out, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += ws[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    float v = s * scale;
    if (denom) v /= *denom;
    *out = accumulate ? *out + v : v;
  }
}

// CC > 0: class count known at compile time, logits of a pixel live in registers (19 = Cityscapes);
// CC == 0: generic fallback that re-reads the (cache-resident) logits instead of indexing a register array.
template <int CC>
__global__ void __launch_bounds__(256) ce_kernel(const float* logits, const int64_t* target, int N, int C, long HW, const float* cw,
                                                int ignore, const float* wsum, float scale, float* dlogits, float* ws) {
  __shared__ float sh[4];
  const long total = (long)N * HW;
  const float inv = scale / *wsum;
  float lsum = 0.f;
  for (long pp = (long)blockIdx.x * 256 + threadIdx.x; pp < total; pp += (long)gridDim.x * 256) {
    int n = (int)(pp / HW); long i = pp - (long)n * HW;
    const float* lp = logits + (long)n * C * HW + i;
    long t = target[pp];
    const bool valid = t != ignore && t >= 0 && t < C;
    const float w = valid ? (cw ? cw[t] : 1.f) : 0.f;
    float* dp = dlogits ? dlogits + (long)n * C * HW + i : nullptr;
    if (CC > 0) {
      float v[CC > 0 ? CC : 1];
      float mx = -INFINITY, lt = 0.f;
#pragma unroll
      for (int c = 0; c < CC; ++c) { v[c] = lp[(long)c * HW]; mx = fmaxf(mx, v[c]); if (c == t) lt = v[c]; }
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < CC; ++c) { v[c] = expf(v[c] - mx); se += v[c]; }
      if (valid) lsum += w * (logf(se) + mx - lt);
      if (dp) {
        float k = w * inv / se;
#pragma unroll
        for (int c = 0; c < CC; ++c) dp[(long)c * HW] = v[c] * k - ((valid && c == t) ? w * inv : 0.f);
      }
    } else {
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(long)c * HW]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(lp[(long)c * HW] - mx);
      if (valid) lsum += w * (logf(se) + mx - lp[t * HW]);
      if (dp) {
        float k = w * inv / se;
        for (int c = 0; c < C; ++c) dp[(long)c * HW] = expf(lp[(long)c * HW] - mx) * k - ((valid && c == t) ? w * inv : 0.f);
      }
    }
  }
  lsum = block_sum(lsum, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = lsum;
}

__global__ void __launch_bounds__(256) entropy_kernel(const float* logits, int N, int C, long HW, float* ws) {
  __shared__ float sh[4];
  const long total = (long)N * HW;
  float s = 0.f;
  for (long pp = (long)blockIdx.x * 256 + threadIdx.x; pp < total; pp += (long)gridDim.x * 256) {
    int n = (int)(pp / HW); long i = pp - (long)n * HW;
    const float* lp = logits + (long)n * C * HW + i;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(long)c * HW]);
    float se = 0.f, sx = 0.f;
    for (int c = 0; c < C; ++c) { float d = lp[(long)c * HW] - mx; float e = expf(d); se += e; sx += e * d; }
    // -sum p log p = log(se) - sx/se
    s += logf(se) - sx / se;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ void argmax_kernel(const float* logits, int N, int C, long HW, int64_t* out) {
  const long total = (long)N * HW;
  for (long pp = (long)blockIdx.x * blockDim.x + threadIdx.x; pp < total; pp += (long)gridDim.x * blockDim.x) {
    int n = (int)(pp / HW); long i = pp - (long)n * HW;
    const float* lp = logits + (long)n * C * HW + i;
    float best = lp[0]; int bi = 0;
    for (int c = 1; c < C; ++c) { float v = lp[(long)c * HW]; if (v > best) { best = v; bi = c; } }
    out[pp] = bi;
  }
}

__global__ void confusion_kernel(const int64_t* gt, const int64_t* pred, long n, int nc, unsigned long long* cm) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long g = gt[i], p = pred[i];
    if (g >= 0 && g < nc && p >= 0 && p < nc) atomicAdd(&cm[g * nc + p], 1ULL);
  }
}


// ---- fused logits up-sampling + cross-entropy, forward and backward (decoder.py:28 + train.py:70,231) -------------------
// The training step never needs the full-resolution logits: loss and gradient are computed straight from the low-resolution
// NHWC logits.  F.interpolate(mode='bilinear', align_corners=False) index arithmetic as in resize.hip (ATen's
// area_pixel_compute_source_index in fp32).  A high-resolution pixel (Y, X) reads the low-resolution rows h0(Y), h0+1 and
// columns w0(X), w0+1; it is OWNED by (h0, w0): thread (r, xl) of a block walks the pixels of row Y = first_row(h) + r whose
// w0 is its column, computes each pixel's softmax ONCE and accumulates the pixel's gradient into A0 (column w0) and A1
// (column w0 + 1).  A1 moves to the right neighbour lane, the 16 rows of the band are reduced through LDS in a fixed order
// with the row weights, and the part that belongs to row h + 1 is carried in registers to the next band: a gather with no
// atomics and no second pass over the 1.3 GB/exit of full-resolution logits and gradients the three-kernel form wrote and re-read.
// A block owns 31 output columns x HB output rows; it re-walks one band above (for the carry) and one column to the left
// (for A1): (HB+1)/HB * 32/31 redundant softmax work, counted once in the loss (own bands, xl >= 1).
struct CeUpK {
  const float* x; int ld; int N, H, W, OH, OW;
  const int64_t* target; const float* cw; int ignore;
  const float* wsum; float scale;
  float* g; int ldg; int accumulate;
  float* ws; int HB;
};

__device__ __forceinline__ void ce_src_index(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}
__host__ __device__ __forceinline__ int ce_idx0(int dst, float scale, int in) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  int i0 = (int)s;
  return i0 > in - 1 ? in - 1 : i0;
}
// smallest output index whose i0 is >= i (i0 is monotone in the output index)
__host__ __device__ __forceinline__ int ce_first_out(int i, float scale, int in, int out) {
  if (i <= 0) return 0;
  int g = (int)(((float)i + 0.5f) / scale - 0.5f);
  if (g < 0) g = 0;
  if (g > out) g = out;
  while (g > 0 && ce_idx0(g - 1, scale, in) >= i) --g;
  while (g < out && ce_idx0(g, scale, in) < i) ++g;
  return g;
}

constexpr int CEU_R = 8, CEU_X = 32;       // rows per pass / lanes per row (31 output columns + the left neighbour)
constexpr int CEU_MAXBAND = 16;            // high-resolution rows per low-resolution row the kernel takes (host check)

template <int CC>
__global__ void __launch_bounds__(256) ce_up_kernel(const CeUpK p) {
  constexpr int CP = (CC + 3) / 4 * 4;     // channels incl. the padding of a 16-byte-aligned pixel row
  constexpr int NOUT = ((CEU_X - 1) * CC + 255) / 256;
  __shared__ float S0[CEU_R * CEU_X * CC], S1[CEU_R * CEU_X * CC];
  __shared__ float shs[4];
  const int t = threadIdx.x, xl = t & (CEU_X - 1), r = t / CEU_X;
  const int tx0 = blockIdx.x * (CEU_X - 1), hb = blockIdx.y * p.HB, n = blockIdx.z;
  const int x = tx0 - 1 + xl;                                   // the low-resolution column this thread owns as w0
  const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
  const float inv = p.scale / *(const gfloat*)p.wsum;
  int xlo = 0, xhi = 0;
  if (x >= 0 && x < p.W) { xlo = ce_first_out(x, sw, p.W, p.OW); xhi = (x + 1 < p.W) ? ce_first_out(x + 1, sw, p.W, p.OW) : p.OW; }
  const bool xlast = x == p.W - 1;
  const int x1 = x + (x < p.W - 1 ? 1 : 0);
  const bool vec = (p.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && p.ld >= CP;
  const int64_t __attribute__((address_space(1)))* tgt = (const int64_t __attribute__((address_space(1)))*)p.target;
  const gfloat* cw = (const gfloat*)p.cw;
  auto load_px = [&](const float* q, float (&v)[CP]) {
    if (vec) {
#pragma unroll
      for (int c = 0; c < CP; c += 4) { const float4 f = ld4(q + c); v[c] = f.x; v[c + 1] = f.y; v[c + 2] = f.z; v[c + 3] = f.w; }
    } else {
#pragma unroll
      for (int c = 0; c < CC; ++c) v[c] = ((const gfloat*)q)[c];
    }
  };
  float lsum = 0.f;
  float carry[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; ++k) carry[k] = 0.f;
  const int hstart = hb > 0 ? hb - 1 : 0;
  const int hend = hb + p.HB < p.H ? hb + p.HB : p.H;
  for (int h = hstart; h < hend; ++h) {
    const int ylo = ce_first_out(h, sh, p.H, p.OH);
    const int yhi = (h + 1 < p.H) ? ce_first_out(h + 1, sh, p.H, p.OH) : p.OH;
    const bool own = h >= hb;
    float b0[NOUT], b1[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) { b0[k] = 0.f; b1[k] = 0.f; }
    for (int yb = ylo; yb < yhi; yb += CEU_R) {                 // the band in passes of CEU_R rows (one pass at the x8 of config 2)
      const int Y = yb + r;
      float A0[CC], A1[CC];
#pragma unroll
      for (int c = 0; c < CC; ++c) { A0[c] = 0.f; A1[c] = 0.f; }
      float lh0 = 0.f, lh1 = 0.f;
      if (Y < yhi && xhi > xlo) {
        int h0, h1;
        ce_src_index(Y, sh, p.H, h0, h1, lh0, lh1);
        const float* r0 = p.x + ((long)(n * p.H + h0) * p.W) * p.ld;
        const float* r1 = p.x + ((long)(n * p.H + h1) * p.W) * p.ld;
        float v00[CP], v01[CP], v10[CP], v11[CP];
        load_px(r0 + (long)x * p.ld, v00); load_px(r0 + (long)x1 * p.ld, v01);
        load_px(r1 + (long)x * p.ld, v10); load_px(r1 + (long)x1 * p.ld, v11);
        const int64_t __attribute__((address_space(1)))* tp = tgt + ((long)n * p.OH + Y) * p.OW;
        for (int X = xlo; X < xhi; ++X) {
          int w0, w1; float lw0, lw1;
          ce_src_index(X, sw, p.W, w0, w1, lw0, lw1);
          const long tg = tp[X];
          const bool valid = tg != p.ignore && tg >= 0 && tg < CC;
          const float w = valid ? (cw ? cw[tg] : 1.f) : 0.f;
          float z[CC];
          float mx = -INFINITY, zt = 0.f;
#pragma unroll
          for (int c = 0; c < CC; ++c) {
            z[c] = lh0 * (lw0 * v00[c] + lw1 * v01[c]) + lh1 * (lw0 * v10[c] + lw1 * v11[c]);
            mx = fmaxf(mx, z[c]);
            if (c == tg) zt = z[c];
          }
          float se = 0.f;
#pragma unroll
          for (int c = 0; c < CC; ++c) { z[c] = __expf(z[c] - mx); se += z[c]; }
          if (own && xl >= 1 && valid) lsum += w * (logf(se) + mx - zt);
          const float k = w * inv / se, kt = w * inv;
          const float a0 = lw0 + (xlast ? lw1 : 0.f), a1 = xlast ? 0.f : lw1;
#pragma unroll
          for (int c = 0; c < CC; ++c) {
            const float gz = z[c] * k - ((valid && c == tg) ? kt : 0.f);
            A0[c] = fmaf(a0, gz, A0[c]); A1[c] = fmaf(a1, gz, A1[c]);
          }
        }
      }
      // column w0 + 1 belongs to the right neighbour lane; then the rows of the pass go through LDS
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        float fromleft = __shfl_up(A1[c], 1, CEU_X);
        if (xl == 0) fromleft = 0.f;
        const float R = A0[c] + fromleft;
        S0[(r * CEU_X + xl) * CC + c] = lh0 * R;
        S1[(r * CEU_X + xl) * CC + c] = lh1 * R;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const int j = t + 256 * k;
        if (j < (CEU_X - 1) * CC) {
          const int off = CC + j;                               // (ox = 1 + j / CC, c = j % CC) -> ox * CC + c
#pragma unroll
          for (int rr = 0; rr < CEU_R; ++rr) { b0[k] += S0[rr * CEU_X * CC + off]; b1[k] += S1[rr * CEU_X * CC + off]; }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      const int j = t + 256 * k;
      if (j < (CEU_X - 1) * CC) {
        const int ox = 1 + j / CC, c = j - (ox - 1) * CC;
        if (h == p.H - 1) { b0[k] += b1[k]; b1[k] = 0.f; }
        const float o = carry[k] + b0[k];
        carry[k] = b1[k];
        const int col = tx0 - 1 + ox;
        if (own && col < p.W) {
          gfloat* gp = (gfloat*)p.g + ((long)(n * p.H + h) * p.W + col) * p.ldg + c;
          *gp = p.accumulate ? *gp + o : o;
        }
      }
    }
  }
  lsum = block_sum(lsum, shs);
  if (t == 0) ((gfloat*)p.ws)[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = lsum;
}

int ceu_hb(int H) { return H >= 64 ? 4 : (H >= 16 ? 2 : 1); }
bool ceu_ok(int N, int H, int W, int OH, int OW, int C) {
  if (C != 19 || N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return false;
  const float sh = (float)H / (float)OH;
  for (int h = 0; h < H; ++h) {
    const int lo = ce_first_out(h, sh, H, OH), hi = h + 1 < H ? ce_first_out(h + 1, sh, H, OH) : OH;
    if (hi - lo > CEU_MAXBAND) return false;
  }
  return true;
}

int ce_blocks(long total) { long b = cdiv(total, 256 * 4); if (b < 1) b = 1; if (b > 1024) b = 1024; return (int)b; }

}  // namespace

extern "C" int64_t addk_ce_ws_floats(int32_t N, int64_t HW) { (void)N; (void)HW; return 1024; }

extern "C" int addk_ce_count(const int64_t* target, int64_t n, const float* class_w, int32_t ignore_index, int32_t num_classes,
                             float* wsum, float* ws, void* stream) {
  ADDK_REQUIRE(target && wsum && ws && n > 0 && num_classes > 0, "ce_count: bad args");
  hipStream_t st = (hipStream_t)stream;
  int b = ce_blocks(n);
  hipLaunchKernelGGL(ce_count_kernel, dim3(b), dim3(256), 0, st, target, (long)n, class_w, ignore_index, num_classes, ws);
  int rc = addk_check_launch("ce_count");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, b, 1.f, (const float*)nullptr, wsum, 0);
  return addk_check_launch("ce_count_sum");
}

extern "C" int addk_ce_fwd_bwd(const float* logits, const int64_t* target, int32_t N, int32_t C, int64_t HW, const float* class_w,
                               int32_t ignore_index, const float* wsum, float scale, float* loss_out, float* dlogits, float* ws,
                               void* stream) {
  ADDK_REQUIRE(logits && target && wsum && loss_out && ws && N > 0 && C > 0 && HW > 0, "ce_fwd_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  int b = ce_blocks((long)N * HW);
  if (C == 19)
    hipLaunchKernelGGL(ce_kernel<19>, dim3(b), dim3(256), 0, st, logits, target, N, C, (long)HW, class_w, ignore_index, wsum, scale, dlogits, ws);
  else
    hipLaunchKernelGGL(ce_kernel<0>, dim3(b), dim3(256), 0, st, logits, target, N, C, (long)HW, class_w, ignore_index, wsum, scale, dlogits, ws);
  int rc = addk_check_launch("ce");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, b, scale, wsum, loss_out, 1);
  return addk_check_launch("ce_sum");
}


extern "C" int addk_ce_upsample_supported(int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t C) {
  return ceu_ok(N, H, W, OH, OW, C) ? 1 : 0;
}
extern "C" int64_t addk_ce_upsample_ws_floats(int32_t N, int32_t H, int32_t W) {
  return (int64_t)N * cdiv(H, ceu_hb(H)) * cdiv(W, CEU_X - 1);
}
extern "C" int addk_ce_upsample_fwd_bwd(const addk_ce_upsample_args* a, void* stream) {
  ADDK_REQUIRE(a && a->logits && a->target && a->wsum && a->loss_out && a->g && a->ws, "ce_upsample: null pointer");
  ADDK_REQUIRE(a->ld >= a->C && a->ldg >= a->C, "ce_upsample: short stride");
  ADDK_REQUIRE(ceu_ok(a->N, a->H, a->W, a->OH, a->OW, a->C), "ce_upsample: unsupported shape (19 classes, at most %d output rows per input row)", CEU_MAXBAND);
  CeUpK k;
  k.x = a->logits; k.ld = a->ld; k.N = a->N; k.H = a->H; k.W = a->W; k.OH = a->OH; k.OW = a->OW;
  k.target = a->target; k.cw = a->class_w; k.ignore = a->ignore_index; k.wsum = a->wsum; k.scale = a->scale;
  k.g = a->g; k.ldg = a->ldg; k.accumulate = a->accumulate; k.ws = a->ws; k.HB = ceu_hb(a->H);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv(a->W, CEU_X - 1), cdiv(a->H, k.HB), a->N);
  hipLaunchKernelGGL(ce_up_kernel<19>, grid, dim3(256), 0, st, k);
  int rc = addk_check_launch("ce_upsample");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, a->ws, (int)(grid.x * grid.y * grid.z), a->scale, a->wsum, a->loss_out, 1);
  return addk_check_launch("ce_upsample_sum");
}

extern "C" int addk_entropy_sum(const float* logits, int32_t N, int32_t C, int64_t HW, float* out1, float* ws, void* stream) {
  ADDK_REQUIRE(logits && out1 && ws && N > 0 && C > 0 && HW > 0, "entropy_sum: bad args");
  hipStream_t st = (hipStream_t)stream;
  int b = ce_blocks((long)N * HW);
  hipLaunchKernelGGL(entropy_kernel, dim3(b), dim3(256), 0, st, logits, N, C, (long)HW, ws);
  int rc = addk_check_launch("entropy");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, b, 1.f, (const float*)nullptr, out1, 0);
  return addk_check_launch("entropy_sum");
}

extern "C" int addk_argmax_nchw(const float* logits, int32_t N, int32_t C, int64_t HW, int64_t* out, void* stream) {
  ADDK_REQUIRE(logits && out && N > 0 && C > 0 && HW > 0, "argmax: bad args");
  long b = cdiv((long)N * HW, 256); if (b > 8192) b = 8192;
  hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, logits, N, C, (long)HW, out);
  return addk_check_launch("argmax");
}

extern "C" int addk_confusion(const int64_t* gt, const int64_t* pred, int64_t n, int32_t num_class, int64_t* cm, void* stream) {
  ADDK_REQUIRE(gt && pred && cm && n > 0 && num_class > 0, "confusion: bad args");
  long b = cdiv(n, 256); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, gt, pred, (long)n, num_class,
                     reinterpret_cast<unsigned long long*>(cm));
  return addk_check_launch("confusion");
}
