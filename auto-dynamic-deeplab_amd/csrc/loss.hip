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
