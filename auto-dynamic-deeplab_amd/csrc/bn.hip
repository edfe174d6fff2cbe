// BatchNorm statistics kernels (F.batch_norm training mode at every BatchNorm(...) call site of
// the path: operations.py:25,39,54,58,93; ADD.py:156,162,168,258; aspp_train.py:27-32; decoder.py:15,19).
// They only touch [rows][C][2] partial slabs written by the producing kernels and C-length vectors.
// Slabs are fp64 end to end (squares, partial sums, cross-block sums): the E[x^2]-E[x]^2 form must survive
// the 2-sample BatchNorm of the ASPP image-pool branch at bs=2 as ATen's fp64-accumulating CPU path does.
#include "common.h"

namespace {

// block = BN_CH channels x BN_RG row groups (1024 threads); result (sum0, sum1) per channel for threads with rg == 0.
// 16 x 64: a 40-channel slab of ~1000 rows is three blocks of two 8-deep batches of independent 16-byte loads per thread
// (the 32 x 32 split it replaces walked 31 rows per thread in 8 dependent batches — these kernels are pure latency).
constexpr int BN_CH = 16, BN_RG = 64;
__device__ __forceinline__ void slab_sum(const double* slab, int rows, int C, int c, int rg, double& s0, double& s1,
                                         double (*sh)[BN_CH][2]) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    int r = rg;
    for (; r + 7 * BN_RG < rows; r += 8 * BN_RG) {
      double2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2*>(slab + ((long)(r + k * BN_RG) * C + c) * 2);
      a += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
      b += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
    }
    double2 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int rr = r + k * BN_RG;
      const bool ok = rr < rows;
      const double2 t = *reinterpret_cast<const double2*>(slab + ((long)(ok ? rr : 0) * C + c) * 2);
      w[k].x = ok ? t.x : 0.0; w[k].y = ok ? t.y : 0.0;
    }
    a += ((w[0].x + w[1].x) + (w[2].x + w[3].x)) + ((w[4].x + w[5].x) + (w[6].x + w[7].x));
    b += ((w[0].y + w[1].y) + (w[2].y + w[3].y)) + ((w[4].y + w[5].y) + (w[6].y + w[7].y));
  }
  const int cl = threadIdx.x % BN_CH;
  sh[rg][cl][0] = a; sh[rg][cl][1] = b;
  __syncthreads();
  for (int s = BN_RG / 2; s > 0; s >>= 1) {
    if (rg < s) { sh[rg][cl][0] += sh[rg + s][cl][0]; sh[rg][cl][1] += sh[rg + s][cl][1]; }
    __syncthreads();
  }
  s0 = sh[0][cl][0]; s1 = sh[0][cl][1];
  __syncthreads();
}

__device__ __forceinline__ void bn_finalize_body(const addk_bn_finalize_args& p, double (*sh)[BN_CH][2]) {
  const int cl = threadIdx.x % BN_CH, rg = threadIdx.x / BN_CH;
  const int c = blockIdx.x * BN_CH + cl;
  double s0, s1;
  slab_sum((const double*)p.partial, p.rows, p.C, c, rg, s0, s1, sh);
  if (rg == 0 && c < p.C) {
    double mean = s0 / p.count;
    double var = s1 / p.count - mean * mean;
    if (var < 0.0) var = 0.0;
    double invstd = 1.0 / sqrt(var + (double)p.eps);
    float g = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
    float a = (float)(g * invstd);
    p.a[c] = a;
    p.b[c] = (float)(be - mean * (g * invstd));
    if (p.mean) p.mean[c] = (float)mean;
    if (p.invstd) p.invstd[c] = (float)invstd;
    if (p.running_mean) {
      double unb = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
      p.running_mean[c] = (float)((1.0 - p.momentum) * p.running_mean[c] + p.momentum * mean);
      p.running_var[c] = (float)((1.0 - p.momentum) * p.running_var[c] + p.momentum * unb);
    }
  }
}
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const addk_bn_finalize_args p) {
  __shared__ double sh[BN_RG][BN_CH][2];
  bn_finalize_body(p, sh);
}
// several independent BatchNorms in one launch: block (x, y) = channel block x of table entry y
__global__ void __launch_bounds__(1024) bn_finalize_batch_kernel(const addk_bn_finalize_args* __restrict__ tab) {
  __shared__ double sh[BN_RG][BN_CH][2];
  const addk_bn_finalize_args p = tab[blockIdx.y];
  if (blockIdx.x * BN_CH >= p.C) return;
  bn_finalize_body(p, sh);
}

__global__ void __launch_bounds__(1024) slab_reduce_kernel(const double* slab, int rows, int C, double* out) {
  __shared__ double sh[BN_RG][BN_CH][2];
  const int cl = threadIdx.x % BN_CH, rg = threadIdx.x / BN_CH;
  const int c = blockIdx.x * BN_CH + cl;
  double s0, s1;
  slab_sum(slab, rows, C, c, rg, s0, s1, sh);
  if (rg == 0 && c < C) { out[2 * c] = s0; out[2 * c + 1] = s1; }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* a, float* b) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    float s = g / sqrtf(rv[c] + eps);
    a[c] = s; b[c] = be - rm[c] * s;
  }
}

// eval mode, all BatchNorms of a plan in one launch: block b handles table entry b
struct BnEvalEntry { const float* gamma; const float* beta; const float* rm; const float* rv; float* a; float* b; int C; float eps; };
__global__ void bn_eval_affine_batch_kernel(const BnEvalEntry* __restrict__ tab) {
  const BnEvalEntry e = tab[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    float g = e.gamma ? e.gamma[c] : 1.f, be = e.beta ? e.beta[c] : 0.f;
    float s = g / sqrtf(e.rv[c] + e.eps);
    e.a[c] = s; e.b[c] = be - e.rm[c] * s;
  }
}

__device__ __forceinline__ void bn_bwd_body(const addk_bn_bwd_args& p, double (*sh)[BN_CH][2]) {
  const int cl = threadIdx.x % BN_CH, rg = threadIdx.x / BN_CH;
  const int c = blockIdx.x * BN_CH + cl;
  double dA = 0.0, dB = 0.0;
  for (int k = 0; k < p.nslab; ++k) {
    double s0, s1;
    slab_sum((const double*)p.slab[k], p.rows[k], p.C, c, rg, s0, s1, sh);
    dA += s0; dB += s1;
  }
  if (rg == 0 && c < p.C) {
    double mean = p.mean[c], invstd = p.invstd[c], gamma = p.gamma ? p.gamma[c] : 1.0, a = p.a[c];
    double t = dA - mean * dB;
    double dgamma = invstd * t, dbeta = dB;
    double dvar = -0.5 * gamma * t * invstd * invstd * invstd;
    double dmean_tot = -a * dB - (p.centered ? 0.0 : 2.0 * mean * dvar);
    if (p.dgamma) p.dgamma[c] = (float)((p.accumulate ? (double)p.dgamma[c] : 0.0) + dgamma);
    if (p.dbeta) p.dbeta[c] = (float)((p.accumulate ? (double)p.dbeta[c] : 0.0) + dbeta);
    if (p.dmv) { p.dmv[2 * c] = (float)dmean_tot; p.dmv[2 * c + 1] = (float)dvar; }
    if (p.c1) { p.c1[c] = (float)(dmean_tot / p.count); p.c2[c] = (float)(2.0 * dvar / p.count); }
  }
}
__global__ void __launch_bounds__(1024) bn_bwd_kernel(const addk_bn_bwd_args p) {
  __shared__ double sh[BN_RG][BN_CH][2];
  bn_bwd_body(p, sh);
}
__global__ void __launch_bounds__(1024) bn_bwd_batch_kernel(const addk_bn_bwd_args* __restrict__ tab) {
  __shared__ double sh[BN_RG][BN_CH][2];
  const addk_bn_bwd_args& p = tab[blockIdx.y];
  if (blockIdx.x * BN_CH >= p.C) return;
  bn_bwd_body(p, sh);
}

__global__ void __launch_bounds__(1024) slab_reduce_batch_kernel(const addk_slab_reduce_item* __restrict__ tab) {
  __shared__ double sh[BN_RG][BN_CH][2];
  const addk_slab_reduce_item it = tab[blockIdx.y];
  if (blockIdx.x * BN_CH >= it.C) return;
  const int cl = threadIdx.x % BN_CH, rg = threadIdx.x / BN_CH;
  const int c = blockIdx.x * BN_CH + cl;
  double s0, s1;
  slab_sum(it.partial, it.rows, it.C, c, rg, s0, s1, sh);
  if (rg == 0 && c < it.C) { it.out[2 * c] = s0; it.out[2 * c + 1] = s1; }
}
__global__ void bn_coeffs_batch_kernel(const addk_bn_coeffs_item* __restrict__ tab) {
  const addk_bn_coeffs_item it = tab[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < it.C) { it.c1[c] = (float)((double)it.dmv[2 * c] / it.count); it.c2[c] = (float)(2.0 * (double)it.dmv[2 * c + 1] / it.count); }
}

__global__ void bn_coeffs_kernel(const float* dmv, int C, double count, float* c1, float* c2) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { c1[c] = (float)((double)dmv[2 * c] / count); c2[c] = (float)(2.0 * (double)dmv[2 * c + 1] / count); }
}

}  // namespace

extern "C" int addk_bn_finalize(const addk_bn_finalize_args* a, void* stream) {
  ADDK_REQUIRE(a && a->partial && a->a && a->b && a->C > 0 && a->rows > 0 && a->count > 0, "bn_finalize: bad args");
  ADDK_REQUIRE((a->running_mean == nullptr) == (a->running_var == nullptr), "bn_finalize: running stats come together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(a->C, BN_CH)), dim3(1024), 0, (hipStream_t)stream, *a);
  return addk_check_launch("bn_finalize");
}

extern "C" int addk_slab_reduce(const double* partial, int32_t rows, int32_t C, double* out, void* stream) {
  ADDK_REQUIRE(partial && out && rows > 0 && C > 0, "slab_reduce: bad args");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(C, BN_CH)), dim3(1024), 0, (hipStream_t)stream, partial, rows, C, out);
  return addk_check_launch("slab_reduce");
}

extern "C" int addk_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                   int32_t C, float* a, float* b, void* stream) {
  ADDK_REQUIRE(rm && rv && a && b && C > 0, "bn_eval_affine: bad args");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, C, a, b);
  return addk_check_launch("bn_eval_affine");
}

extern "C" int addk_bn_bwd(const addk_bn_bwd_args* a, void* stream) {
  ADDK_REQUIRE(a && a->nslab >= 0 && a->nslab <= ADDK_MAX_SLAB && a->C > 0 && a->count > 0, "bn_bwd: bad args");
  ADDK_REQUIRE(a->mean && a->invstd && a->a, "bn_bwd: saved statistics missing");
  ADDK_REQUIRE((a->c1 == nullptr) == (a->c2 == nullptr) && (a->c1 || a->dmv), "bn_bwd: need c1/c2 or dmv");
  for (int i = 0; i < a->nslab; ++i) ADDK_REQUIRE(a->slab[i] && a->rows[i] > 0, "bn_bwd: bad slab %d", i);
  hipLaunchKernelGGL(bn_bwd_kernel, dim3(cdiv(a->C, BN_CH)), dim3(1024), 0, (hipStream_t)stream, *a);
  return addk_check_launch("bn_bwd");
}

extern "C" int addk_bn_bwd_coeffs_from_dmv(const float* dmv, int32_t C, double count, float* c1, float* c2, void* stream) {
  ADDK_REQUIRE(dmv && c1 && c2 && C > 0 && count > 0, "bn_bwd_coeffs: bad args");
  hipLaunchKernelGGL(bn_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dmv, C, count, c1, c2);
  return addk_check_launch("bn_bwd_coeffs");
}

extern "C" int addk_bn_eval_affine_batch(const void* dev_table, int32_t n, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0, "bn_eval_affine_batch: bad args");
  hipLaunchKernelGGL(bn_eval_affine_batch_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const BnEvalEntry*>(dev_table));
  return addk_check_launch("bn_eval_affine_batch");
}

extern "C" int addk_bn_finalize_batch(const addk_bn_finalize_args* dev_table, int32_t n, int32_t max_C, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0 && max_C > 0, "bn_finalize_batch: bad args");
  hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3(cdiv(max_C, BN_CH), n), dim3(1024), 0, (hipStream_t)stream, dev_table);
  return addk_check_launch("bn_finalize_batch");
}
extern "C" int addk_bn_bwd_batch(const addk_bn_bwd_args* dev_table, int32_t n, int32_t max_C, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0 && max_C > 0, "bn_bwd_batch: bad args");
  hipLaunchKernelGGL(bn_bwd_batch_kernel, dim3(cdiv(max_C, BN_CH), n), dim3(1024), 0, (hipStream_t)stream, dev_table);
  return addk_check_launch("bn_bwd_batch");
}

extern "C" int addk_slab_reduce_batch(const addk_slab_reduce_item* dev_table, int32_t n, int32_t max_C, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0 && max_C > 0, "slab_reduce_batch: bad args");
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3(cdiv(max_C, BN_CH), n), dim3(1024), 0, (hipStream_t)stream, dev_table);
  return addk_check_launch("slab_reduce_batch");
}
extern "C" int addk_bn_bwd_coeffs_batch(const addk_bn_coeffs_item* dev_table, int32_t n, int32_t max_C, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0 && max_C > 0, "bn_bwd_coeffs_batch: bad args");
  hipLaunchKernelGGL(bn_coeffs_batch_kernel, dim3(cdiv(max_C, 256), n), dim3(256), 0, (hipStream_t)stream, dev_table);
  return addk_check_launch("bn_bwd_coeffs_batch");
}
