// Probe: is the accumulation inside the bf16 MFMA biased?  Signed error statistics of C = A*B (K = 2304) against fp64 for
// the exact-fp32 MFMA chain and the 6-term split-bf16 form on v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16.
// A rounding bias shows as |mean(err)| comparable to rms(err); an unbiased rounding has |mean| ~ rms / sqrt(samples).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}
// C[32][32] = A[32][K] * B[K][32]; mode 0: fp32 16x16x4 (4 tiles), 1: bf16x6 32x32x16, 2: bf16x6 16x16x32 (4 tiles), 3: bf16x6 32x32x16 big terms first,
// 4: bf16x6 32x32x16 with the three small terms summed among themselves first and joined by rounded fp32 adds (what conv3b_kernel does)
__global__ void k(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x;
  if (mode == 1 || mode == 3 || mode == 4) {
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int j = 0; j < 8; ++j) {
        __bf16 x, y, z;
        split3(A[r * K + k0 + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
        split3(B[(k0 + 8 * h + j) * 32 + r], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
      }
      if (mode == 4) {
        f32x16 z; for (int e = 0; e < 16; ++e) z[e] = 0.f;
        f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, z, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, t, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) acc[e] += t[e];
      } else if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      }
    }
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
  } else {
    const int r = lane & 15, q = lane >> 4;
    for (int ti = 0; ti < 2; ++ti) for (int tj = 0; tj < 2; ++tj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (mode == 0) {
        for (int kk = 0; kk < K; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(ti * 16 + r) * K + kk + q], B[(kk + q) * 32 + tj * 16 + r], acc, 0, 0, 0);
      } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
          bf16x8 ah, am, al, bh, bm, bl;
          for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(A[(ti * 16 + r) * K + k0 + 8 * q + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
            split3(B[(k0 + 8 * q + j) * 32 + tj * 16 + r], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
      }
      for (int e = 0; e < 4; ++e) C[(ti * 16 + q * 4 + e) * 32 + tj * 16 + r] = acc[e];
    }
  }
}
int main() {
  const int K = 2304, R = 16;        // R independent problems -> 16384 samples
  float *dA, *dB, *dC;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dC, 1024 * 4);
  const char* names[5] = {"fp32 mfma 16x16x4", "bf16x6 32x32x16 small-first", "bf16x6 16x16x32 small-first", "bf16x6 32x32x16 big-first", "bf16x6 small terms apart"};
  for (int sgn = 0; sgn < 2; ++sgn) {
    printf(sgn ? "--- all-positive operands (no cancellation)\n" : "--- signed operands\n");
    for (int mode = 0; mode < 5; ++mode) {
      double se = 0, se2 = 0, sref = 0; long n = 0;
      srand(7);
      for (int rep = 0; rep < R; ++rep) {
        std::vector<float> A(32 * K), B(K * 32), C(1024);
        for (auto& v : A) { v = (float)(rand() / (double)RAND_MAX * 2 - 1); if (sgn) v = fabsf(v); }
        for (auto& v : B) { v = (float)(rand() / (double)RAND_MAX * 2 - 1) * 0.1f; if (sgn) v = fabsf(v); }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
        hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double s = 0; for (int kk = 0; kk < K; ++kk) s += (double)A[i * K + kk] * B[kk * 32 + j];
          const double e = C[i * 32 + j] - s; se += e; se2 += e * e; sref += fabs(s); ++n;
        }
      }
      const double mean = se / n, rms = sqrt(se2 / n);
      printf("%-30s mean err %+.3e  rms err %.3e  mean/rms %+.3f  (expected |mean/rms| ~ %.3f if unbiased)  rms/mean|ref| %.2e\n", names[mode], mean, rms, mean / rms, 1.0 / sqrt((double)n), rms / (sref / n));
    }
  }
  return 0;
}
