// Probe: fp32 products evaluated as sums of bf16 x bf16 MFMA products (x = h + m + l, three bf16 terms = 24 significand
// bits).  Measures (1) accuracy of 3-term (hh, hm, mh) and 6-term (hh, hm, mh, mm, hl, lh) evaluation on
// v_mfma_f32_16x16x32_bf16 against an fp64 reference, next to the exact-fp32 v_mfma_f32_16x16x4_f32 chain, and
// (2) the issue rate of each form from registers (one wave per SIMD, every CU).
//   hipcc --offload-arch=gfx950 -O3 scripts/bf16_split_probe.hip -o /tmp/bf16_split_probe && /tmp/bf16_split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h;
  m = (__bf16)r; r = r - (float)m;
  l = (__bf16)r;
}

// one wave: C[16x16] = A[16xK] * B[Kx16]; mode 0 fp32 mfma, 3 / 6 = number of bf16 product terms
__global__ void gemm_tile(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int k = 0; k < K; k += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int j = 0; j < 8; ++j) {
        __bf16 h, m, l;
        split3(A[r * K + k0 + 8 * q + j], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
        split3(B[(k0 + 8 * q + j) * 16 + r], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
      }
      if (mode == 6) {      // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int e = 0; e < 4; ++e) C[(q * 4 + e) * 16 + r] = acc[e];
}

// register-only issue-rate loops: 4 waves per block (one per SIMD), each 16 accumulator tiles
template <int MODE>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float s = (float)threadIdx.x * 1e-3f + 1.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(s + j); b[j] = (__bf16)(s - j); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k)          // K = 32 as eight 16x16x4 fp32 steps
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(s + k, s - i, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < MODE; ++k)       // K = 32 as MODE bf16 product terms
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
  const int K = 2048;
  std::vector<float> A(16 * K), B(K * 16), C(256);
  srand(3);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * expf((float)(rand() % 7 - 3)); };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> ref(256), mag(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0, m = 0;
    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 16 + j]; m += fabs((double)A[i * K + k] * B[k * 16 + j]); }
    ref[i * 16 + j] = s; mag[i * 16 + j] = m;
  }
  for (int mode : {0, 3, 6}) {
    hipLaunchKernelGGL(gemm_tile, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
    hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    double e = 0, e2 = 0;
    for (int i = 0; i < 256; ++i) { double d = fabs(C[i] - ref[i]) / mag[i]; e = fmax(e, d); e2 += d * d; }
    printf("K=%d %-28s max |err| / sum|a*b| = %.3e   rms %.3e\n", K, mode == 0 ? "fp32 mfma 16x16x4" : mode == 3 ? "bf16 x3 (hh,hm,mh)" : "bf16 x6 (+mm,hl,lh)", e, sqrt(e2 / 256));
  }
  float* dO; hipMalloc(&dO, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  auto timeit = [&](auto kern, const char* name, double flop_per_iter) {
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dO, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dO, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave and iteration: 16 tiles x (16x16 outputs x K=32) x 2 flop of fp32-equivalent work
    double fl = 1024.0 * 4 * iters * flop_per_iter;
    printf("%-34s %.3f ms  -> %.1f TFLOP/s fp32-equivalent\n", name, ms, fl / ms / 1e9);
  };
  const double fpi = 16.0 * 16 * 16 * 32 * 2;
  timeit(rate<0>, "fp32 mfma 16x16x4 (8 per K=32)", fpi);
  timeit(rate<3>, "bf16 mfma 16x16x32 x3 terms", fpi);
  timeit(rate<6>, "bf16 mfma 16x16x32 x6 terms", fpi);
  return 0;
}
