// Do a matrix-phase wave and a VALU-phase wave of ONE SIMD overlap?  (wgrad_h3b's staging is not hidden under the other
// workgroup's MFMA phase: 762 us of MFMA + 517 us of staging = 1279 us, scripts/wgrad_ablate.sh.)
//
// One workgroup of 512 threads per CU: waves 0-3 are role A, waves 4-7 role B (one of each per SIMD).  Per mode, the time of
//   0  A: chain of v_mfma_f32_16x16x32_bf16 (4 independent accumulators), B: nothing
//   1  A: nothing, B: chain of v_fma_f32 (8 independent registers), 3 per MFMA of mode 0
//   2  A: the MFMA chain, B: the VALU chain                                   -> max(0, 1) if the phases overlap across waves, the sum if not
//   3  A and B: half of the MFMA chain each
//   4  A and B: each wave runs 1 MFMA + 3 VALU interleaved in ONE instruction stream, half of the iterations each (same total work as mode 2)
//   5  A: MFMA chain, B: chain of ds_write_b64 / global loads are not covered (the VALU port is the question)
//   6  like 2 with v_cvt_pk_bf16_f32 as the VALU instruction, 7 like 1 with it (is the conversion full rate?)
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c))
#define VCVT(r) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(m))

template <int VOP>
__device__ __forceinline__ void valu12(float (&r)[8], float m, float c) {
  if (VOP == 0) { VFMA(r[0]); VFMA(r[1]); VFMA(r[2]); VFMA(r[3]); VFMA(r[4]); VFMA(r[5]); VFMA(r[6]); VFMA(r[7]); VFMA(r[0]); VFMA(r[1]); VFMA(r[2]); VFMA(r[3]); }
  else { VCVT(r[0]); VCVT(r[1]); VCVT(r[2]); VCVT(r[3]); VCVT(r[4]); VCVT(r[5]); VCVT(r[6]); VCVT(r[7]); VCVT(r[0]); VCVT(r[1]); VCVT(r[2]); VCVT(r[3]); }
}

__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)(0.002f * (threadIdx.x & 3)); }
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 0.5f + 0.01f * i;
  const float m = 0.999f, c = 0.001f;
  // one iteration = 4 MFMA (role A work) and 12 VALU (role B work)
  if (mode == 0 || mode == 2 || mode == 6) { if (roleA) for (int it = 0; it < iters; ++it) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); } }
  if (mode == 1 || mode == 2) { if (!roleA) for (int it = 0; it < iters; ++it) valu12<0>(r, m, c); }
  if (mode == 7 || mode == 6) { if (!roleA) for (int it = 0; it < iters; ++it) valu12<1>(r, m, c); }
  if (mode == 3) for (int it = 0; it < iters / 2; ++it) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
  if (mode == 4) for (int it = 0; it < iters / 2; ++it) {
    MFMA(acc0); VFMA(r[0]); VFMA(r[1]); VFMA(r[2]);
    MFMA(acc1); VFMA(r[3]); VFMA(r[4]); VFMA(r[5]);
    MFMA(acc2); VFMA(r[6]); VFMA(r[7]); VFMA(r[0]);
    MFMA(acc3); VFMA(r[1]); VFMA(r[2]); VFMA(r[3]);
  }
  if (mode == 8) for (int it = 0; it < iters / 2; ++it) {      // 4 MFMA back to back, then their 12 VALU, in one stream
    MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); valu12<0>(r, m, c);
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i];
  for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  if (s == 12345.678f) out[threadIdx.x] = s;      // never true: keeps the chains alive
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount, iters = 200000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# %d CUs, one 512-thread workgroup per CU, %d iterations of (4 MFMA 16x16x32 bf16 | 12 VALU) per role; clock-independent ratios are what matters\n", cus, iters);
  const char* name[] = {"0 A: MFMA chain            B: -", "1 A: -                     B: v_fma chain", "2 A: MFMA chain            B: v_fma chain", "3 A, B: half the MFMA chain each",
                        "4 A, B: 1 MFMA + 3 v_fma interleaved in one stream, half each", "", "6 A: MFMA chain            B: v_cvt_pk_bf16_f32 chain", "7 A: -                     B: v_cvt_pk_bf16_f32 chain",
                        "8 A, B: 4 MFMA then 12 v_fma in one stream, half each"};
  for (int mode : {0, 1, 2, 3, 4, 8, 7, 6}) {
    hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, mode, 1000, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, mode, iters, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double mf = 4.0 * iters * 4 * cus * 16384.0;      // FLOP of the MFMA work (modes with it)
    printf("mode %-70s %8.3f ms   (MFMA work at %.0f TFLOP/s if present)\n", name[mode], best, mf / best * 1e-9);
  }
  return 0;
}
