// Probe: what does the bf16 matrix pipe of THIS MI355X sustain, and at what shader clock?  ("measure the in-kernel
// clock with s_memtime / s_memrealtime instead of inferring 1.7 GHz".)
//
// Register-resident v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 chains (no memory traffic, 4 independent accumulators per wave,
// 1 / 2 waves per SIMD on every CU) run for ~1-5 ms.  Every wave reads the shader-clock counter (s_memtime) and the constant 100 MHz
// reference counter (s_memrealtime) before and after its loop: their ratio is the clock the CU actually ran at under this load,
// the MFMA count over the s_memtime delta is the issue rate in MFMAs per SIMD-cycle (1 / passes when the pipe is saturated).
//   hipcc -O3 -Wno-unused-value --offload-arch=gfx950 scripts/mfma_clock_probe.hip -o /tmp/mfma_clock_probe && /tmp/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Stamp { unsigned long long clk, ref; };

template <int SHAPE, int NACC, bool TOGGLE = false>      // SHAPE 0: 32x32x16 (32768 flop), 1: 16x16x32 (16384 flop); TOGGLE: operands change every iteration (switching power)
__global__ void __launch_bounds__(256) mfma_loop(float* out, Stamp* stamps, int iters, float idle_scale) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1e-3f * (threadIdx.x + j)); b[j] = (__bf16)(2e-3f * (threadIdx.x - j)); }
  f32x16 acc32[NACC]; f32x4 acc16[NACC];
  for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f; acc16[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 flipa = {0x3a5c3a5cu, 0x15a315a3u, 0x2c6a2c6au, 0x19951995u}, flipb = {0x1d3b1d3bu, 0x2e472e47u, 0x0b6d0b6du, 0x37193719u};   // mantissa / low exponent bits
  for (int it = 0; it < iters; ++it) {
    if (TOGGLE) { a = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a) ^ flipa); b = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, b) ^ flipb); }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (SHAPE == 0) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
      else acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) s += acc32[i][j]; s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3]; }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s * idle_scale;
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0};
}

template <int SHAPE, int NACC, bool TOGGLE = false>
static void run(const char* name, int blocks, int iters, float* out, Stamp* dst) {
  const double flop = SHAPE == 0 ? 32768.0 : 16384.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_loop<SHAPE, NACC, TOGGLE>), dim3(blocks), dim3(256), 0, 0, out, dst, iters, 0.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_loop<SHAPE, NACC, TOGGLE>), dim3(blocks), dim3(256), 0, 0, out, dst, iters, 0.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks * 4);
  hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> mhz, rate;
  for (auto& s : h) { if (s.ref == 0) continue; mhz.push_back(100.0 * (double)s.clk / (double)s.ref); rate.push_back((double)iters * NACC / (double)s.clk); }
  std::sort(mhz.begin(), mhz.end()); std::sort(rate.begin(), rate.end());
  const double tf = flop * NACC * (double)iters * 4.0 * blocks / ms * 1e-9;
  printf("%-22s blocks %5d (%.0f waves/SIMD)  %.3f ms  %7.1f TFLOP/s  | shader clock MHz min %.0f median %.0f max %.0f | MFMA per SIMD-cycle per wave median %.4f\n",
         name, blocks, blocks / 256.0, ms, tf, mhz.front(), mhz[mhz.size() / 2], mhz.back(), rate[rate.size() / 2]);
}

int main() {
  float* out; Stamp* st;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 4 * sizeof(Stamp));
  for (int rep = 0; rep < 2; ++rep) {
    for (int blocks : {256, 512}) {
      run<0, 4>("32x32x16 bf16, 4 acc", blocks, 40000, out, st);
      run<1, 4>("16x16x32 bf16, 4 acc", blocks, 80000, out, st);
    }
    run<0, 4, true>("32x32x16 toggling", 512, 40000, out, st);
    run<0, 4>("32x32x16 bf16, long", 512, 400000, out, st);        // ~30+ ms: the clock after the power manager has reacted
    run<0, 4, true>("32x32x16 toggling, long", 512, 400000, out, st);
  }
  return 0;
}
