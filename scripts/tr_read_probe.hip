// Probe: lane -> element map of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_*) on gfx950, and a 16x16x32 bf16 MFMA fed
// by transposed reads of [k][16] row-major LDS images for BOTH operands (K = rows of the images), checked against a host sum.
//   hipcc --offload-arch=gfx950 -O3 scripts/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ void map_kernel(unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[64 * 16];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (unsigned short)i;    // element (row, col) = row*16 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned short* a = &sm[(8 * g + q) * 16 + 4 * p];                   // lane 4q+p of a group: (row q, cols 4p..4p+3) of the block at row 8g
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  unsigned long long r = 0;
  for (int e = 0; e < 4; ++e) r |= (unsigned long long)(unsigned short)v[e] << (16 * e);
  out[lane] = r;
}

// D[16 co][16 ci] = sum_k A[k][co] * B[k][ci], K = 32, both images row-major [k][16] in LDS (bf16)
__global__ void mfma_kernel(const unsigned short* A, const unsigned short* B, float* D) {
  __shared__ __attribute__((aligned(16))) unsigned short sa[32 * 16], sb[32 * 16];
  for (int i = threadIdx.x; i < 512; i += 64) { sa[i] = A[i]; sb[i] = B[i]; }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  auto frag = [&](const unsigned short* img) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&img[(8 * g + q) * 16 + 4 * p]);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&img[(8 * g + 4 + q) * 16 + 4 * p]);
    // NB: assembling the fragment element by element (f[e] = bit_cast<bf16>(lo[e]) ...) is miscompiled by hipcc 7.2 — it keeps only
    // the low dword of each 64-bit result; reinterpret the register pair as a whole instead
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8, pr);
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(sa), frag(sb), acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) D[((lane >> 4) * 4 + e) * 16 + (lane & 15)] = acc[e];      // row = co, col = ci
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(map_kernel, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    for (int e = 0; e < 4; ++e) { const int got = (int)((h[l] >> (16 * e)) & 0xffff), want = (8 * g + e) * 16 + i; if (got != want) { if (bad < 8) printf("lane %d elem %d: got (row %d, col %d) want (row %d, col %d)\n", l, e, got / 16, got % 16, want / 16, want % 16); ++bad; } }
  }
  printf("tr16_b64 map: lane 16g+i receives column i of rows 8g..8g+3 in elements 0..3: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  std::vector<unsigned short> A(512), B(512); std::vector<float> Af(512), Bf(512);
  for (int i = 0; i < 512; ++i) { Af[i] = (float)((i * 7) % 13 - 6); Bf[i] = (float)((i * 5 + 3) % 11 - 5); A[i] = f2bf(Af[i]); B[i] = f2bf(Bf[i]); }
  unsigned short *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int co = 0; co < 16; ++co) for (int ci = 0; ci < 16; ++ci) {
    float s = 0; for (int k = 0; k < 32; ++k) s += Af[k * 16 + co] * Bf[k * 16 + ci];
    if (s != D[co * 16 + ci]) { if (bad2 < 5) printf("D[%d][%d] = %g want %g\n", co, ci, D[co * 16 + ci], s); ++bad2; }
  }
  printf("16x16x32 MFMA on transposed reads of [k][16] images: %s (%d mismatches)\n", bad2 ? "WRONG" : "exact", bad2);
  return 0;
}
