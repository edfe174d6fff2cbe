// conv3b_s2.hip — the stride-2 instantiations of conv3b_kernel (conv3b.h; launch logic: conv3.hip): stem2's 3x3 stride-2 forward on de-interleaved patch rows and
// its data gradient as four stride-1 parity classes in one launch (conv3b_s2d_kernel).
#include "conv3b.h"

#define C3B_GO(KERNEL, THREADS) { \
    static bool attr = false; \
    auto fn = &KERNEL; \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL(fn, grid, dim3(THREADS), lds, st, k); return 1; }
#ifdef ADDK_C3B_DIAG
#define C3B_DIAG_READER(NAME) void NAME(unsigned long long* acc12) { \
    unsigned long long h[64][12]; \
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3b_diag), sizeof h) != hipSuccess) { (void)hipGetLastError(); return; } \
    for (int q = 0; q < 12; ++q) for (int i = 0; i < 64; ++i) acc12[q] += h[i][q]; \
    memset(h, 0, sizeof h); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c3b_diag), h, sizeof h); }
#else
#define C3B_DIAG_READER(NAME)
#endif

int c3b_run_s2f(const void* kp, int wc, int bpx, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
  if (wc != 4) return 0;
  if (bpx == 64) { if (np == 3) C3B_GO((conv3b_kernel<4, 3, MODE_FWD, 3, false, 1, 64, 2>), 256) else C3B_GO((conv3b_kernel<4, 3, MODE_FWD, 2, false, 1, 64, 2>), 256) }
  return 0;
}
int c3b_run_s2d(const void* kp, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K4& k = *reinterpret_cast<const C3K4*>(kp);
  if (np == 3) C3B_GO((conv3b_s2d_kernel<3>), 256) else C3B_GO((conv3b_s2d_kernel<2>), 256)
  return 0;
}
C3B_DIAG_READER(c3b_diag_s2)
