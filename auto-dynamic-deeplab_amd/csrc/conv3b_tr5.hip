// conv3b_tr5.hip — instantiations of conv3b_kernel with TWO-ROW tiles for the 5x5 convolutions at dilation <= 2 (conv3b.h; launch logic: conv3.hip c3b_launch).
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

// (waves across channels, pixels per tile row): 64-pixel rows for the full-width launches, 32 for the half-width ones of 3-wave blocks
#define C3B_TR(W_, X_) if (wc == W_ && rpx == X_) { \
    constexpr int H_ = W_ == 2 ? 2 : 1; \
    if (mode == MODE_FWD) { if (np == 3) C3B_GO((conv3b_kernel<W_, 5, MODE_FWD, 3, false, H_, X_, 1, 2>), 64 * W_ * H_) else C3B_GO((conv3b_kernel<W_, 5, MODE_FWD, 2, false, H_, X_, 1, 2>), 64 * W_ * H_) } \
    else { if (np == 3) C3B_GO((conv3b_kernel<W_, 5, MODE_DGRAD, 3, false, H_, X_, 1, 2>), 64 * W_ * H_) else C3B_GO((conv3b_kernel<W_, 5, MODE_DGRAD, 2, false, H_, X_, 1, 2>), 64 * W_ * H_) } }
int c3b_run_tr5(const void* kp, int wc, int rpx, int mode, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
  C3B_TR(2, 64) C3B_TR(3, 64) C3B_TR(4, 64) C3B_TR(5, 64) C3B_TR(3, 32)
  return 0;
}
C3B_DIAG_READER(c3b_diag_tr5)
