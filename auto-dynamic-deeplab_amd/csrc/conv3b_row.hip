// conv3b_row.hip — instantiations of conv3b_kernel with ONE-ROW tiles (conv3b.h; launch logic: conv3.hip c3b_launch): the wide dilations of ASPP
// (3x3, d = 6 / 12 / 18: the rows of a tile pair share nothing), half- and quarter-width tiles of small maps, and the pointwise GEMM form (KS = 1).
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

#define C3B_ROW(W_, K_, D_, X_) if (wc == W_ && ks == K_ && bigd == D_ && bpx == X_) { \
    constexpr int H_ = W_ == 2 ? 2 : 1; \
    if (mode == MODE_FWD) { if (np == 3) C3B_GO((conv3b_kernel<W_, K_, MODE_FWD, 3, D_, H_, X_>), 64 * W_ * H_) else C3B_GO((conv3b_kernel<W_, K_, MODE_FWD, 2, D_, H_, X_>), 64 * W_ * H_) } \
    else { if (np == 3) C3B_GO((conv3b_kernel<W_, K_, MODE_DGRAD, 3, D_, H_, X_>), 64 * W_ * H_) else C3B_GO((conv3b_kernel<W_, K_, MODE_DGRAD, 2, D_, H_, X_>), 64 * W_ * H_) } }
int c3b_run_row(const void* kp, int wc, int ks, bool bigd, int bpx, int mode, int np, dim3 grid, size_t lds, hipStream_t st) {
  const C3K& k = *reinterpret_cast<const C3K*>(kp);
  C3B_ROW(2, 3, true, C3_BP) C3B_ROW(3, 3, true, C3_BP) C3B_ROW(4, 3, true, C3_BP) C3B_ROW(5, 3, true, C3_BP) C3B_ROW(4, 3, true, 64)
  C3B_ROW(4, 3, false, 64) C3B_ROW(3, 3, false, 32) C3B_ROW(3, 5, false, 32)
  C3B_ROW(2, 1, false, C3_BP) C3B_ROW(3, 1, false, C3_BP) C3B_ROW(4, 1, false, C3_BP) C3B_ROW(5, 1, false, C3_BP) C3B_ROW(4, 1, false, 64)
  return 0;
}
C3B_DIAG_READER(c3b_diag_row)
