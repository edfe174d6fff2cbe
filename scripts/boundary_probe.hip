// What does a kernel boundary between DEPENDENT launches cost as a function of the launch's shape (the fused SepConv half at 40 channels
// shows a 23.5 us period for an 11.3 us in-kernel span; the gap tracks the workgroup count: 128 / 256 / 512 workgroups -> 2.6 / 5.6 / 12.2 us)?
// A chain of 40 trivial dependent launches (each workgroup reads 16 bytes per thread of the previous launch's output, adds, stores) captured in a hipGraph;
// swept over workgroups x dynamic LDS per workgroup x bytes moved per thread x store flavour (plain / write-through sc1).
//   hipcc -O2 --offload-arch=gfx950 scripts/boundary_probe.hip -o /tmp/boundary_probe && /tmp/boundary_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool WT>
__global__ void __launch_bounds__(256) step_kernel(const float* __restrict__ in, float* __restrict__ out, int per_thread, int spin) {
  extern __shared__ float sm[];
  const long base = ((long)blockIdx.x * 256 + threadIdx.x) * 4 * per_thread;
  float keep = 0.f;
  if (spin) { sm[threadIdx.x] = (float)threadIdx.x; __syncthreads(); for (int i = 0; i < spin; ++i) keep += sm[(threadIdx.x + i) & 255]; }      // ~in-kernel work, LDS touched
  for (int i = 0; i < per_thread; ++i) {
    f4 v = *(const f4*)(in + base + 4 * i);
    v += 1.0f + keep * 0.f;
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"((__attribute__((address_space(1))) f4*)(out + base + 4 * i)), "v"(v) : "memory");
    else *(f4*)(out + base + 4 * i) = v;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int WGS[] = {128, 256, 512, 1024, 2048};
  const int LDS[] = {0, 32 * 1024, 65 * 1024};
  const int PER[] = {1, 5};          // x 16 bytes per thread: 512 workgroups x 256 threads x 80 B = 10 MB, the 40-channel tensor of the launch in question
  float *a, *b; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20)); CK(hipMemset(a, 0, 64 << 20)); CK(hipMemset(b, 0, 64 << 20));
  CK(hipFuncSetAttribute((const void*)step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  CK(hipFuncSetAttribute((const void*)step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  printf("%-6s %-8s %-10s %-6s %-6s  us per dependent launch\n", "wgs", "lds", "B/thread", "spin", "store");
  for (int spin : {0, 600})
  for (int wt = 0; wt < 2; ++wt)
  for (int per : PER) for (int lds : LDS) for (int wg : WGS) {
    if ((long)wg * 256 * 16 * per > (64 << 20)) continue;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 40; ++i) {
      float* in = i & 1 ? b : a; float* out = i & 1 ? a : b;
      if (wt) hipLaunchKernelGGL(step_kernel<true>, dim3(wg), dim3(256), lds, st, in, out, per, spin);
      else hipLaunchKernelGGL(step_kernel<false>, dim3(wg), dim3(256), lds, st, in, out, per, spin);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-6d %-8d %-10d %-6d %-6s  %6.2f\n", wg, lds, 16 * per, spin, wt ? "sc1" : "plain", ms * 1e3 / 400);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
