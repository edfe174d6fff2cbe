// Microbenchmark: sustained fp32 MFMA rate on gfx950 with operands held in registers (the ceiling conv.hip can reach).
// hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> static void run(const char* name, F launch, double flop_per_wave_iter, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = flop_per_wave_iter * iters * 4.0 * blocks;
    printf("%-28s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms * 1e-9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    int iters = 20000;
    for (int blocks : {256, 512, 1024, 2048}) {
        run("16x16x4 f32, 16 acc", [&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 16 * 2048.0, blocks, iters);
        run("16x16x4 f32, 4 acc", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 4 * 2048.0, blocks, iters);
        run("32x32x2 f32, 4 acc", [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 4 * 4096.0, blocks, iters);
    }
    return 0;
}
