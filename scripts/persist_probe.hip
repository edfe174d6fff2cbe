// Persistent launch with grid barriers vs a chain of dependent launches, at the size of ONE operator of a level-3 cell.
//
// A level-3 cell of ADD (32x64 maps, 160 channels, bs = 2: 2.6 MB per tensor) is ~15 dependent operators of a few microseconds each;
// as launches each costs its in-kernel time PLUS what lies between two dependent launches.  The question: does ONE persistent launch
// whose phases are separated by a grid barrier beat the chain?  This probe runs L phases of a streaming pass  y[i] = fma(1.0001, x[i], 1)
// over `bytes` of fp32 (every workgroup reads what OTHER workgroups wrote in the previous phase: the block -> element map rotates per phase,
// as a real operator's halo / channel mixing does) three ways:
//   (a) L dependent launches on one stream, replayed from a hipGraph;
//   (b) one launch, L phases, counter barrier: every workgroup drains its stores, one lane releases (agent), adds to a monotonic counter,
//       polls it with relaxed sc1 loads, acquires (agent) — MI355X_MICROARCH.md 'barrier-counter';
//   (c) the same with the XCD-hierarchical form ('barrier-xcd'): per-XCC arrival counters, the XCD's last arriver releases and adds to the
//       top counter, polls it, then publishes a per-XCC generation word the others of its XCD poll.
// Every spin is BOUNDED (a stuck barrier sets an error flag and every workgroup leaves).  Results are checked against the host.
//   hipcc -O3 --offload-arch=gfx950 scripts/persist_probe.hip -o /tmp/persist_probe && /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned int gu32;
constexpr long SPIN_LIMIT = 2000000;       // polls; ~0.5 s: a healthy barrier needs a few hundred

__device__ __forceinline__ void pass(const float4* __restrict__ x, float4* __restrict__ y, long n4, int phase, int bid, int nblk) {
  // block b works on slice (b + 17 * phase) % nblk: the data it reads was written by another workgroup (another CU, usually another XCD)
  const int s = (bid + 17 * phase) % nblk;
  const long per = (n4 + nblk - 1) / nblk, lo = (long)s * per, hi = lo + per < n4 ? lo + per : n4;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float4 v = x[i];
    v.x = fmaf(1.0001f, v.x, 1.f); v.y = fmaf(1.0001f, v.y, 1.f); v.z = fmaf(1.0001f, v.z, 1.f); v.w = fmaf(1.0001f, v.w, 1.f);
    y[i] = v;
  }
}

__global__ void __launch_bounds__(256) pass_kernel(const float4* x, float4* y, long n4, int phase) { pass(x, y, n4, phase, blockIdx.x, gridDim.x); }

// (b) single monotonic counter
__device__ __forceinline__ bool barrier_counter(unsigned* ctr, unsigned target, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add((gu32*)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load((gu32*)ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT || __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __shared__ int okflag;
  if (threadIdx.x == 0) okflag = ok ? 1 : 0;
  __syncthreads();
  const bool r = okflag != 0;
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) persist_counter_kernel(float4* a, float4* b, long n4, int L, unsigned* ctr, unsigned* err) {
  for (int ph = 0; ph < L; ++ph) {
    pass((ph & 1) ? b : a, (ph & 1) ? a : b, n4, ph, blockIdx.x, gridDim.x);
    if (ph + 1 < L && !barrier_counter(ctr, (unsigned)(ph + 1) * gridDim.x, err)) return;
  }
}

// (c) XCD-hierarchical: xc[0..7] arrival counters, top counter, gen[0..7] generation words (each on its own 128-byte line)
__device__ __forceinline__ bool barrier_xcd(unsigned* ws, unsigned phase1, unsigned* nper, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 | (0 << 6) | (3 << 11))) & 7u;      // HW_REG_XCC_ID bits [3:0]
    unsigned* arrive = ws + 32 * xcc; unsigned* top = ws + 32 * 8; unsigned* gen = ws + 32 * (9 + xcc);
    const unsigned mine = nper[xcc];                      // workgroups of this launch on this XCD (census phase)
    const unsigned t = __hip_atomic_fetch_add((gu32*)arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    if (t + 1 == phase1 * mine) {                         // this XCD's last arriver: publish the XCD's stores, meet the other leaders
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add((gu32*)top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load((gu32*)top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1 * 8u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
      }
      __hip_atomic_store((gu32*)gen, phase1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load((gu32*)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
      }
    }
    if (!ok) __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __shared__ int okflag;
  if (threadIdx.x == 0) okflag = ok ? 1 : 0;
  __syncthreads();
  const bool r = okflag != 0;
  __syncthreads();
  return r;
}
// NOTE: plain (non write-through) stores + the XCD leader's release only cover the LEADER's XCD L2 — which is the point of the hierarchy:
// every workgroup of an XCD shares that L2, and `buffer_wbl2` writes the whole L2's dirty lines back.

__global__ void __launch_bounds__(256) census_kernel(unsigned* nper) {
  if (threadIdx.x == 0) atomicAdd(&nper[__builtin_amdgcn_s_getreg((20 | (0 << 6) | (3 << 11))) & 7u], 1u);
}

__global__ void __launch_bounds__(256) persist_xcd_kernel(float4* a, float4* b, long n4, int L, unsigned* ws, unsigned* nper, unsigned* err) {
  for (int ph = 0; ph < L; ++ph) {
    pass((ph & 1) ? b : a, (ph & 1) ? a : b, n4, ph, blockIdx.x, gridDim.x);
    if (ph + 1 < L && !barrier_xcd(ws, (unsigned)(ph + 1), nper, err)) return;
  }
}

static float host_expect(float v, int L) { for (int i = 0; i < L; ++i) v = std::fmaf(1.0001f, v, 1.f); return v; }

int main(int argc, char** argv) {
  const int L = 16, NBLK = 256, REP = 20;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long bytes : {2621440L, 5242880L, 20971520L}) {            // one level-3 tensor, two, and a level-2 tensor (63x127x80x2 ~ 5 MB .. 20 MB)
    const long n = bytes / 4, n4 = n / 4;
    float *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    std::vector<float> h(n); for (long i = 0; i < n; ++i) h[i] = (float)(i % 97) * 0.01f;
    unsigned *ws, *nper, *err; CK(hipMalloc(&ws, 32 * 17 * 4)); CK(hipMalloc(&nper, 8 * 4)); CK(hipMalloc(&err, 4));
    auto reset = [&] { CK(hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemset(ws, 0, 32 * 17 * 4)); CK(hipMemset(err, 0, 4)); };
    auto check = [&](const char* what) {
      std::vector<float> out(n); CK(hipMemcpy(out.data(), (L & 1) ? b : a, bytes, hipMemcpyDeviceToHost));
      unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      long bad = 0; for (long i = 0; i < n; i += 37) if (out[i] != host_expect(h[i], L)) ++bad;
      if (herr || bad) printf("  !! %s: barrier error flag %u, %ld wrong elements\n", what, herr, bad);
      return !(herr || bad);
    };
    // (a) chain of launches in a hipGraph
    reset();
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int ph = 0; ph < L; ++ph) hipLaunchKernelGGL(pass_kernel, dim3(NBLK), dim3(256), 0, st, (const float4*)((ph & 1) ? b : a), (float4*)((ph & 1) ? a : b), n4, ph);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    const bool ok_a = check("launch chain");
    CK(hipEventRecord(e0, st)); for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms_a; CK(hipEventElapsedTime(&ms_a, e0, e1));
    // (b) persistent, counter barrier
    reset();
    hipLaunchKernelGGL(persist_counter_kernel, dim3(NBLK), dim3(256), 0, st, (float4*)a, (float4*)b, n4, L, ws, err); CK(hipStreamSynchronize(st));
    const bool ok_b = check("persistent / counter barrier");
    float ms_b = 0;
    if (ok_b) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < REP; ++r) { CK(hipMemsetAsync(ws, 0, 4, st)); hipLaunchKernelGGL(persist_counter_kernel, dim3(NBLK), dim3(256), 0, st, (float4*)a, (float4*)b, n4, L, ws, err); }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    // (c) persistent, XCD-hierarchical barrier (census of the placement first: blocks per XCD of a 256-block launch)
    reset(); CK(hipMemset(nper, 0, 32));
    hipLaunchKernelGGL(census_kernel, dim3(NBLK), dim3(256), 0, st, nper); CK(hipStreamSynchronize(st));
    unsigned hn[8]; CK(hipMemcpy(hn, nper, 32, hipMemcpyDeviceToHost));
    bool even = true; for (int i = 0; i < 8; ++i) even = even && hn[i] == NBLK / 8;
    float ms_c = 0; bool ok_c = false;
    if (even) {                                             // the barrier assumes the same placement for the next launch of the same shape; only run it then
      hipLaunchKernelGGL(persist_xcd_kernel, dim3(NBLK), dim3(256), 0, st, (float4*)a, (float4*)b, n4, L, ws, nper, err); CK(hipStreamSynchronize(st));
      ok_c = check("persistent / XCD barrier");
      if (ok_c) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) { CK(hipMemsetAsync(ws, 0, 32 * 17 * 4, st)); hipLaunchKernelGGL(persist_xcd_kernel, dim3(NBLK), dim3(256), 0, st, (float4*)a, (float4*)b, n4, L, ws, nper, err); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
      }
    }
    printf("%5.1f MB per phase, %d phases, %d workgroups: launch chain %6.2f us/phase%s | persistent + counter barrier %6.2f us/phase%s | persistent + XCD barrier %s\n",
           bytes / 1048576.0, L, NBLK, ms_a * 1e3 / (REP * L), ok_a ? "" : " (WRONG)", ms_b * 1e3 / (REP * L), ok_b ? "" : " (failed)",
           even ? (ok_c ? (std::to_string(ms_c * 1e3 / (REP * L)).substr(0, 6) + " us/phase").c_str() : "failed") : "skipped (uneven XCD placement)");
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ws)); CK(hipFree(nper)); CK(hipFree(err));
  }
  return 0;
}
