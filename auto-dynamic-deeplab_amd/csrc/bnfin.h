// BatchNorm finalize by the producing kernel's LAST-ARRIVING workgroup.
//
// A conv kernel in training mode ends with every workgroup writing one row [C][2] of fp64 (sum, sum of squares) partials.
// Until round 3 a second launch (bn_finalize) reduced the rows and produced the lazy affine (a, b): a ~6 us latency chain
// plus a kernel boundary behind EVERY one of the 312 BatchNorms of a step, all on the critical path.  Here the workgroup
// whose ticket comes last does it in the same launch:
//   * every workgroup stores its row WRITE-THROUGH (8-byte agent-scope stores = `global_store ... sc1`), every storing wave
//     drains them (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane takes a ticket with a relaxed agent-scope
//     atomic add  (cdna_hip_programming.md Guideline 16, form R1: no release fence, the payload is already out of the L2);
//   * the workgroup whose ticket is nblocks-1 runs ONE agent-scope acquire (its CU's L1 may hold stale lines of the slab
//     from an earlier replay), reads all rows with agent-scope loads and reduces them in a FIXED order (bit-reproducible:
//     hipGraph replay == eager), writes (a, b, mean, invstd, running statistics) and puts the ticket counter back to 0, so
//     the launch can be replayed from a graph without a memset node.
// Nothing spins: no workgroup ever waits for another one.
#pragma once
#include "common.h"

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct BnFin {
  float* a; float* b; float* mean; float* invstd;          // a == nullptr: no fused finalize
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  double count; float momentum, eps;
  unsigned* counter;                                       // one zero-initialised word per BatchNorm call
};

__device__ __forceinline__ void slab_store_wt(double* p, double v) {
  __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double slab_load_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// To be called by ALL threads of the workgroup (256 of them) after the slab-row stores of this workgroup have been issued with
// slab_store_wt.  `sh` is LDS scratch of at least 256 * 2 doubles + 4 bytes, free at this point.  slab is [nblocks][ld][2].
__device__ __forceinline__ void bn_finalize_by_last_block(const BnFin& f, const double* slab, int ld, int c0, int C, unsigned nblocks, double* sh) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores have left
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(sh + 512);
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add((gu32*)f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = tk == nblocks - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  if (*flag == 0u) return;
  // ---- the last workgroup: rows in fixed order.  Thread (c, j): channel c, row lane j of NL; then the NL lanes in order ----
  const int t = threadIdx.x;
  for (int cb = 0; cb < C; cb += 256) {                     // C <= 256 on the path: one pass
    const int cn = (C - cb < 256) ? C - cb : 256;
    const int NL = 256 / cn;                                // row lanes per channel (>= 1)
    const int c = t % cn, j = t / cn;
    double s0 = 0.0, s1 = 0.0;
    if (j < NL) {
      const double* base = slab + ((long)(c0 + cb + c)) * 2;
      unsigned r = j;
      for (; r + 3u * NL < nblocks; r += 4u * NL) {         // four independent rows in flight
        const double* p0 = base + (long)r * ld * 2, *p1 = p0 + (long)NL * ld * 2, *p2 = p1 + (long)NL * ld * 2, *p3 = p2 + (long)NL * ld * 2;
        const double a0 = slab_load_agent(p0), b0 = slab_load_agent(p0 + 1), a1 = slab_load_agent(p1), b1 = slab_load_agent(p1 + 1);
        const double a2 = slab_load_agent(p2), b2 = slab_load_agent(p2 + 1), a3 = slab_load_agent(p3), b3 = slab_load_agent(p3 + 1);
        s0 += (a0 + a1) + (a2 + a3); s1 += (b0 + b1) + (b2 + b3);
      }
      for (; r < nblocks; r += NL) { const double* p0 = base + (long)r * ld * 2; s0 += slab_load_agent(p0); s1 += slab_load_agent(p0 + 1); }
    }
    __syncthreads();
    sh[2 * t] = s0; sh[2 * t + 1] = s1;
    __syncthreads();
    if (j == 0) {
      double m0 = 0.0, m1 = 0.0;
      for (int k = 0; k < NL; ++k) { m0 += sh[2 * (k * cn + c)]; m1 += sh[2 * (k * cn + c) + 1]; }
      const int ch = cb + c;                                // index into the BatchNorm's vectors (c0 = column offset inside the slab)
      const double mean = m0 / f.count;
      double var = m1 / f.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)f.eps);
      const float g = f.gamma ? ((const gfloat*)f.gamma)[ch] : 1.f, be = f.beta ? ((const gfloat*)f.beta)[ch] : 0.f;
      ((gfloat*)f.a)[ch] = (float)(g * invstd);
      ((gfloat*)f.b)[ch] = (float)(be - mean * (g * invstd));
      if (f.mean) ((gfloat*)f.mean)[ch] = (float)mean;
      if (f.invstd) ((gfloat*)f.invstd)[ch] = (float)invstd;
      if (f.running_mean) {
        const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
        gfloat* rm = (gfloat*)f.running_mean; gfloat* rv = (gfloat*)f.running_var;
        rm[ch] = (float)((1.0 - f.momentum) * rm[ch] + f.momentum * mean);
        rv[ch] = (float)((1.0 - f.momentum) * rv[ch] + f.momentum * unb);
      }
    }
  }
  if (t == 0) __hip_atomic_store((gu32*)f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
