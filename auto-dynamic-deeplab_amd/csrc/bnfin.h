// BatchNorm finalize by the producing kernel's LAST-ARRIVING workgroups (two levels).
//
// A conv kernel in training mode ends with every workgroup writing one row [C][2] of fp64 (sum, sum of squares) partials.
// Until round 3 a second launch (bn_finalize) reduced the rows and produced the lazy affine (a, b): a ~6 us latency chain
// plus a kernel boundary behind EVERY one of the 312 BatchNorms of a step, all on the critical path.  Here the producing
// launch does it itself:
//   * every workgroup stores its row WRITE-THROUGH (8-byte agent-scope stores = `global_store ... sc1`), every storing wave
//     drains them (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane takes a ticket on the counter of its
//     GROUP of BNF_G consecutive workgroups with a relaxed agent-scope atomic add (cdna_hip_programming.md Guideline 16,
//     form R1: no release fence, the payload has already left the L2);
//   * the workgroup whose ticket completes a group runs ONE agent-scope acquire, adds the group's rows in a FIXED order
//     (all loads in flight at once: one memory round trip) into a group row, publishes it the same way and takes a ticket
//     on the top counter; the workgroup that completes the LAST group adds the group rows in fixed order, writes (a, b,
//     mean, invstd, running statistics) and puts every counter back to 0, so the launch can be replayed from a graph
//     without a memset node.  A single level (one workgroup reading 512 rows = 330 KB through one CU) measured +18 us on a
//     13 us kernel: the chain of dependent round trips of ONE workgroup is the price, so the first level spreads it over
//     the groups' last arrivers, most of which finish while other workgroups still compute.
// The sums are bit-reproducible (fixed order at both levels): hipGraph replay == eager.  Nothing spins: no workgroup ever
// waits for another one.
#pragma once
#include "common.h"

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int BNF_G = 16;                                   // workgroups per first-level group

struct BnFin {
  float* a; float* b; float* mean; float* invstd;          // a == nullptr: no fused finalize
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  double count; float momentum, eps;
  unsigned* counter;                                       // zero-initialised words: [0] top, [1 + g] group g
  double* gslab;                                           // [ngroups][ld][2] group rows
};

// words / doubles the caller provides for a launch of `nblocks` workgroups writing rows of `ld` channels
static inline int bnfin_groups(int nblocks) { return (nblocks + BNF_G - 1) / BNF_G; }
static inline int bnfin_counter_words(int nblocks) { return 1 + bnfin_groups(nblocks); }
static inline long bnfin_gslab_doubles(int nblocks, int ld) { return (long)bnfin_groups(nblocks) * ld * 2; }
// ONE zero-initialised workspace per BatchNorm call: the counter words (padded to 16 bytes), then the group rows
static inline long bnfin_ws_bytes(int nblocks, int ld) { return ((long)bnfin_counter_words(nblocks) * 4 + 15) / 16 * 16 + bnfin_gslab_doubles(nblocks, ld) * 8; }
static inline void bnfin_bind_ws(BnFin& f, void* ws, int nblocks) {
  f.counter = (unsigned*)ws;
  f.gslab = (double*)((char*)ws + ((long)bnfin_counter_words(nblocks) * 4 + 15) / 16 * 16);
}

__device__ __forceinline__ void slab_store_wt(double* p, double v) {
  __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double slab_load_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// one lane takes a ticket after the whole workgroup's write-through stores have drained; returns (to every thread) whether
// this workgroup completed the count.  `flag` is one LDS word.
__device__ __forceinline__ bool bnfin_arrive(unsigned* counter, unsigned expect, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = tk == expect - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  const bool r = *flag != 0u;
  __syncthreads();                                          // the flag word is reused by the next level
  return r;
}

// fixed-order sum of `nrows` rows [ld][2] (row stride ld*2 doubles) for channels [0, C): thread (c, j) adds rows j, j+NL, ..
// (all loads issued before the first add), then the NL lanes of a channel in order.  Result valid for threads with j == 0.
template <int MAXROWS_PER_LANE>
__device__ __forceinline__ void bnfin_sum_rows(const double* rows, unsigned nrows, int ld, int C, double* sh, double& m0, double& m1, int& c_out, bool& lead) {
  const int t = threadIdx.x;
  const int cn = C < 256 ? C : 256;
  const int NL = 256 / cn;
  const int c = t % cn, j = t / cn;
  double s0 = 0.0, s1 = 0.0;
  if (j < NL) {
    const double* base = rows + (long)c * 2;
    double v0[MAXROWS_PER_LANE], v1[MAXROWS_PER_LANE];
    for (unsigned r0 = j; r0 < nrows; r0 += (unsigned)(MAXROWS_PER_LANE * NL)) {
#pragma unroll
      for (int u = 0; u < MAXROWS_PER_LANE; ++u) {
        const unsigned r = r0 + (unsigned)(u * NL);
        const bool ok = r < nrows;
        const double* p = base + (long)(ok ? r : 0) * ld * 2;
        v0[u] = slab_load_agent(p); v1[u] = slab_load_agent(p + 1);
        if (!ok) { v0[u] = 0.0; v1[u] = 0.0; }
      }
#pragma unroll
      for (int u = 0; u < MAXROWS_PER_LANE; ++u) { s0 += v0[u]; s1 += v1[u]; }
    }
  }
  sh[2 * t] = s0; sh[2 * t + 1] = s1;
  __syncthreads();
  m0 = 0.0; m1 = 0.0;
  lead = j == 0;
  c_out = c;
  if (lead) for (int k = 0; k < NL; ++k) { m0 += sh[2 * (k * cn + c)]; m1 += sh[2 * (k * cn + c) + 1]; }
  __syncthreads();
}

// To be called by ALL 256 threads of the workgroup after this workgroup's slab row (row index `bid` of `nblocks`, C <= 256
// channels at column offset 0 of rows of `ld` channels) has been stored with slab_store_wt.  `sh`: LDS scratch of at least
// 512 doubles + 1 word, free at this point.
__device__ __forceinline__ void bn_finalize_by_last_block(const BnFin& f, const double* slab, int ld, int C, unsigned bid, unsigned nblocks, double* sh) {
  unsigned* flag = reinterpret_cast<unsigned*>(sh + 512);
  const unsigned ngroups = (nblocks + BNF_G - 1) / BNF_G;
  const unsigned grp = bid / BNF_G;
  const unsigned gsize = (grp + 1) * BNF_G <= nblocks ? BNF_G : nblocks - grp * BNF_G;
  if (!bnfin_arrive(f.counter + 1 + grp, gsize, flag)) return;
  double m0, m1; int c; bool lead;
  bnfin_sum_rows<BNF_G>(slab + (long)grp * BNF_G * ld * 2, gsize, ld, C, sh, m0, m1, c, lead);
  if (lead) { double* o = f.gslab + ((long)grp * ld + c) * 2; slab_store_wt(o, m0); slab_store_wt(o + 1, m1); }
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)(f.counter + 1 + grp), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!bnfin_arrive(f.counter, ngroups, flag)) return;
  bnfin_sum_rows<8>(f.gslab, ngroups, ld, C, sh, m0, m1, c, lead);
  if (lead) {
    const double mean = m0 / f.count;
    double var = m1 / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)f.eps);
    const float g = f.gamma ? ((const gfloat*)f.gamma)[c] : 1.f, be = f.beta ? ((const gfloat*)f.beta)[c] : 0.f;
    ((gfloat*)f.a)[c] = (float)(g * invstd);
    ((gfloat*)f.b)[c] = (float)(be - mean * (g * invstd));
    if (f.mean) ((gfloat*)f.mean)[c] = (float)mean;
    if (f.invstd) ((gfloat*)f.invstd)[c] = (float)invstd;
    if (f.running_mean) {
      const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
      gfloat* rm = (gfloat*)f.running_mean; gfloat* rv = (gfloat*)f.running_var;
      rm[c] = (float)((1.0 - f.momentum) * rm[c] + f.momentum * mean);
      rv[c] = (float)((1.0 - f.momentum) * rv[c] + f.momentum * unb);
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
