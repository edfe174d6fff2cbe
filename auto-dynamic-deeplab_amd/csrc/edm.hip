// Earlier-Decision-Maker head in ONE launch (reference modeling/ADD.py:502-525: in-place ReLU -> conv 3x3 stride 2, 400 -> 128, no bias
// -> ReLU -> global average pool -> Linear 128-64 -> ReLU -> Linear 64-32 -> ReLU -> Linear 32-1).  The gate of dynamic_inference
// (ADD.py:420-423) waits for this scalar, so the head sits on the critical path of EVERY image: as five generic launches (strided conv,
// two pooling kernels, three 1x1 GEMMs of one pixel) plus a layout launch and a device-to-host copy it was seven dependent steps.
//
//   * the convolution is an implicit GEMM on the exact fp32 matrix cores (v_mfma_f32_16x16x4_f32): a wave owns 16 output pixels x 16
//     output channels and walks K = 9 taps x C input channels in 16-channel groups; both fragments come straight from global memory
//     (16 bytes per lane: lane (li, kq) holds k = 16g + 4kq + {0..3} of pixel / output channel li — the k order inside a group is a
//     permutation that weights and pixels share), the lazy BatchNorm / ReLU prologue applied in registers, zero padding by masking; no
//     LDS, no barrier in the loop, a five-group software pipeline (ten 16-byte loads per lane in flight).  A workgroup = 4 waves = 64
//     output channels of one 16-pixel tile (the four waves read the same pixels: L1 hits); grid = pixel tiles x 2 channel halves
//     (256 workgroups at 1 x 64 x 128 x 400: one per CU);
//   * ReLU, then the 16 pixel lanes are summed by a butterfly and every wave writes ONE partial row [16 channels] write-through;
//   * the workgroup that arrives LAST (ticket by a relaxed agent-scope atomic, csrc/bnfin.h) adds the partial rows in a FIXED order
//     (fp64; bit-reproducible whichever workgroup is last), divides by the pixel count and runs the three Linear layers out of LDS,
//     then writes the confidence to `out` and — when given — to a host-mapped word the gate reads, and puts the ticket back to zero
//     (graph-replayable).  Nothing spins.
#include <stdlib.h>
#include "common.h"
#include "bnfin.h"

namespace {

constexpr int EDM_CO = 128, EDM_H1 = 64, EDM_H2 = 32, EDM_PF = 5;      // PF: 16-channel groups per pipeline stage

struct EdmK {
  addk_src src; int N, H, W, OH, OW, C;
  const float* w; int ldw;                         // [128][9 * C] (OHWI)
  const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;
  float* partial;                                  // [N][tiles_img][128]
  unsigned* counter;
  float* out; int ldo; float* out_host;
  int tiles_img, ngroups;                          // 16-pixel tiles per image; 16-channel groups per tap
};

__device__ __forceinline__ void st_wt(float* p, float v) {
  __hip_atomic_store((gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __uint_as_float(__hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__global__ void __launch_bounds__(256) edm_head_kernel(const EdmK p) {
  __shared__ double sh[512 + 2];
  __shared__ float v0[EDM_CO], v1[EDM_H1], v2[EDM_H2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int tile = blockIdx.x, n = tile / p.tiles_img, ti = tile - n * p.tiles_img;
  const int co = (blockIdx.y * 4 + wave) * 16;                     // this wave's 16 output channels
  const int q = ti * 16 + li;                                     // output pixel of this lane inside image n
  const bool qok = q < p.OH * p.OW;
  const int oh = qok ? q / p.OW : 0, ow = qok ? q - (q / p.OW) * p.OW : 0;
  const float* xb = p.src.x + (long)n * p.H * p.W * p.src.ld;
  const float* wrow = p.w + (long)(co + li) * p.ldw;
  const bool relu = p.src.relu != 0;
  const int total = 9 * p.ngroups;                                // (tap, group) steps of the K walk

  struct Grp { float4 x, w; bool ok; int k; };
  auto load = [&](Grp& G, int s) {
    const int tap = s / p.ngroups, g = s - tap * p.ngroups;
    const int ih = 2 * oh - 1 + tap / 3, iw = 2 * ow - 1 + tap % 3;
    const int k = 16 * g + 4 * kq;
    const bool kok = s < total && k < p.C;
    G.ok = kok && qok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    G.x = ld4(xb + (G.ok ? ((long)ih * p.W + iw) * p.src.ld + k : 0));
    G.w = ld4(wrow + (kok ? (long)tap * p.C + k : 0));
    if (!kok) G.w = zero4();
    G.k = kok ? k : 0;
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const Grp& G) {
    float4 v = G.x;
    if (p.src.a) {             // (wave-uniform; the gated feature is a cell's concat: no pending BatchNorm in the shipped networks — loaded late, L1 hits)
      const float4 a = ld4(p.src.a + G.k), b = ld4(p.src.b + G.k);
      v.x = fmaf(a.x, v.x, b.x); v.y = fmaf(a.y, v.y, b.y); v.z = fmaf(a.z, v.z, b.z); v.w = fmaf(a.w, v.w, b.w);
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    v.x = G.ok ? v.x : 0.f; v.y = G.ok ? v.y : 0.f; v.z = G.ok ? v.z : 0.f; v.w = G.ok ? v.w : 0.f;      // zero padding AFTER the prologue, as the reference pads relu(x)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(get4(G.w, e), get4(v, e), acc, 0, 0, 0);
  };
  Grp A[EDM_PF], B[EDM_PF];
#pragma unroll
  for (int u = 0; u < EDM_PF; ++u) load(A[u], u);
  for (int s = 0; s < total; s += 2 * EDM_PF) {
#pragma unroll
    for (int u = 0; u < EDM_PF; ++u) load(B[u], s + EDM_PF + u);
#pragma unroll
    for (int u = 0; u < EDM_PF; ++u) compute(A[u]);
#pragma unroll
    for (int u = 0; u < EDM_PF; ++u) load(A[u], s + 2 * EDM_PF + u);
#pragma unroll
    for (int u = 0; u < EDM_PF; ++u) compute(B[u]);
  }
  // ---- ReLU, sum over the 16 pixel lanes: lane (li, kq) holds channels co + 4kq + {0..3} of pixel li ----
  float s4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float f = qok ? fmaxf(acc[e], 0.f) : 0.f;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) f += __shfl_xor(f, m);
    s4[e] = f;
  }
  if (li == 0) {
    float* o = p.partial + ((long)n * p.tiles_img + ti) * EDM_CO + co + 4 * kq;
#pragma unroll
    for (int e = 0; e < 4; ++e) st_wt(o + e, s4[e]);
  }
  unsigned* flag = reinterpret_cast<unsigned*>(sh + 512);
  if (!bnfin_arrive(p.counter, gridDim.x * gridDim.y, flag)) return;
  // ---- the last workgroup: pooled vector -> MLP -> confidence, image by image ----
  for (int img = 0; img < p.N; ++img) {
    {
      const int c = t & (EDM_CO - 1), j = t >> 7;                // 2 lanes per channel, rows j, j + 2, ... in a fixed order
      double s = 0.0;
      const float* base = p.partial + (long)img * p.tiles_img * EDM_CO + c;
      for (int r = j; r < p.tiles_img; r += 2) s += (double)ld_agent(base + (long)r * EDM_CO);
      sh[t] = s;
    }
    __syncthreads();
    if (t < EDM_CO) v0[t] = (float)((sh[t] + sh[t + EDM_CO]) / (double)(p.OH * p.OW));
    __syncthreads();
    if (t < EDM_H1) {
      float s = ((const gfloat*)p.b1)[t];
      for (int k = 0; k < EDM_CO; ++k) s = fmaf(((const gfloat*)p.w1)[t * EDM_CO + k], v0[k], s);
      v1[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t < EDM_H2) {
      float s = ((const gfloat*)p.b2)[t];
      for (int k = 0; k < EDM_H1; ++k) s = fmaf(((const gfloat*)p.w2)[t * EDM_H1 + k], v1[k], s);
      v2[t] = fmaxf(s, 0.f);
    }
    __syncthreads();
    if (t == 0) {
      float s = ((const gfloat*)p.b3)[0];
      for (int k = 0; k < EDM_H2; ++k) s = fmaf(((const gfloat*)p.w3)[k], v2[k], s);
      ((gfloat*)p.out)[(long)img * p.ldo] = s;
      if (p.out_host) { p.out_host[img] = s; }
    }
    __syncthreads();
  }
  if (t == 0) {
    if (p.out_host) __threadfence_system();
    __hip_atomic_store((gu32*)p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

bool edm_fill(const addk_edm_args* a, EdmK& k) {
  if (!a || a->N <= 0 || a->H <= 1 || a->W <= 1) return false;
  const addk_src& s = a->src;
  if (!s.x || s.C < 16 || s.rs_hw || !src_vec_ok(s) || (s.a == nullptr) != (s.b == nullptr)) return false;
  if (!a->conv_w || !aligned16(a->conv_w) || (9 * s.C) % 4 || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->w3 || !a->b3 || !a->out || a->ldo < 1 || !a->ws) return false;
  k = EdmK{};
  k.src = s; k.N = a->N; k.H = a->H; k.W = a->W; k.C = s.C;
  k.OH = (a->H + 2 - 3) / 2 + 1; k.OW = (a->W + 2 - 3) / 2 + 1;
  k.w = a->conv_w; k.ldw = 9 * s.C;
  k.w1 = a->w1; k.b1 = a->b1; k.w2 = a->w2; k.b2 = a->b2; k.w3 = a->w3; k.b3 = a->b3;
  k.tiles_img = cdiv((long)k.OH * k.OW, 16); k.ngroups = cdiv(s.C, 16);
  if ((long)a->N * k.tiles_img > 65535L * 16) return false;
  k.counter = (unsigned*)a->ws;
  k.partial = (float*)((char*)a->ws + 16);
  k.out = a->out; k.ldo = a->ldo; k.out_host = a->out_host;
  return true;
}

}  // namespace

extern "C" int64_t addk_edm_head_ws_bytes(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 1 || W <= 1) return 0;
  const long OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  return 16 + (int64_t)N * cdiv(OH * OW, 16) * EDM_CO * 4;
}
extern "C" int addk_edm_head_supported(const addk_edm_args* a) {
  EdmK k;
  return edm_fill(a, k) ? 1 : 0;
}
extern "C" int addk_edm_head(const addk_edm_args* a, void* stream) {
  EdmK k;
  ADDK_REQUIRE(edm_fill(a, k), "edm_head: bad arguments (NHWC source with C >= 16 and C %% 4 == 0, 16-byte aligned weights, ws of addk_edm_head_ws_bytes() zero-initialised bytes)");
  hipLaunchKernelGGL(edm_head_kernel, dim3(a->N * k.tiles_img, EDM_CO / 64), dim3(256), 0, (hipStream_t)stream, k);
  return addk_check_launch("edm_head");
}
