// Shared device/host helpers for the addk gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "addk.h"

void addk_set_error(const char* fmt, ...);
int addk_check_launch(const char* what);
int addk_env(const char* name, int dflt);      // util.cpp: the library's only getenv (integer switches, read once under a lock)
int addk_env_math(int dflt);

#define ADDK_REQUIRE(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      addk_set_error(__VA_ARGS__);                 \
      return ADDK_ERR_INVALID;                     \
    }                                              \
  } while (0)

// arithmetic of the halo-patch convolutions when neither ADDK_MATH nor addk_set_conv_precision says otherwise (conv.hip)
#ifndef ADDK_DEFAULT_PRECISION
#define ADDK_DEFAULT_PRECISION 1
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// A source can use 16-byte accesses when its base, pixel stride and channel count allow it.
static inline bool src_vec_ok(const addk_src& s) {
  return aligned16(s.x) && (s.ld % 4 == 0) && (s.C % 4 == 0) && (!s.a || aligned16(s.a)) && (!s.b || aligned16(s.b));
}

// Pointers that reach a kernel through a descriptor TABLE in device memory (the batched launches) carry no address-space
// information: the compiler emits flat_load / flat_store for them, which count on lgkmcnt together with the LDS traffic
// (every wait for a fragment read then also waits for the global loads in flight) and probe the apertures.  gptr() states
// that a pointer is global memory; ld4 / st4 (and everything built on them) are GLOBAL accesses — LDS staging uses
// lds_ld4 / lds_st4.
// gfloat* / gdouble*: explicitly-global views for the scalar accesses.
template <class T> __device__ __forceinline__ T* gptr(T* p) { return (T*)(__attribute__((address_space(1))) T*)p; }
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) double gdouble;
typedef float addk_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
  const addk_f32x4 v = *(const __attribute__((address_space(1))) addk_f32x4*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
// 16-byte load at a 32-bit BYTE offset from a wave-uniform base: the `global_load_dwordx4 v, v_off, s[base]` form (one address register, no 64-bit
// vector arithmetic) — for loops whose base moves in scalar registers while every thread's offset stays put
__device__ __forceinline__ float4 ld4so(const float* sbase, unsigned byte_off) {
  const addk_f32x4 v = *(const __attribute__((address_space(1))) addk_f32x4*)((const __attribute__((address_space(1))) char*)sbase + byte_off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  const addk_f32x4 t = {v.x, v.y, v.z, v.w};
  *(__attribute__((address_space(1))) addk_f32x4*)p = t;
}
// Write-through 16-byte store (`sc1`: the bytes leave the XCD's L2 as they are written instead of waiting, dirty, for the end-of-kernel
// write-back — MI355X_MICROARCH.md, 'stores of each flavour' / 'publish-large').  For launches whose output is consumed by the NEXT kernel
// (another XCD's L2 never sees it anyway).  Measured on the fused SepConv half: 0.5-0.9 us per launch (profiles/r04_sepf_phases.txt); as a
// blanket replacement of st4 it was slower (36.0 vs 35.6 ms per step: profiles/r04_ab_store_wt_global_rejected.txt) — only sepf.hip uses it.
// The "memory" clobber is what keeps the compiler from moving dependent memory operations across a store it cannot see.
__device__ __forceinline__ void st4_wt(float* p, float4 v) {
  const addk_f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"((__attribute__((address_space(1))) addk_f32x4*)p), "v"(t) : "memory");
}
__device__ __forceinline__ float4 lds_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void lds_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// guarded 4-element load: elements with index >= n read as 0
__device__ __forceinline__ float4 ld4g(const float* p, int n, bool vec) {
  if (vec && n >= 4) return ld4(p);
  float4 v = zero4();
  const gfloat* g = (const gfloat*)p;
  if (n > 0) v.x = g[0];
  if (n > 1) v.y = g[1];
  if (n > 2) v.z = g[2];
  if (n > 3) v.w = g[3];
  return v;
}
__device__ __forceinline__ void st4g(float* p, float4 v, int n, bool vec) {
  if (vec && n >= 4) { st4(p, v); return; }
  gfloat* g = (gfloat*)p;
  if (n > 0) g[0] = v.x;
  if (n > 1) g[1] = v.y;
  if (n > 2) g[2] = v.z;
  if (n > 3) g[3] = v.w;
}

// lazy prologue z = relu?(a*x+b) on 4 channels; lanes >= n stay 0
__device__ __forceinline__ float4 prologue4(float4 x, const float* a, const float* b, int c, int n, bool relu, bool vec) {
  if (a) {
    float4 av = ld4g(a + c, n, vec), bv = ld4g(b + c, n, vec);
    x.x = fmaf(av.x, x.x, bv.x); x.y = fmaf(av.y, x.y, bv.y);
    x.z = fmaf(av.z, x.z, bv.z); x.w = fmaf(av.w, x.w, bv.w);
  }
  if (relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
  if (n < 4) { if (n < 1) x.x = 0.f; if (n < 2) x.y = 0.f; if (n < 3) x.z = 0.f; x.w = 0.f; }
  return x;
}

// Bilinear source index of F.interpolate(mode='bilinear', align_corners=False) in fp32, as ATen's area_pixel_compute_source_index:
//     scale = in/out;  src = max(0, scale*(dst+0.5)-0.5);  i0 = floor(src); i1 = min(i0+1, in-1); l1 = src-i0.
// One definition for the stand-alone resize kernels (resize.hip) and for the 1x1 convolutions that sample their input on the fly
// (pw.hip, addk_src.rs_hw): the two forms are bit-identical by construction.
__device__ __forceinline__ void src_index(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}
__device__ __forceinline__ float4 lerp4(float4 v00, float4 v01, float4 v10, float4 v11, float lh0, float lh1, float lw0, float lw1) {
  float4 o;
  o.x = lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
  o.y = lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
  o.z = lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
  o.w = lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
  return o;
}

__device__ __forceinline__ float get4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ void set4(float4& v, int e, float f) { if (e == 0) v.x = f; else if (e == 1) v.y = f; else if (e == 2) v.z = f; else v.w = f; }

// ---- split-fp16 arithmetic ("f16x3": conv3b.h, wgrad.hip) -------------------------------------------------------------------------------------
// x = h + l with h = fp16(x), l = fp16(x - h); products as l*wh + h*wl + h*wh on the fp16 matrix pipe; operands scaled by exact powers of two into fp16's range
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4h(const float4 v, uint2 (&pl)[2]) {
  const f16x2 h01 = {(_Float16)v.x, (_Float16)v.y}, h23 = {(_Float16)v.z, (_Float16)v.w};
  const f16x2 l01 = {(_Float16)(v.x - (float)h01[0]), (_Float16)(v.y - (float)h01[1])}, l23 = {(_Float16)(v.z - (float)h23[0]), (_Float16)(v.w - (float)h23[1])};
  pl[0] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  pl[1] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
// 8 scaled floats -> the two fp16 planes of a packed weight fragment (16 bytes each)
__device__ __forceinline__ void split8h(const float (&v)[8], uint4& ph, uint4& pl) {
  uint2 a[2], b[2];
  split4h(make_float4(v[0], v[1], v[2], v[3]), a); split4h(make_float4(v[4], v[5], v[6], v[7]), b);
  ph = make_uint4(a[0].x, a[0].y, b[0].x, b[0].y); pl = make_uint4(a[1].x, a[1].y, b[1].x, b[1].y);
}
// largest of a non-negative bit pattern (|x| as unsigned: order-preserving) over the 64 lanes, wave-uniform: two quad steps and two row rotations on the DPP
// path, then the four rows by readlane
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  auto mx = [](unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; };
  v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));       // quad_perm [1, 0, 3, 2]
  v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));       // quad_perm [2, 3, 0, 1]
  v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true));      // row_ror:4
  v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));      // row_ror:8
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned absbits4(const float4 v) {
  return __float_as_uint(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
// exponent field of the power of two that takes a tensor / chunk of largest magnitude `amax_bits` into [2^14, 2^15): 268 - exponent field, kept in [13, 253]
// (zero / denormal maxima: 2^126; Inf / NaN: 2^-114) so that the scale and its inverse (field 254 - k) are normal numbers
__device__ __forceinline__ int f16_scale_field(unsigned amax_bits) {
  const int want = 268 - (int)(amax_bits >> 23);
  return want > 253 ? 253 : want;
}


// Thread mapping for channel-reducing elementwise kernels: a 256-thread block is viewed as
// npl "pixel lanes" x nq channel quads; a thread keeps its quad for the whole kernel so that
// per-channel sums stay in registers.
struct EwMap { int nq, npl; };
static inline EwMap ew_map(int C) {
  EwMap m; m.nq = (C + 3) / 4; if (m.nq > 256) m.nq = 256; m.npl = 256 / m.nq; if (m.npl < 1) m.npl = 1; return m;
}

// pw.hip: register-stationary 1x1 convolution; 0 = launched, 1 = shape not covered (fall back), <0 = error
int addk_pw_try_fwd(const addk_conv_args* a, int rows, void* stream);
int addk_pw_try_dgrad(const addk_conv_dgrad_args* a, int rows, void* stream);
// conv3.hip: halo-patch 3x3 stride-1 convolution (needs a->wpack); same return convention
int addk_c3_try_fwd(const addk_conv_args* a, int rows, void* stream);
int addk_c3_try_dgrad(const addk_conv_dgrad_args* a, int rows, void* stream);
