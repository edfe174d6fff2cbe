// conv3b.h — the split-bf16 halo-patch convolution kernel template (conv3b_body / conv3b_kernel) shared by the translation units
// that instantiate it: conv3b_tr3.hip (two-row 3x3 tiles), conv3b_tr5.hip (two-row 5x5), conv3b_row.hip (one-row tiles: wide dilations, half / quarter
// widths, the pointwise GEMM form) and conv3b_s2.hip (stride 2: stem2 forward and data gradient).  One file used to hold all ~140 instantiations and took
// three minutes to compile; host logic, weight packing and the fp32 halo kernel stay in conv3.hip.  Everything here has internal linkage.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };
constexpr int C3_BP = 128;                 // pixels per tile (one row segment)
constexpr int C3_BK = 16;                  // channels per chunk

struct C3K {
  addk_src src[ADDK_MAX_SRC];
  int nsrc;
  int N, H, W, dil;       // H, W: the OUTPUT map (tiles, epilogue)
  int HT;                   // tile rows in the walk: H, or ceil(H / 2) for two-row tiles
  int IH, IW;               // the input map (= H, W for the stride-1 launches)
  int Cn, ldy;
  float* y;
  const float* wp;          // packed weights of this launch
  const float* wsc;         // NP = 2 (split-fp16): [0] = 2^-kw, the inverse of the scale the pack pass applied to the weights (trailer of the pack buffer)
  long wp_blk;              // floats per column block in wp
  int nT;                   // chunks * taps
  const float* bias; const float* bias_n;
  double* slab; int slab_ld; int slab_rows;      // slab_rows: rows the caller allocated; the launch has min(rows, tiles) workgroups along x and zero-fills the rest
  addk_src dst; int accumulate;
  int vecY, red32;
  long P; int ntiles, spr;
  int ny;                   // conv3b_kernel: channel blocks per tile (the grid is one-dimensional: tile workgroups x ny)
  int st;                   // stride (host-side dispatch; 2 = the de-interleaved 3x3 forward of conv3b_kernel)
  int om, oro, oco, OHo, OWo;   // output pixel of tile-grid position (oh, ow): (om oh + oro, om ow + oco) in an OHo x OWo map (om = 1: the grid itself)
};

// ======================================================================================================================
// Split-bf16 form of the same halo-patch convolution: every fp32 operand is written as x = h + m + l with h, m, l in bf16
// (3 x 8 significand bits = the 24 of fp32, exactly) and a product is evaluated on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) as the sum of its six largest bf16 x bf16 terms
//     x*w  ~=  l*wh + h*wl + m*wm + m*wh + h*wm + h*wh          (dropped: m*wl, l*wm, l*wl  ~ 2^-26 relative)
// Measured (scripts/bf16_split_probe.hip, K = 2048, random magnitudes over 6 octaves): max |err| / sum|a*b| = 3.0e-7, rms 5.6e-8 —
// the same as the exact-fp32 v_mfma_f32_16x16x4_f32 chain (3.2e-7 / 6.8e-8) — at 6 x 16 = 96 matrix-pipe cycles per
// 16x16x32 block of fp32 work instead of 256: 2.5x the fp32 MFMA rate (374 vs 149 TFLOP/s from registers).  This is NP = 3 (--math bf16x6);
// NP = 2 is the split-FP16 form (two fp16 terms under exact power-of-two scales, three product terms: common.h, the default since round 5) — it took the
// place of the three-term split-bf16 form (h*wh + m*wh + h*wm: rms error 4.7e-7), whose speed it has at the six-term form's accuracy.
//
// Structure.  A block owns 128 pixels of one output row x (32 * WC) output channels; WAVE w owns channels [32w, 32w+32) of ALL
// 128 pixels (four 32x32 accumulator tiles): the pixel fragments are shared through LDS (which has the bandwidth: 128 B/clk
// per CU at 8 waves), the weight fragments are private to a wave and stream from L1/L2 (32 B/clk).  K is walked in
// 16-channel chunks = one MFMA k-step; per chunk the KS input rows are staged ONCE: fp32 loads, lazy BatchNorm/ReLU prologue,
// zero padding, then the split into planes — so the ~6 VALU operations per element of the split are paid once per staged
// element, not per use (each element feeds KS*KS * 6 * WC MFMA operands).  LDS image per plane: [row][pixel][16 ch] bf16 =
// 32 B per pixel, the two 16-byte halves of a pixel swapped when bit 3 of the pixel index is set: the ds_read_b128 of a
// 32-pixel fragment then touches all 64 banks once per 16-lane group for every tap shift (no padding needed).
// ======================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CB_PT = 4;                   // 32-pixel tiles per wave (128 pixels)
constexpr int cb_pwmax(int ks, bool bigd, int bpx = C3_BP, int st = 1) { return st == 2 ? 2 * bpx + 16 : ks > 10 ? bpx + 4 : ks == 1 ? bpx : ks == 3 ? (bigd ? bpx + 2 * 18 : bpx + 2 * 2) : bpx + 4 * 2; }

__device__ __forceinline__ unsigned bf16_hi(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_f(unsigned b) { return __uint_as_float(b << 16); }
// ======================================================================================================================
// [r5] NP = 2 is the SPLIT-FP16 form ("f16x3"): x = h + l with h = fp16(x), l = fp16(x - h) — 2 x 11 significand bits = 22 of fp32's 24 —
// and a product is the sum of its three largest terms on v_mfma_f32_32x32x16_f16 (l*wh + h*wl + h*wh; dropped: l*wl ~ 2^-22 relative, rms
// 2^-24).  Half the matrix instructions of the six-term bf16 form at fp32-class accuracy: the split error (rms 3e-9 of sum |a b| at K = 2736)
// is BELOW the rounding noise of an fp32 accumulation chain (1e-8) and 60x below the three-term bf16 form it replaces (1.9e-7).
// What fp16 lacks is RANGE (2^-14 .. 2^16), so both operands carry a power-of-two scale, exact in both directions:
//   * weights: the pack pass scales the tensor by 2^kw so that its largest magnitude lands in [2^14, 2^15) (c3_pack_amax_kernel, PackK.amax);
//   * activations / gradients: every staged 16-channel chunk of a tile publishes its largest magnitude (after the prologue) and the workgroup keeps a
//     RUNNING scale 2^k — the scale of the largest chunk seen so far in this tile; when a chunk arrives that would overflow it, the accumulators
//     are multiplied by the (exact) ratio and the scale drops.  Elements more than 2^17 below the running maximum lose low-part bits gradually
//     (absolute error <= max * 2^-40): they cannot matter to a sum that contains the maximum.  No tensor-wide pass, no producer-side bookkeeping.
//   The epilogue multiplies by 2^-k 2^-kw.  A NaN element stays out of the maxima (v_max drops it) and makes the outputs it touches NaN, as in fp32; an Inf element
//   enters them: the outputs it touches become NaN and the finite elements staged under that scale flush to zero — a non-finite loss either way.
// ======================================================================================================================
// 4 floats -> 4 bf16 per plane (8 bytes each); NP = 2: the two fp16 planes
template <int NP>
__device__ __forceinline__ void split4(const float4 v, uint2 (&pl)[NP]) {
  if constexpr (NP == 2) { split4h(v, pl); return; } else {
  float r[4] = {v.x, v.y, v.z, v.w};
  unsigned b[NP][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = r[e];
#pragma unroll
    for (int k = 0; k < NP; ++k) { b[k][e] = bf16_hi(x); x = x - bf16_f(b[k][e]); }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) pl[k] = make_uint2(b[k][0] | (b[k][1] << 16), b[k][2] | (b[k][3] << 16));
  }
}

// PH = 2 (the <= 64-channel launches): the block's waves are also split over the two 64-pixel halves of the tile — wave w owns
// channels [32 (w % WC), +32) of pixels [64 (w / WC), +64), two accumulator tiles — so that 4 waves share the staging work of
// a 64-channel block (2-wave blocks staged 13-22 slots per thread and spilled).
// BPX = 64: half-width pixel tiles (two accumulator tiles per wave) for launches that would otherwise put fewer than ~1.5
// blocks on a CU (ASPP at 64x128, the 80-channel cell convs at 63x127: 126-256 blocks of 128 pixels = one wave per SIMD or less).
// ST = 2 (stem2: 3x3, stride 2, pad 1, forward): output pixel ow reads input columns 2 ow - 1 + kw.  The patch row is staged
// DE-INTERLEAVED — even patch columns pj = 2 e at LDS position e (0..BPX), odd ones pj = 2 o + 1 at position OB + o — so that
// the fragment of tap kw is again 32 CONSECUTIVE positions (kw = 0: e = lp, kw = 1: o = lp, kw = 2: e = lp + 1) and the
// swizzle / bank picture of the stride-1 kernel holds unchanged.  Checkerboard: position idx carries the sign (-1)^(oh+ow0+idx),
// which is the output pixel's for kw = 0, 1 and its negative for kw = 2 — that tap's weights are packed negated.
#ifdef ADDK_C3B_DIAG
// diagnostic build (make CXXFLAGS+=-DADDK_C3B_DIAG, scripts/c3b_clock.sh): every workgroup adds its lifetime in shader-clock ticks (s_memtime)
// and in 100 MHz reference ticks (s_memrealtime): their ratio is the clock the CUs ran at INSIDE this kernel
__device__ unsigned long long g_c3b_diag[64][12];       // 64 slots: the workgroups' atomics do not queue on one L2 line; [4..9]: phase ticks of wave 0 (prologue, matrix phase, barrier, prologue+split+LDS stores, barrier, epilogue)
#define C3B_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define C3B_STAMP(v)
#endif
// TR = 2 (3x3 / 5x5, dilation d <= 2): the tile is TWO output rows of BPX pixels, d rows apart — oh0 and oh0 + d, whose input rows oh0 + (r - HK) d,
// r = 0..KS, overlap in KS - 1 of KS + 1 — with the accumulator count of one row of 2 BPX pixels: the patch holds KS + 1 input rows instead
// of 2 KS for the same outputs, 33 % (3x3) / 40 % (5x5) less staging (global loads, prologue, split, LDS writes) and HBM-side traffic.  The
// tile walk counts row PAIRS (p.HT; pair q -> oh0 = (q / d) 2d + q % d) and the epilogue masks second rows beyond the map.
// Reduce-scatter of 16 per-lane values over the 32 pixel lanes of a half-wave (lane bits 4..0): four halving exchanges (xor 16, 8, 4, 2) and a
// final pair add leave in v[0] of lane l the sum over the 32 lanes of value (l & 31) >> 1 — 15 + 1 cross-lane moves for 16 sums where an
// all-reduce butterfly per value takes 80.  Fixed order: bit-reproducible.
template <int N, int M>
__device__ __forceinline__ void rs32_step(float (&v)[16], const int lp32) {
  const bool b = (lp32 & M) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float keep = b ? v[i + N] : v[i], send = b ? v[i] : v[i + N];
    v[i] = keep + __shfl_xor(send, M);
  }
}
__device__ __forceinline__ void rs32(float (&v)[16], const int lp32) {
  rs32_step<8, 16>(v, lp32); rs32_step<4, 8>(v, lp32); rs32_step<2, 4>(v, lp32); rs32_step<1, 2>(v, lp32);      // (written as a loop over (N, M) the steps are not unrolled and v[] lands in scratch)
  v[0] += __shfl_xor(v[0], 1);
}

#ifndef ADDK_C3B_ABL
#define ADDK_C3B_ABL 0          // ablation study (scripts/c3b_ablation.sh): bit 0 no weight-fragment loads, 1 no prologue / split (raw bits stored), 2 no statistics,
#endif                          // 3 only the first chunk's patch is loaded, 4 one MFMA per (tap, tile) instead of NP*(NP+1)/2 — every ablated build computes WRONG numbers
template <int WC, int KS, int MODE, int NP, bool BIGD, int PH = 1, int BPX = C3_BP, int ST = 1, int TR = 1>
__device__ __forceinline__ void conv3b_body(const C3K& p, const int bx, const int gx, const int by) {      // workgroup bx of gx along x (tiles, slab row), channel block by
#ifdef ADDK_C3B_DIAG
  const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long dph[6] = {0, 0, 0, 0, 0, 0};
#endif
  static_assert(ST == 1 || (ST == 2 && KS == 3 && !BIGD && MODE == MODE_FWD), "stride 2: 3x3 forward only");
  // KS = 12 / 21 / 22: a 1x2 / 2x1 / 2x2 tap set anchored at its first tap (no centring) — the parity classes of the stride-2 data
  // gradient (c3b_s2_dgrad below); 1, 3, 5: the centred square kernels
  constexpr int KH_ = KS > 10 ? KS / 10 : KS, KW_ = KS > 10 ? KS % 10 : KS;
  static_assert(KS < 10 || (MODE == MODE_DGRAD && !BIGD && ST == 1), "anchored tap sets: data gradient only");
  constexpr int OB = BPX + 16;                                      // ST = 2: LDS position of the first odd patch column
  static_assert(TR == 1 || (TR == 2 && (KS == 3 || KS == 5) && !BIGD && ST == 1), "two-row tiles: centred kernels at dilation 1");
  constexpr int BC = 32 * WC, NTHR = 64 * WC * PH, PT = TR * BPX / 32 / PH, TPR = BPX / 32;      // TPR: 32-pixel tiles per tile row
  constexpr int PR_ = (KS > 10 ? KS / 10 : KS) + TR - 1;                                          // patch rows
  constexpr int TAPS = KH_ * KW_, HK = KS > 10 ? 0 : KS / 2;
  constexpr int PWP = cb_pwmax(KS, BIGD, BPX, ST);                         // LDS row pitch in pixels (compile time: tap offsets are immediates)
  constexpr int NS = (PR_ * PWP * 4 + NTHR - 1) / NTHR;           // 16-byte (4-channel) patch slots per thread, enumerated over the pitch grid
  static_assert(NS <= 32, "slot mask is 32 bits");
  constexpr int PLANE = PR_ * PWP * 2;                             // uint4 units per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* red = reinterpret_cast<double*>(smem);                  // [PH][BC][2] running statistics of this block (per pixel half)
  unsigned* wmax = reinterpret_cast<unsigned*>(smem);             // NP = 2: [waves] largest staged magnitude of the chunk on its way to LDS (aliases red[], which is written once, behind the last tile's barriers)
  uint4* Pl = reinterpret_cast<uint4*>(smem + ((PH * BC * 16 + 15) & ~15));   // [NP][KS][PWP][2] 16-byte halves
  uint2* Pl2 = reinterpret_cast<uint2*>(Pl);

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wave = wv % WC, wpx = wv / WC, lp32 = lane & 31, hh = lane >> 5;
  const int n0 = by * BC;
  const int d = p.dil;
  // statistics of this workgroup: after each tile's reduce-scatter (rs32) lane l holds the tile's sums of channel value (l & 31) >> 1 and adds them
  // here in double; red[] is written ONCE, after the last tile (it was 64 dependent LDS read-modify-writes by one lane per tile and wave:
  // profiles/r05_c3b_ablation.txt, 5-6 us per tile)
  double tot_a = 0.0, tot_b = 0.0;

  const int q = t & 3;
  // patch slot -> (patch row r, LDS position sp in the row, patch column pj = input column - first input column of the tile)
  auto slot_geo = [&](int k, int& r, int& sp, int& pj, bool& live) {
    const int pix = (t + NTHR * k) >> 2;
    r = pix / PWP; sp = pix - r * PWP;
    if (ST == 1) { pj = sp; live = r < PR_ && pj < BPX + (KW_ - 1) * p.dil; }
    else { const bool odd = sp >= OB; const int idx = odd ? sp - OB : sp; pj = 2 * idx + (odd ? 1 : 0); live = r < KH_ && (odd ? idx < BPX : idx <= BPX); }
  };
  // fragment read base per kernel column: pixel lane%32 + kw*d of patch row 0, the 16-byte half swizzled by bit 3 of the pixel
  // (tile j adds 32 pixels: bit 3 unchanged); patch row kh and tile j are immediate offsets
  int xb[KW_];
  // 32-pixel tile j of this wave: tile row wrow + jrow(j), column tile wcol + jcol(j) (4-wave blocks of <= 64 channels split the tiles over two wave pairs:
  // by rows when the tile has two, else by columns)
  const int wrow = (PH == 2 && TR == 2) ? wpx : 0, wcol = (PH == 2 && TR == 1) ? wpx * PT : 0;
  auto jrow = [](int j) { return PH == 1 ? j / TPR : 0; };
  auto jcol = [](int j) { return PH == 1 ? j % TPR : j; };
#pragma unroll
  for (int kw = 0; kw < KW_; ++kw) {
    const int pj = ST == 2 ? (kw == 1 ? OB + lp32 : lp32 + (kw >> 1)) : lp32 + kw * d;
    xb[kw] = pj * 2 + (hh ^ ((pj >> 3) & 1)) + (wrow * PWP + wcol * 32) * 2;
  }
  const uint4* wpl = reinterpret_cast<const uint4*>(p.wp) + ((long)by * p.wp_blk + (long)wave * NP * 64 + lane);
  const int nT = p.nT;
  const float winv = NP == 2 ? p.wsc[0] : 1.f;
  unsigned pmask = 0;                         // bit k: parity of (r*d + pj) of this thread's patch slot k (checkerboard sign, below)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int r, sp, pj; bool live;
    slot_geo(k, r, sp, pj, live);
    pmask |= (unsigned)((ST == 2 ? sp : r * d + pj) & 1) << k;
  }

  const int tpx = p.ntiles >> 3;
  const bool swz = (p.ntiles & 7) == 0 && p.ntiles >= 64;
  for (int tlin = bx; tlin < p.ntiles; tlin += gx) {
    const int tile = swz ? (tlin & 7) * tpx + (tlin >> 3) : tlin;
    const int rowid = tile / p.spr, sx = tile - rowid * p.spr;
    const int n = rowid / p.HT, prq = rowid - n * p.HT;
    const int oh = TR == 1 ? prq : (prq / d) * (2 * d) + prq % d;
    const int ow0 = sx * BPX;
    unsigned vmask = 0;
    const int pbase = (n * p.IH + oh * ST - HK * d) * p.IW + ow0 * ST - HK * d;
    const unsigned par0 = (unsigned)(oh + ow0);                 // parity of (ih + iw) of patch element (r, pj) = par0 + r*d + pj (the -2 HK d is even)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      int r, sp, pj; bool live;
      slot_geo(k, r, sp, pj, live);
      const int ih = oh * ST + (r - HK) * d, iw = ow0 * ST - HK * d + pj;
      const bool ok = live && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      vmask |= (ok ? 1u : 0u) << k;
    }

    f32x16 acc[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // Blocked accumulation (the <= 64-channel launches: stem1, the cells' 40-channel dilated convs — the layers every later layer
    // inherits its error from): the running sums are flushed into a second accumulator set after every 16-channel chunk, so a
    // rounding error grows with sqrt(MFMAs per chunk) + sqrt(chunks) instead of sqrt(all MFMAs) (stem1: 54 + 4 instead of 216) —
    // what a CPU library's blocked partial sums do (the even-size gradient deficit starts at stem1's K = 576 chain).
    constexpr bool BLK = PH == 2 && (KS == 3 || KS > 10) && !BIGD;          // (the 5x5 and wide-dilation variants have no registers left: 256 + scratch with a second set)
    f32x16 acc2[BLK ? PT : 1];
    if (BLK) {
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
    }

    float4 ra[NS];
    float4 pa = make_float4(1.f, 1.f, 1.f, 1.f), pb = zero4();
    bool prelu = false, pch = false;
    int kf = 0;                                   // NP = 2: exponent field of the tile's running operand scale 2^(kf - 127); 0 = no chunk seen yet
    auto load_patch = [&](int s_, int c0_) {      // branch-free: masked slots read the source base and are zeroed at store time
      const addk_src S = p.src[s_];
      const int c = c0_ + 4 * q;
      pch = c < S.C;
      prelu = S.relu != 0;
      pa = make_float4(1.f, 1.f, 1.f, 1.f); pb = zero4();
      if (S.a && pch) { pa = ld4(S.a + c); pb = ld4(S.b + c); }
      const float* sb = S.x + (pch ? c : 0);
      if ((ADDK_C3B_ABL & 8) && (s_ || c0_)) return;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        int r, sp, pj; bool live;
        slot_geo(k, r, sp, pj, live);
        const int po = ((vmask >> k) & 1u) ? pbase + r * d * p.IW + pj : 0;
        ra[k] = ld4(sb + (long)po * S.ld);
      }
    };
    // NP = 2 (split-fp16): the prologue runs BEFORE the barrier in front of the staging (prep_patch: ra[] <- prologue, zero padding, sign; the wave's largest
    // magnitude goes to wmax[]), and after it every thread folds the waves' maxima into the tile's running scale (update_scale) — no extra barrier
    auto prep_patch = [&]() {
      unsigned mx = 0;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 v = ra[k];
        v.x = fmaf(pa.x, v.x, pb.x); v.y = fmaf(pa.y, v.y, pb.y); v.z = fmaf(pa.z, v.z, pb.z); v.w = fmaf(pa.w, v.w, pb.w);
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = pch && ((vmask >> k) & 1u);
        const float sg = ok ? ((((pmask >> k) ^ par0) & 1u) ? -1.f : 1.f) : 0.f;
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
        ra[k] = v;
        const unsigned b = absbits4(v);
        mx = b > mx ? b : mx;
      }
      mx = wave_umax(mx);
      if (lane == 0) wmax[wv] = mx;
    };
    auto update_scale = [&]() {
      unsigned m = 0;
#pragma unroll
      for (int w = 0; w < WC * PH; ++w) { const unsigned b = wmax[w]; m = b > m ? b : m; }
      const int want = f16_scale_field(m);
      if (kf != 0 && want < kf) {                  // (workgroup-uniform) a larger chunk: what was accumulated at the finer scale moves to the new one, exactly
        const int rf = 127 + want - kf;
        const float r = rf > 0 ? __uint_as_float((unsigned)rf << 23) : 0.f;
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[j][e] *= r;
        if (BLK) {
#pragma unroll
          for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] *= r;
        }
      }
      if (kf == 0 || want < kf) kf = want;
    };
    auto store_patch = [&]() {                    // prologue, zero padding, split into planes
      const float sc = NP == 2 ? __uint_as_float((unsigned)kf << 23) : 1.f;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float4 v = ra[k];
        if (NP == 2) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        else if (!(ADDK_C3B_ABL & 2)) {
        v.x = fmaf(pa.x, v.x, pb.x); v.y = fmaf(pa.y, v.y, pb.y); v.z = fmaf(pa.z, v.z, pb.z); v.w = fmaf(pa.w, v.w, pb.w);
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        // zero padding and the checkerboard sign in one factor: 0 outside the image, -1 for input pixels of odd (ih + iw), else +1
        const bool ok = pch && ((vmask >> k) & 1u);
        const float sg = ok ? ((((pmask >> k) ^ par0) & 1u) ? -1.f : 1.f) : 0.f;
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
        }
        int r, sp, pj; bool live;
        slot_geo(k, r, sp, pj, live);
        if (r < PR_) {
          uint2 pl[NP];
          if (ADDK_C3B_ABL & 2) {
#pragma unroll
            for (int m = 0; m < NP; ++m) pl[m] = make_uint2(__float_as_uint(v.x) + m, __float_as_uint(v.z));
          } else
          split4<NP>(v, pl);
          const int slot = (r * PWP + sp) * 2 + ((q >> 1) ^ ((sp >> 3) & 1));
#pragma unroll
          for (int m = 0; m < NP; ++m) Pl2[(m * PLANE + slot) * 2 + (q & 1)] = pl[m];
        }
      }
    };
    // weight fragments of one tap: NP planes x 16 bytes per lane, the next tap's set is fetched while this tap's MFMAs issue
    auto load_w = [&](int T, uint4* dst) {
      if ((ADDK_C3B_ABL & 1) && T > 2) return;
      const int Tc = T < nT ? T : nT - 1;
      const uint4* src = wpl + (long)Tc * (WC * NP * 64);
#pragma unroll
      for (int m = 0; m < NP; ++m) dst[m] = src[m * 64];
    };
    // pixel fragment of (tap, tile j): lane reads 16 bytes of pixel (32 j + lane%32 + kw*d) in patch row kh, k half lane/32
    auto read_x = [&](int tap, int j, uint4* x) {
      const int kh = tap / KW_, kw = tap - kh * KW_;
      const uint4* b = Pl + xb[kw];
#pragma unroll
      for (int m = 0; m < NP; ++m) x[m] = b[m * PLANE + ((kh + jrow(j)) * PWP + jcol(j) * 32) * 2];        // the wave's own tile row / first column tile sit in xb[]
    };
    // The accumulation inside the bf16 MFMA is not symmetric: what falls below its internal guard bits is floored, not rounded, so
    // a result sits, on average, 0.17 rms errors BELOW the exact sum whatever the sign of the data (scripts/bf16_bias_probe.hip:
    // mean error -2e-7 at |sum| ~ 1 for K = 2304; same for 32x32x16 and 16x16x32 and any term order; the fp32 MFMA shows
    // none).  Per element that is below the rounding noise, but it is COHERENT: neighbouring pixels all err the same way, and
    // the layers behind (3x3 windows, sums over 10^5-10^6 pixels) respond to such a DC shift far more than to white noise
    // (whole-network frozen-BN gradients at 2x512x1024: 2.5x the error of the exact-fp32 kernels).  The kernel therefore
    // computes a CHECKERBOARD of signs: output pixel (oh, ow) is accumulated as (-1)^(oh+ow) * y.  Input pixels of odd
    // (ih + iw) are negated when they are staged (sign bits of the bf16 planes: exact), which puts (-1)^(oh+ow) * (-1)^((kh+kw) d)
    // on the operand of tap (kh, kw); the second factor is uniform per tap and is baked into the packed weights; the epilogue undoes the
    // pixel's sign.  The floor bias then alternates from pixel to pixel and averages out in every window and every sum.
    auto mma = [&](f32x16& c, const uint4* w, const uint4* x) {
      if constexpr (NP == 2) {          // split-fp16: l*wh + h*wl + h*wh, smallest terms first
        auto Wh = [&](int m) { return __builtin_bit_cast(f16x8, w[m]); };
        auto Xh = [&](int m) { return __builtin_bit_cast(f16x8, x[m]); };
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(1), Xh(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(0), Xh(1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wh(0), Xh(0), c, 0, 0, 0);
        return;
      }
      auto W = [&](int m) { return __builtin_bit_cast(bf16x8, w[m]); };
      auto X = [&](int m) { return __builtin_bit_cast(bf16x8, x[m]); };
      if (ADDK_C3B_ABL & 16) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(0), c, 0, 0, 0); return; }
      if (NP == 3) {          // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(2), X(0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(1), c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(1), X(0), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(1), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W(0), X(0), c, 0, 0, 0);
    };

    // Software pipeline, pinned with sched_barrier (left alone, the compiler sinks every prefetch down to its first use and the
    // wave then waits out a full L2 round trip per tap): weight fragments are fetched TWO taps ahead into a ring of three
    // register sets, pixel fragments one 32-pixel tile ahead, and the next chunk's patch is loaded to registers during tap 0.
    int s = 0, c0 = 0, T0 = 0;
    uint4 wr[3][NP], xr[2][NP];
    C3B_STAMP(dt_a);
    load_w(0, wr[0]);
    load_w(1, wr[1]);
    load_patch(0, 0);
    if (NP == 2) prep_patch();
    __syncthreads();                 // every wave is done with the previous tile's patch
    if (NP == 2) update_scale();
    store_patch();
    __syncthreads();
    C3B_STAMP(dt_b);
#ifdef ADDK_C3B_DIAG
    dph[0] += dt_b - dt_a;
#endif
    while (true) {
      int s2 = s, c2 = c0 + C3_BK;
      if (c2 >= p.src[s].C) { c2 = 0; ++s2; }
      const bool more = s2 < p.nsrc;
      C3B_STAMP(dt_0);
      read_x(0, 0, xr[0]);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        load_w(T0 + tap + 2, wr[(tap + 2) % 3]);
        if (tap == 0 && more) load_patch(s2, c2);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int nj = (j + 1) % PT, ntap = tap + (j + 1) / PT;
          const int cur = (tap * PT + j) & 1;                       // the two fragment buffers alternate over the (tap, tile) sequence (PT may be odd: 32-pixel tiles)
          if (ntap < TAPS) read_x(ntap, nj, xr[cur ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc[j], wr[tap % 3], xr[cur]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      T0 += TAPS;
      if (TAPS % 3) {                // the two sets fetched ahead sit in ring slots TAPS%3 and (TAPS+1)%3: rotate them to 0 and 1
        uint4 t0[NP], t1[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) { t0[m] = wr[TAPS % 3][m]; t1[m] = wr[(TAPS + 1) % 3][m]; }
#pragma unroll
        for (int m = 0; m < NP; ++m) { wr[0][m] = t0[m]; wr[1][m] = t1[m]; }
      }
      if (BLK) {
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { acc2[j][e] += acc[j][e]; acc[j][e] = 0.f; }
      }
      C3B_STAMP(dt_1);
      if (NP == 2 && more) prep_patch();
      __syncthreads();
      C3B_STAMP(dt_2);
#ifdef ADDK_C3B_DIAG
      dph[1] += dt_1 - dt_0; dph[2] += dt_2 - dt_1;
#endif
      if (!more) break;
      s = s2; c0 = c2;
      if (NP == 2) update_scale();
      store_patch();
      C3B_STAMP(dt_3);
      __syncthreads();
      C3B_STAMP(dt_4);
#ifdef ADDK_C3B_DIAG
      dph[3] += dt_3 - dt_2; dph[4] += dt_4 - dt_3;
#endif
    }
    C3B_STAMP(dt_e0);
    if (BLK) {
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[j] = acc2[j];
    }

    const float inv_run = NP == 2 ? __uint_as_float((unsigned)(254 - kf) << 23) : 1.f;
    // ---- epilogue: lane holds pixel (32 j + lane%32), channels n0 + 32 wave + 8 g + 4 (lane/32) + {0..3}, g = 0..3 ----
    const bool want_red = p.slab != nullptr && !(ADDK_C3B_ABL & 4);
    float s1[4][4], s2v[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[g][e] = 0.f; s2v[g][e] = 0.f; }
    // per-channel operands of the epilogue (the same for every pixel tile): bias (forward), the destination's lazy BatchNorm (data gradient)
    float4 eav[4], ebv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = n0 + wave * 32 + 8 * g + 4 * hh;
      const int nrem = p.Cn - c;
      eav[g] = MODE == MODE_FWD ? zero4() : make_float4(1.f, 1.f, 1.f, 1.f); ebv[g] = zero4();
      if (nrem <= 0) continue;
      if (MODE == MODE_FWD) {
        if (p.bias) eav[g] = ld4g(p.bias + c, nrem, false);
        if (p.bias_n) { const float4 b = ld4g(p.bias_n + (long)n * p.Cn + c, nrem, false); eav[g].x += b.x; eav[g].y += b.y; eav[g].z += b.z; eav[g].w += b.w; }
      } else if (p.dst.a) { eav[g] = ld4g(p.dst.a + c, nrem, p.vecY); ebv[g] = ld4g(p.dst.b + c, nrem, p.vecY); }
    }
    // Pixel tiles in groups of JG: the data gradient first REQUESTS everything it reads for the group (the destination's activation for the ReLU
    // mask, the gradient it accumulates into), then computes and stores.  One loop of load -> use -> store per (tile, quad) cannot be
    // pipelined by the compiler (the stores may alias the next loads): 8-16 dependent memory round trips per tile, 28-42 % of a wave's life
    // in the stem data gradients (profiles/r05_c3b_ablation.txt).
    constexpr int JG = PT < 2 ? PT : 2;
#pragma unroll
    for (int j0 = 0; j0 < PT; j0 += JG) {
      float4 xq[JG][4], oq[JG][4];
      if (MODE == MODE_DGRAD) {
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
          const int j = j0 + jj;
          if (j >= PT) continue;
          const int jr = wrow + jrow(j);
          const int lp = (wcol + jcol(j)) * 32 + lp32;
          const long pp = p.om == 1 ? ((long)n * p.H + oh + jr * d) * p.W + ow0 + lp : ((long)n * p.OHo + oh * p.om + p.oro) * p.OWo + (long)(ow0 + lp) * p.om + p.oco;
          const bool pin = ow0 + lp < p.W && oh + jr * d < p.H;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = n0 + wave * 32 + 8 * g + 4 * hh;
            const int nrem = p.Cn - c;
            xq[jj][g] = zero4(); oq[jj][g] = zero4();
            if (!pin || nrem <= 0) continue;
            xq[jj][g] = ld4g(p.dst.x + pp * p.dst.ld + c, nrem, p.vecY);
            if (p.accumulate) oq[jj][g] = ld4g(p.y + pp * p.ldy + c, nrem, p.vecY);
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) {
        const int j = j0 + jj;
        if (j >= PT) continue;
        const int jr = wrow + jrow(j);
        const int lp = (wcol + jcol(j)) * 32 + lp32;
        // output pixel: the tile grid's (oh, ow) itself, or (stride-2 data gradient) pixel (om oh + oro, om ow + oco) of the OHo x OWo map
        const long pp = p.om == 1 ? ((long)n * p.H + oh + jr * d) * p.W + ow0 + lp : ((long)n * p.OHo + oh * p.om + p.oro) * p.OWo + (long)(ow0 + lp) * p.om + p.oco;
        const bool pin = ow0 + lp < p.W && oh + jr * d < p.H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = n0 + wave * 32 + 8 * g + 4 * hh;
          const int nrem = p.Cn - c;
          if (!pin || nrem <= 0) continue;
          float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
          const bool odd = ((par0 + (unsigned)(jr * d + lp32)) & 1u) != 0;                                           // undo the checkerboard sign of pixel (oh, ow0 + 32 j + lane%32)
          if (NP == 2) {                 // ... and the two operand scales (exact powers of two, one after the other: their product may leave the fp32 range where the result does not)
            const float f = odd ? -inv_run : inv_run;
            v.x = v.x * f * winv; v.y = v.y * f * winv; v.z = v.z * f * winv; v.w = v.w * f * winv;
          } else if (odd) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
          if (MODE == MODE_FWD) {
            v.x += eav[g].x; v.y += eav[g].y; v.z += eav[g].z; v.w += eav[g].w;
            st4g(p.y + pp * p.ldy + c, v, nrem, p.vecY);
            if (want_red) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float f = (e < nrem) ? get4(v, e) : 0.f; s1[g][e] += f; s2v[g][e] += f * f; }
            }
          } else {
            const float4 x = xq[jj][g];
            float4 gq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xe = get4(x, e), ae = get4(eav[g], e), be = get4(ebv[g], e), dz = get4(v, e);
              const bool m = (e < nrem) && (!p.dst.relu || fmaf(ae, xe, be) > 0.f);
              set4(gq, e, m ? dz * ae : 0.f);
              if (want_red && m) { s1[g][e] += dz * xe; s2v[g][e] += dz; }
            }
            if (p.accumulate) { const float4 o = oq[jj][g]; gq.x += o.x; gq.y += o.y; gq.z += o.z; gq.w += o.w; }
            st4g(p.y + pp * p.ldy + c, gq, nrem, p.vecY);
          }
        }
      }
    }
    if (want_red) {          // a wave owns its 32 channels alone: reduce-scatter over the 32 pixel lanes, the owning lane adds in double
      float va[16], vb[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[4 * g + e] = s1[g][e]; vb[4 * g + e] = s2v[g][e]; }
      rs32(va, lp32); rs32(vb, lp32);
      tot_a += (double)va[0]; tot_b += (double)vb[0];
    }
#ifdef ADDK_C3B_DIAG
    dph[5] += __builtin_amdgcn_s_memtime() - dt_e0;
#endif
  }
#ifdef ADDK_C3B_DIAG
  if (t == 0) {
    unsigned long long* dslot = g_c3b_diag[(bx + 7 * by) & 63];
    atomicAdd(&dslot[0], __builtin_amdgcn_s_memtime() - diag_c0); atomicAdd(&dslot[1], __builtin_amdgcn_s_memrealtime() - diag_r0);
    atomicAdd(&dslot[2], 1ull);
    for (int i = 0; i < 6; ++i) atomicAdd(&dslot[4 + i], dph[i]);
  }
#endif
  if (p.slab) {
    __syncthreads();                 // every wave is done with the patch; red[] (disjoint from it) gets each channel's totals from the lane that owns them
    if (!(lp32 & 1)) {
      const int v = lp32 >> 1;
      double* r = red + ((wpx * BC) + wave * 32 + 8 * (v >> 2) + 4 * hh + (v & 3)) * 2;
      r[0] = tot_a; r[1] = tot_b;
    }
    __syncthreads();
    if (t < BC && n0 + t < p.Cn) {
      double* o = p.slab + ((long)bx * p.slab_ld + n0 + t) * 2;
      o[0] = red[2 * t] + (PH > 1 ? red[2 * (BC + t)] : 0.0); o[1] = red[2 * t + 1] + (PH > 1 ? red[2 * (BC + t) + 1] : 0.0);
      for (int r = bx + gx; r < p.slab_rows; r += gx) { double* z = p.slab + ((long)r * p.slab_ld + n0 + t) * 2; z[0] = 0.0; z[1] = 0.0; }      // rows no workgroup owns
    }
  }
}

template <int WC, int KS, int MODE, int NP, bool BIGD, int PH = 1, int BPX = C3_BP, int ST = 1, int TR = 1>
__global__ void __launch_bounds__(64 * WC * PH, 2) conv3b_kernel(const C3K p) {
  // [r5] one-dimensional grid of (tile workgroups) x (channel blocks) with the channel blocks of a tile NEXT to each other on ONE XCD: workgroups are dealt
  // round-robin over the 8 XCDs, so linear ids l and l + 8 share an L2 — with a two-dimensional grid the second channel block of a tile started after ALL
  // first ones and fetched the same input rows from HBM again (PMC: 1.97x the algorithmic bytes for the decoder conv, rounds 2-5)
  const int ny = p.ny, gx = (int)gridDim.x / ny, lin = (int)blockIdx.x;
  int bx, by;
  if (ny == 1) { bx = lin; by = 0; }
  else if ((gx & 7) == 0) { const int q = lin >> 3; by = q % ny; bx = (q / ny) * 8 + (lin & 7); }
  else { by = lin % ny; bx = lin / ny; }
  conv3b_body<WC, KS, MODE, NP, BIGD, PH, BPX, ST, TR>(p, bx, gx, by);
}
}  // namespace

// dispatch entry points of the instantiating translation units: 1 = launched, 0 = no instantiation for this shape
int c3b_run_tr3(const void* k, int wc, int rpx, int mode, int np, dim3 grid, size_t lds, hipStream_t st);
int c3b_run_tr5(const void* k, int wc, int rpx, int mode, int np, dim3 grid, size_t lds, hipStream_t st);
int c3b_run_row(const void* k, int wc, int ks, bool bigd, int bpx, int mode, int np, dim3 grid, size_t lds, hipStream_t st);
int c3b_run_s2f(const void* k, int wc, int bpx, int np, dim3 grid, size_t lds, hipStream_t st);
int c3b_run_s2d(const void* k4, int np, dim3 grid, size_t lds, hipStream_t st);
int c3n_run(const void* k, int ks, int mode, int np, dim3 grid, hipStream_t st);      // conv3n.hip: the <= 48-channel launches on 16-wide tiles
#ifdef ADDK_C3B_DIAG
void c3b_diag_tr3(unsigned long long* acc12); void c3b_diag_tr5(unsigned long long* acc12); void c3b_diag_row(unsigned long long* acc12); void c3b_diag_s2(unsigned long long* acc12);
#endif
