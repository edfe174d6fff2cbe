/*
 * addk — C ABI of the MI355X (gfx950) kernels behind the Auto-Dynamic-DeepLab hot path.
 *
 * The reference has no native interface: its boundary is a Python object protocol
 * (SURVEY.md §8b).  Each entry point below replaces one family of PyTorch op call
 * sites on that path; the reference file:line it replaces is cited per function.
 * The Python host (auto-dynamic-deeplab_amd/, imported as `addk`) binds these with
 * ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every tensor is fp32, NHWC ("pixel-major"): element (n,h,w,c) of a tensor with pixel
 *     stride `ld` lives at  base[((n*H + h)*W + w)*ld + c].  `ld >= C` lets a tensor be a
 *     channel slice of a wider concat buffer (producers write at a channel offset, so
 *     torch.cat on the path — ADD.py:92,112, aspp_train.py:57, decoder.py:26 — is free).
 *   - a "lazy" activation is (raw, a, b, relu): its value is relu?(a[c]*raw + b[c]).
 *     BatchNorm (+ReLU) is never materialised: the consumer applies it while staging
 *     its input tile (a == NULL means a=1,b=0).
 *   - all pointers are device pointers; nothing is allocated or freed; no host sync;
 *     every launch goes to `stream` (a hipStream_t passed as void*).
 *   - return value: 0 ok, <0 error (see addk_last_error()).
 *   - float4 paths need base pointers 16-byte aligned and ld, C, channel offsets % 4 == 0;
 *     otherwise a scalar path is taken automatically.
 */
#ifndef ADDK_H_
#define ADDK_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ADDK_MAX_SRC 12
#define ADDK_MAX_SLAB 32
#define ADDK_MAX_TERMS 4

#define ADDK_OK 0
#define ADDK_ERR_INVALID (-1)
#define ADDK_ERR_UNSUPPORTED (-2)
#define ADDK_ERR_HIP (-3)

const char* addk_last_error(void);
int addk_version(void);
/* Probes MFMA f32 16x16x4 fragment layout on the device: out[256] = A(16x4)·B(4x16) with
 * A[i][k] = i*4+k, B[k][j] = (k+1)*(j+2) (asymmetric).  Used by smoke tests. */
int addk_selftest_mfma(float* out256, void* stream);

/* One source of a virtually concatenated NHWC input. */
typedef struct addk_src {
  const float* x;   /* raw data at channel 0 of this source, pixel 0 */
  const float* a;   /* lazy-BN scale [C] or NULL */
  const float* b;   /* lazy-BN shift [C] or NULL */
  int32_t ld;       /* pixel stride (floats) */
  int32_t C;        /* channels of this source */
  int32_t relu;     /* ReLU after the affine */
  int32_t rs_hw;    /* 0: plain source.  (SH << 16) | SW: `x` is an [N, SH, SW] map that the consumer samples bilinearly
                       (F.interpolate(mode='bilinear'), align_corners=False: ADD.py:76-77,84-90) onto ITS OWN input grid while it
                       stages the tile — the resize in front of a 1x1 convolution never materialises.  The lazy affine / ReLU apply
                       AFTER the interpolation (an affine commutes with it; the ReLU is the consumer's own, operations.py:23).  Only
                       addk_conv_fwd honours it, and only for the shapes addk_conv_fwd_resample_ok() accepts. */
} addk_src;

/* BatchNorm statistics -> lazy affine (F.batch_norm training mode, batchnorm.py:51-53): argument block of addk_bn_finalize, also
 * embedded in the conv launches whose last workgroup does the finalize itself (csrc/bnfin.h). */
typedef struct addk_bn_finalize_args {
  const double* partial; /* fp64 [rows][C][2] (sum, sumsq) — or the all-reduced [1][C][2] */
  int32_t rows, C;
  double count;          /* number of values per channel (global batch under SyncBN) */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;   /* updated in place (NULL: skip) */
  float momentum, eps;
  float* a; float* b;            /* out: lazy affine  a = gamma*invstd, b = beta - mean*a */
  float* mean; float* invstd;    /* out: saved for backward */
} addk_bn_finalize_args;

/* ---------------------------------------------------------------------------------------
 * Dense convolution, implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32).
 * Replaces nn.Conv2d(groups=1) call sites: operations.py:24,38,53,57,91-92,109-110;
 * ADD.py:155,161,167,257; aspp_train.py:16-25; decoder.py:14,18,21, with the preceding
 * ReLU / BatchNorm of the producer fused as the input prologue and the per-channel
 * (sum, sum of squares) needed by the following training-mode BatchNorm fused as the
 * epilogue.  Weights are "OHWI": w[co*ldw + (kh*KW+kw)*cin_total + ci] — the memory of a
 * torch [O,I,KH,KW] tensor in channels_last format.  `pad` may be negative
 * (FactorizedReduce's shifted branch, operations.py:94,99).
 * ------------------------------------------------------------------------------------- */
typedef struct addk_conv_args {
  addk_src src[ADDK_MAX_SRC];
  int32_t nsrc;
  int32_t N, H, W;            /* input spatial size */
  int32_t OH, OW;             /* output spatial size */
  int32_t KH, KW, stride, pad, dil;
  int32_t Cout;
  int32_t ldw;                /* weight row stride (floats) */
  int32_t cin_total;          /* channels per tap in the weight layout */
  int32_t w_choff;            /* channel offset of src[0] inside a tap (sources are consecutive) */
  int32_t ldy;
  const float* w;
  float* y;                   /* raw output at channel 0 of the destination slice */
  const float* bias;          /* [Cout] or NULL (decoder.py:21) */
  const float* bias_n;        /* [N][Cout] per-image bias or NULL (ASPP image-pool branch folded
                                 into conv1, aspp_train.py:50-59) */
  double* stats;              /* fp64 [rows][stats_ld][2] partial (sum, sumsq) or NULL; rows = addk_conv_rows().
                                 Points at this conv's first channel inside the row. */
  int32_t stats_ld;           /* channels per slab row (>= Cout; FactorizedReduce's two convs share one BN) */
  int32_t _pad;
  float* wpack;               /* optional workspace of >= addk_conv_fwd_pack_floats() floats: lets wide 3x3 stride-1 'same'
                                 convolutions run on the halo-patch kernel (weights re-packed into MFMA fragment order
                                 per launch); NULL = generic kernel */
  int64_t wpack_floats;
  int32_t wpack_ready;        /* 1: wpack already holds this launch's packed weights (addk_conv_pack_batch ran since the
                                 last weight update); 0: the launch packs them itself */
  int32_t _pad2;
  float* rs_y;                /* src[0].rs_hw != 0 only: optional materialised copy of the interpolated src[0] (before the affine /
                                 ReLU), pixel stride rs_ldy — training writes it because the backward pass (weight gradient, ReLU
                                 mask, the resize's own gather backward) reads it; NULL at inference */
  int32_t rs_ldy;
  int32_t _pad3;
} addk_conv_args;
int addk_conv_fwd(const addk_conv_args* a, void* stream);
/* 1 when addk_conv_fwd takes this launch with src[i].rs_hw set (a 1x1 stride-1 convolution on the register-stationary or the
 * streaming-K pointwise kernel, pw.hip: ADD.py:84-90 `pre_preprocess` / `preprocess` behind F.interpolate), else 0: the caller
 * then runs addk_resize_fwd first.  Results are bit-identical either way (the same fp32 expressions in the same order). */
int addk_conv_fwd_resample_ok(const addk_conv_args* a);
/* floats of `wpack` this launch can use; 0 = the halo-patch kernel does not cover the shape */
int64_t addk_conv_fwd_pack_floats(const addk_conv_args* a);
/* Arithmetic of the wide k x k stride-1 contractions (the halo-patch kernels: forward, data gradient, weight gradient):
 *   0 = exact fp32 products on v_mfma_f32_16x16x4_f32;
 *   1 = "f16x3", split-fp16 (default since round 5): every fp32 operand is x = h + l with h = fp16(x), l = fp16(x - h) (2 x 11 = 22 of
 *       fp32's 24 significand bits) under an exact power-of-two scale (weights: per tensor, from an amax pass in front of the pack;
 *       activations / gradients: a running scale per tile, csrc/conv3b.h), and a product is the sum of its three largest terms on
 *       v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Split error rms 3e-9 of sum|a*b| at K = 2736 — below an fp32 accumulation chain's
 *       rounding noise (1e-8) — at half the matrix instructions of mode 2;
 *   2 = "bf16x6", split-bf16, six terms: x = h + m + l in bf16 (3 x 8 bits = fp32's 24, exact), the six largest bf16 x bf16 terms on
 *       v_mfma_f32_32x32x16_bf16 (rms 5.6e-8 vs the fp32 MFMA chain's 6.8e-8 of sum|a*b| at K = 2048);
 *   3 = "tail_x3": mode 1 in the exit heads only (launches with >= 192 output / gradient channels: ASPP and decoder forward,
 *       data gradient and weight gradient — decoder.py:14-21, aspp_train.py:16-25), mode 2 everywhere else.
 * The reference's own GPU path is 16-bit throughout (apex O1, train.py:145-165).
 * Every other kernel computes in fp32.  Process-wide; set before plans are built (packed-weight buffers are sized per mode).
 * Environment: ADDK_MATH=fp32|f16x3|bf16x6|tail_x3 (bf16x3: the old name of mode 1). */
int addk_set_conv_precision(int mode);
/* Split-bf16 modes: launches with fewer output channels than this stay on the exact fp32 kernel (default 0 = none, see
 * conv3.hip; 65 = round-2's first rule, kept for A/B runs).  c < 0 restores the default. */
int addk_set_split_min_channels(int c);
/* Specialised kernels that may replace the generic ones (all on by default; results agree to fp32 rounding).  The mask is
 * process-wide and is meant for tests and A/B timing; the environment (ADDK_PW=0, ADDK_C3=0, ADDK_WGRAD_H3=0, ADDK_DWTILE=0, ADDK_WGRAD_RS=0) sets the
 * initial value. */
#define ADDK_FAST_PW      1   /* register-stationary 1x1 convolution (pw.hip) */
#define ADDK_FAST_CONV3   2   /* halo-patch 3x3 stride-1 forward / data gradient (conv3.hip; needs wpack) */
#define ADDK_FAST_WGRAD3  4   /* halo-patch 3x3 stride-1 weight gradient (wgrad.hip) */
#define ADDK_FAST_DWTILE  8   /* LDS-tiled depthwise forward / backward, tiled logits-upsample backward */
#define ADDK_FAST_WGRAD_RS 16  /* register-streaming weight gradient of the narrow cell convs (wgrad.hip) */
int addk_set_fast_paths(int mask);
int addk_get_fast_paths(void);
int addk_get_conv_precision(void);
/* number of partial-statistics rows a launch with these shapes writes (<= 1024) */
int addk_conv_rows(int64_t P, int32_t Cout);

/* Data gradient of the same convolution with respect to ONE source (autograd of
 * nn.Conv2d + the ReLU/BN prologue):  g[p,c] (+)= da_mask * sum_{tap,co} dy[..]*w[..], where
 * da_mask = a[c] * (relu ? a[c]*x[p,c]+b[c] > 0 : 1).  Also emits the partial reductions
 * dA[c] = sum_p dz*mask*x, dB[c] = sum_p dz*mask that drive the producer BN's backward. */
typedef struct addk_conv_dgrad_args {
  const float* dy; int32_t lddy; int32_t Cout;     /* gradient of the raw conv output [N,OH,OW,Cout] */
  int32_t N, H, W, OH, OW, KH, KW, stride, pad, dil;
  const float* w; int32_t ldw, cin_total, w_choff; /* w_choff: channel offset of THIS source in a tap */
  addk_src dst;                /* the source whose gradient is produced (x,a,b,relu,ld,C) */
  float* g; int32_t ldg;       /* gradient wrt dst.x (raw), same geometry as dst */
  int32_t accumulate;          /* 0: overwrite g, 1: g += */
  double* dab;                 /* fp64 [rows][C][2] partial (dA,dB) or NULL; rows = addk_conv_rows(N*H*W, C) */
  float* wpack;                /* optional packed-weight workspace (see addk_conv_args.wpack) */
  int64_t wpack_floats;
  int32_t wpack_ready;         /* see addk_conv_args.wpack_ready */
  int32_t _pad2;
} addk_conv_dgrad_args;
int addk_conv_dgrad(const addk_conv_dgrad_args* a, void* stream);
int64_t addk_conv_dgrad_pack_floats(const addk_conv_dgrad_args* a);
/* Weight packs hoisted out of the step: fill one opaque descriptor (addk_conv_pack_desc_bytes() bytes, host memory) per
 * launch that has a wpack workspace, upload the array and run addk_conv_pack_batch once per step before the first
 * launch; those launches then set wpack_ready = 1. */
/* Batched pointwise convolutions: mutually independent 1x1 launches (one dependency level of the cell DAG) that map to the
 * same kernel variant run as ONE launch.  key >= 0 names the variant (-1: the launch is not covered); prepare() fills a
 * host blob of kernel descriptors (host_blob = NULL: returns its size) and meta[4]; the caller uploads the blob once and
 * replays addk_conv_batch_run.  Same arithmetic as the single launches. */
int addk_conv_fwd_batch_key(const addk_conv_args* a);
int addk_conv_dgrad_batch_key(const addk_conv_dgrad_args* a);
int64_t addk_conv_fwd_batch_prepare(const addk_conv_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int64_t addk_conv_dgrad_batch_prepare(const addk_conv_dgrad_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_conv_batch_run(const void* dev_blob, const int64_t* meta, void* stream);
int64_t addk_conv_pack_desc_bytes(void);
int addk_conv_fwd_pack_desc(const addk_conv_args* a, void* host_desc);
int addk_conv_dgrad_pack_desc(const addk_conv_dgrad_args* a, void* host_desc);
int addk_conv_pack_batch(const void* dev_descs, int32_t n, void* stream);

/* Weight gradient for ONE source: dw[co][tap][w_choff+ci] = sum_p dy[p,co] * z[p@tap,ci],
 * z = relu?(a*x+b).  Deterministic split-P: partial tiles go to `ws`, then are reduced into
 * dw (accumulate: the shared ASPP/decoder head is used once per exit — SURVEY Q4). */
typedef struct addk_conv_wgrad_args {
  const float* dy; int32_t lddy; int32_t Cout;
  int32_t N, H, W, OH, OW, KH, KW, stride, pad, dil;
  addk_src src;
  float* dw; int32_t ldw, cin_total, w_choff;
  int32_t accumulate;
  float* ws; int64_t ws_floats;  /* workspace, >= addk_conv_wgrad_ws() floats */
} addk_conv_wgrad_args;
int addk_conv_wgrad(const addk_conv_wgrad_args* a, void* stream);
int64_t addk_conv_wgrad_ws(int64_t P, int32_t Cout, int32_t C, int32_t taps);

/* Batched weight gradients.  The weight gradients of a backward pass are mutually independent and, for the 40-160
 * channel cell convolutions, individually far too small to fill 256 CUs (HBM-ideal 5 us, launch-latency bound at
 * 30-50 us): `n` of them that share a tile configuration run as ONE launch driven by a device-resident work list, plus
 * one batched reduce.  cfg = {kind, cty, ctz, blocks}; prepare() fills a host blob (descriptors + work lists) to be
 * copied to the device once at plan time (call with host_blob == NULL to get its size); meta[8] is host-side. */
int addk_conv_wgrad_config(const addk_conv_wgrad_args* a, int32_t* cfg);
int64_t addk_conv_wgrad_batch_prepare(const addk_conv_wgrad_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_conv_wgrad_batch_run(const void* dev_blob, const int64_t* meta, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused SepConv half (reference modeling/operations.py:51-53 and :55-57): ReLU / lazy BatchNorm prologue -> depthwise
 * K x K (stride 1, dilation 1, 'same' padding) -> pointwise 1x1 on the matrix cores -> BatchNorm statistics, ONE launch
 * over 2-D LDS tiles (csrc/sepf.hip): the depthwise output never makes a round trip through HBM.  In training `t` receives
 * the depthwise output (the backward pass reads it: pointwise weight gradient, depthwise backward) and, when `fin.a` is set,
 * the LAST workgroup of the launch turns the statistics into the lazy affine (a, b) — no bn_finalize launch (`fin.partial`,
 * `fin.rows`, `fin.C` are ignored: the launch knows its own slab; `fin_counter` is a zero-initialised workspace of
 * addk_bn_fin_ws_bytes(addk_sep_rows(a), stats_ld) bytes owned by this BatchNorm call — ticket counters and group rows of the
 * two-level reduction — whose counters the launch leaves at zero again).  In inference t = NULL and the epilogue can apply the op's own
 * (frozen) BatchNorm and add the other branches of the cell block (ADD.py:108):
 *     y = ea[c]*acc + eb[c] + sum_i relu_i?(a_i*term_i + b_i)        (ea == NULL: plain y = acc)
 * Covered shapes: K in {3,5}, C == Cout in (32, 48] or (64, 80], 16-byte aligned tensors; addk_sep_fwd_supported says so.
 * ------------------------------------------------------------------------------------- */
typedef struct addk_sep_args {
  addk_src src;                     /* lazy input of the depthwise conv (x, a, b, relu) */
  int32_t N, H, W;                  /* output size == input size */
  int32_t K;                        /* depthwise kernel size (3 or 5) */
  int32_t Cout; int32_t ldw;        /* pointwise: Cout x src.C weights, row stride ldw */
  const float* dw_w;                /* depthwise weights [src.C][K*K] */
  const float* pw_w;
  float* y; int32_t ldy;
  int32_t ldt; float* t;            /* depthwise output [P][ldt] or NULL */
  void* stats; int32_t stats_ld;    /* fp64 (sum, sumsq) partial slab [stats_rows][stats_ld][2] or NULL */
  int32_t stats_rows;               /* rows the caller allocated: >= addk_sep_rows(a); rows no workgroup owns are zero-filled */
  int32_t nterm;                    /* inference epilogue: number of extra terms (0..ADDK_MAX_TERMS) */
  const float* ea; const float* eb; /* inference epilogue: own affine (both NULL = none) */
  addk_src term[ADDK_MAX_TERMS];
  addk_bn_finalize_args fin;        /* fused finalize of the statistics (fin.a == NULL: none) */
  void* fin_counter;
} addk_sep_args;
int64_t addk_bn_fin_ws_bytes(int32_t nblocks, int32_t ld);
int addk_sep_rows(const addk_sep_args* a);               /* slab rows (= workgroups) of the fused launch; 0: not covered */
int addk_sep_fwd_supported(const addk_sep_args* a);      /* 1: the fused kernel covers this launch */
int addk_sep_fwd(const addk_sep_args* a, void* stream);
/* table-driven batch of the fused halves of one dependency level (same protocol as addk_conv_fwd_batch_prepare / addk_conv_batch_run) */
int addk_sep_fwd_batch_key(const addk_sep_args* a);
int64_t addk_sep_fwd_batch_prepare(const addk_sep_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_sep_batch_run(const void* dev_blob, const int64_t* meta, void* stream);

/* Fused BACKWARD of a SepConv half (csrc/sepb.hip): pointwise data gradient (dt = W^T dy, matrix cores) and depthwise backward
 * (dx, depthwise weight-gradient partials, (dA, dB) of the input's lazy BatchNorm) in one launch; dt stays on chip.  dy is the
 * gradient wrt the pointwise output with the BatchNorm backward already applied (addk_bn_bwd_apply).  The pointwise WEIGHT
 * gradient is not part of it (addk_conv_wgrad on dy and the stored depthwise output).  `ws` ([rows][C][K*K] floats) and `dab`
 * ([rows][C][2] fp64) get one row per workgroup, rows = addk_sep_bwd_rows(a); ws is reduced by addk_dw_wreduce_batch. */
typedef struct addk_sep_bwd_args {
  const float* dy; int32_t lddy;
  int32_t N, H, W, K;               /* stride 1, dilation 1, pad K/2: output size == input size */
  addk_src src;                     /* forward input of the depthwise conv (x, a, b, relu) */
  int32_t Cout; int32_t ldw;        /* pointwise weights [Cout][src.C], row stride ldw; Cout == src.C */
  const float* dw_w; const float* pw_w;
  float* g; int32_t ldg; int32_t accumulate;   /* gradient wrt src.x (NULL: skip) */
  double* dab;                      /* or NULL */
  float* ws;
} addk_sep_bwd_args;
int addk_sep_bwd_rows(const addk_sep_bwd_args* a);
int addk_sep_bwd(const addk_sep_bwd_args* a, void* stream);
int addk_sep_bwd_batch_key(const addk_sep_bwd_args* a);
int64_t addk_sep_bwd_batch_prepare(const addk_sep_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_sep_bwd_batch_run(const void* dev_blob, const int64_t* meta, void* stream);

/* ---------------------------------------------------------------------------------------
 * Depthwise k x k convolution (groups == C), stride 1 or 2, pad k/2*dil: the depthwise
 * halves of SepConv (operations.py:52,56).  w is [C][KH*KW] (torch [C,1,KH,KW]).
 * ------------------------------------------------------------------------------------- */
typedef struct addk_dw_args {
  addk_src src;
  int32_t N, H, W, OH, OW, KH, KW, stride, pad, dil;
  const float* w;
  float* y; int32_t ldy;
} addk_dw_args;
int addk_dw_fwd(const addk_dw_args* a, void* stream);

typedef struct addk_dw_bwd_args {
  const float* dy; int32_t lddy;
  int32_t N, H, W, OH, OW, KH, KW, stride, pad, dil;
  addk_src src;                 /* forward input (x,a,b,relu) */
  const float* w;
  float* g; int32_t ldg; int32_t accumulate;   /* gradient wrt src.x (may be NULL: skip dgrad) */
  double* dab;                  /* fp64 [rows][C][2] or NULL; rows = addk_dw_rows() */
  float* dw; int32_t dw_accumulate;            /* [C][KH*KW] */
  float* ws;                    /* [rows][C][KH*KW] partial weight gradients */
  int32_t defer_wreduce;        /* 1: leave the partials in ws; the caller reduces them later with addk_dw_wreduce_batch */
  int32_t _pad;
} addk_dw_bwd_args;
int addk_dw_bwd(const addk_dw_bwd_args* a, void* stream);
/* Deferred reduction of the depthwise weight-gradient partials of MANY convolutions in one launch (they are mutually
 * independent and only needed by the optimizer): table entry i reduces ws_i [rows_i][n_i] into dw_i [n_i]. */
typedef struct addk_dw_wreduce_item { const float* ws; float* dw; int32_t rows, n, accumulate, _pad; } addk_dw_wreduce_item;
int addk_dw_wreduce_batch(const addk_dw_wreduce_item* dev_items, int32_t n_items, void* stream);
/* Batched depthwise launches (same scheme as addk_conv_*_batch_*): mutually independent depthwise convs of one dependency
 * level that run on the LDS-tiled kernel of one size share ONE launch.  key >= 0 names the variant (-1: not covered; the
 * backward form needs defer_wreduce = 1); prepare() fills the host blob of kernel descriptors and meta[5]. */
int addk_dw_fwd_batch_key(const addk_dw_args* a);
int addk_dw_bwd_batch_key(const addk_dw_bwd_args* a);
int64_t addk_dw_fwd_batch_prepare(const addk_dw_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int64_t addk_dw_bwd_batch_prepare(const addk_dw_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_dw_batch_run(const void* dev_blob, const int64_t* meta, void* stream);
int addk_dw_rows(int64_t P, int32_t C);

/* ---------------------------------------------------------------------------------------
 * BatchNorm statistics (F.batch_norm training mode, batchnorm.py:51-53; eps=1e-5, mom=0.1).
 * ------------------------------------------------------------------------------------- */
int addk_bn_finalize(const addk_bn_finalize_args* a, void* stream);
/* n independent BatchNorms in one launch; dev_table = device array of n argument structs, max_C = largest C among them */
int addk_bn_finalize_batch(const addk_bn_finalize_args* dev_table, int32_t n, int32_t max_C, void* stream);

/* sum the rows of a partial slab into out[C][2] (the vector that is all-reduced over RCCL) */
int addk_slab_reduce(const double* partial, int32_t rows, int32_t C, double* out, void* stream);
typedef struct addk_slab_reduce_item { const double* partial; double* out; int32_t rows, C; } addk_slab_reduce_item;
int addk_slab_reduce_batch(const addk_slab_reduce_item* dev_table, int32_t n, int32_t max_C, void* stream);

/* eval mode: a = gamma/sqrt(running_var+eps), b = beta - running_mean*a */
int addk_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                        float eps, int32_t C, float* a, float* b, void* stream);

/* The same for every BatchNorm of an inference plan in ONE launch: `dev_table` is a device array of n entries
 * {gamma, beta, running_mean, running_var, a, b (pointers), int32 C, float eps} (56 bytes each). */
int addk_bn_eval_affine_batch(const void* dev_table, int32_t n, void* stream);

/* Backward of the statistics + affine:  given the partial (dA,dB) slabs of every consumer,
 *   dgamma (+)= invstd*(dA - mean*dB), dbeta (+)= dB,
 *   dmean_tot = -a*dB - 2*mean*dvar,  dvar = -0.5*gamma*(dA-mean*dB)*invstd^3,
 *   c1 = dmean_tot/count, c2 = 2*dvar/count  so that  dx_raw = G + c1 + c2*x.
 * With SyncBN, `dmv` receives (dmean_tot, dvar) per channel for the cross-rank all-reduce
 * and addk_bn_bwd_coeffs_from_dmv() finishes the job. */
typedef struct addk_bn_bwd_args {
  const double* slab[ADDK_MAX_SLAB]; int32_t rows[ADDK_MAX_SLAB]; int32_t nslab;
  int32_t C; double count;
  const float* gamma; const float* mean; const float* invstd; const float* a;
  float* dgamma; float* dbeta; int32_t accumulate;
  float* c1; float* c2;          /* out (NULL when dmv is used) */
  float* dmv;                    /* out [C][2] (dmean_tot, dvar) or NULL */
  int32_t centered; int32_t _pad; /* 1: c1 (dmv[0]) leaves out the -2*mean*dvar part: the consumer applies c2*(x - mean) */
} addk_bn_bwd_args;
int addk_bn_bwd(const addk_bn_bwd_args* a, void* stream);
int addk_bn_bwd_batch(const addk_bn_bwd_args* dev_table, int32_t n, int32_t max_C, void* stream);
int addk_bn_bwd_coeffs_from_dmv(const float* dmv, int32_t C, double count, float* c1, float* c2, void* stream);
typedef struct addk_bn_coeffs_item { const float* dmv; float* c1; float* c2; double count; int32_t C, _pad; } addk_bn_coeffs_item;
int addk_bn_bwd_coeffs_batch(const addk_bn_coeffs_item* dev_table, int32_t n, int32_t max_C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Elementwise "materialise" kernels.
 * affine_sum: out[p,c] = relu?( sum_i (a_i[c]*x_i[p,c] + b_i[c]) )  — the 2-way branch sum of a
 *   cell block (ADD.py:108) written straight into its slot of the concat buffer (ADD.py:112).
 * ------------------------------------------------------------------------------------- */
typedef struct addk_affine_sum_args {
  addk_src term[ADDK_MAX_TERMS]; int32_t nterm;   /* term.relu applies per term */
  int64_t P; int32_t C;
  float* out; int32_t ldo; int32_t relu_out; int32_t accumulate;
} addk_affine_sum_args;
int addk_affine_sum_fwd(const addk_affine_sum_args* a, void* stream);

/* backward: for every term i with g_i != NULL:  g_i (+)= a_i*mask_i*dout,  dab_i = partial (sum dout*mask*x_i, sum dout*mask) */
typedef struct addk_affine_sum_bwd_args {
  addk_src term[ADDK_MAX_TERMS]; int32_t nterm;
  int64_t P; int32_t C;
  const float* dout; int32_t lddo;
  const float* out; int32_t ldo; int32_t relu_out;   /* forward output, needed only when relu_out */
  float* g[ADDK_MAX_TERMS]; int32_t ldg[ADDK_MAX_TERMS]; int32_t accumulate[ADDK_MAX_TERMS];
  double* dab[ADDK_MAX_TERMS];    /* fp64 [rows][C][2] partials, rows = addk_ew_rows(P, C) */
} addk_affine_sum_bwd_args;
int addk_affine_sum_bwd(const addk_affine_sum_bwd_args* a, void* stream);
int addk_ew_rows(int64_t P, int32_t C);

/* dy[p,c] = g[p,c] + c1[c] + c2[c]*(x[p,c] - mean[c])   (BN backward applied to the accumulated gradient;
 * mean/c1/c2 may be NULL = 0).  The centred form (addk_bn_bwd_args.centered) keeps c2*x from cancelling against a c1 that
 * carries -c2*mean: the reference subtracts the mean first (batchnorm.py:51-53 / ATen batch_norm_backward).  out may alias g. */
int addk_bn_bwd_apply(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* mean,
                      const float* c1, const float* c2, int64_t P, int32_t C, float* out, int32_t ldo,
                      void* stream);
/* the same for n independent tensors in one launch (vector-aligned items only: 16-byte pointers, ld % 4 == 0, C % 4 == 0;
 * out[p,c] = g[p,c] + c1[c] + c2[c]*(x[p,c] - mean[c]), mean may be NULL */
typedef struct addk_bn_apply_item {
  const float* g; const float* x; const float* c1; const float* c2; const float* mean; float* out; int64_t P; int32_t ldg, ldx, ldo, C;
} addk_bn_apply_item;
int addk_bn_bwd_apply_batch(const addk_bn_apply_item* dev_table, int32_t n, int64_t max_P, void* stream);

/* ---------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=False, no antialias (F.interpolate call sites ADD.py:76-77,
 * 84,89,317; decoder.py:24,28).  Optional lazy prologue on the input; NCHW output / NCHW
 * gradient input for the final logits (decoder.py:28) so the caller sees [N,C,H,W].
 * ------------------------------------------------------------------------------------- */
typedef struct addk_resize_args {
  addk_src src; int32_t N, H, W, OH, OW;
  float* y; int32_t ldy;        /* NHWC destination (ignored when nchw_out) */
  int32_t nchw_out;             /* 1: y is [N,C,OH,OW] contiguous */
} addk_resize_args;
int addk_resize_fwd(const addk_resize_args* a, void* stream);

typedef struct addk_resize_bwd_args {
  const float* dy; int32_t lddy; int32_t nchw_in;   /* gradient of the resized tensor */
  const float* dy_scale;        /* optional device scalar multiplying dy (loss scale) */
  addk_src src; int32_t N, H, W, OH, OW;
  float* g; int32_t ldg; int32_t accumulate;
  double* dab;                  /* fp64 [rows][C][2] or NULL (only with a lazy+relu prologue) */
} addk_resize_bwd_args;
int addk_resize_bwd(const addk_resize_bwd_args* a, void* stream);
/* Batched form: the mutually independent resize backwards of one dependency level (the up to ten resized dense inputs of a cell,
 * ADD.py:84-90) as ONE launch.  key >= 0: the launch runs on the table-driven kernel with that variant (equal keys may share a batch);
 * prepare(host_blob = NULL) returns the blob size; meta[8] carries the launch geometry from prepare to run. */
int addk_resize_bwd_batch_key(const addk_resize_bwd_args* a);
int64_t addk_resize_bwd_batch_prepare(const addk_resize_bwd_args* a, int32_t n, void* host_blob, int64_t blob_bytes, int64_t* meta);
int addk_resize_bwd_batch_run(const void* dev_blob, const int64_t* meta, void* stream);

/* ---------------------------------------------------------------------------------------
 * Global average pool with lazy prologue (AdaptiveAvgPool2d(1), aspp_train.py:13,50; ADD.py:506).
 * y[n,c] = mean_hw relu?(a*x+b).  Backward: g (+)= mask*a*dy[n,c]/(H*W).
 * ------------------------------------------------------------------------------------- */
/* ws: >= N * addk_ew_rows(HW, C) * C floats of scratch; mean=0 returns the plain per-image sum
 * (used for bias gradients). */
int addk_gap_fwd(const addk_src* src, int32_t N, int32_t HW, float* y, int32_t ldy, float* ws, int32_t mean, void* stream);
int addk_gap_bwd(const addk_src* src, int32_t N, int32_t HW, const float* dy, int32_t lddy,
                 float* g, int32_t ldg, int32_t accumulate, double* dab, void* stream);

/* 3x3 pooling primitives of the registry (operations.py:9-10; cold on every shipped genotype).
 * mode 0: max, 1: avg with count_include_pad=False.  pad=1. */
int addk_pool3_fwd(const addk_src* src, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW,
                   int32_t stride, int32_t mode, float* y, int32_t ldy, void* stream);
int addk_pool3_bwd(const addk_src* src, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW,
                   int32_t stride, int32_t mode, const float* dy, int32_t lddy,
                   float* g, int32_t ldg, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layout transforms at the model boundary.
 * ------------------------------------------------------------------------------------- */
int addk_nchw_to_nhwc(const float* x, int32_t N, int32_t C, int64_t HW, float* y, int32_t ldy, void* stream);
int addk_nhwc_to_nchw(const addk_src* src, int32_t N, int64_t HW, float* y, void* stream);
/* gradient of nhwc_to_nchw incl. the lazy prologue: g (+)= mask*a*dy_nchw, dab partials */
int addk_nchw_grad_to_nhwc(const float* dy, const addk_src* src, int32_t N, int64_t HW,
                           float* g, int32_t ldg, int32_t accumulate, double* dab, void* stream);

/* ---------------------------------------------------------------------------------------
 * Softmax cross-entropy over NCHW logits (nn.CrossEntropyLoss(weight, ignore_index), train.py:70,231).
 * loss_out[0] += scale * sum_valid w[t]*nll / sum_valid w[t];  dlogits = scale*w[t]*(softmax-onehot)/wsum.
 * `wsum` (device, 1 float) is produced by addk_ce_count.  target is int64 [N,H,W].
 * `ws` is addk_ce_ws_floats() floats of scratch (per-block partial sums, reduced in a fixed order).
 * ------------------------------------------------------------------------------------- */
int addk_ce_count(const int64_t* target, int64_t n, const float* class_w, int32_t ignore_index,
                  int32_t num_classes, float* wsum, float* ws, void* stream);
int addk_ce_fwd_bwd(const float* logits, const int64_t* target, int32_t N, int32_t C, int64_t HW,
                    const float* class_w, int32_t ignore_index, const float* wsum, float scale,
                    float* loss_out, float* dlogits, float* ws, void* stream);
int64_t addk_ce_ws_floats(int32_t N, int64_t HW);

/* Fused form for the training step: the bilinear up-sampling of the decoder's logits (decoder.py:28,
 * F.interpolate(mode='bilinear', align_corners=False)) and the cross-entropy on it (train.py:70,231) in one pass over the
 * LOW-resolution NHWC logits [N,H,W,C] — loss_out[0] += scale * CE(upsample(logits) -> [N,C,OH,OW], target), and
 * g (+)= d loss / d logits.  The [N,C,OH,OW] tensor and its gradient are never materialised.  Deterministic (gather, fixed
 * summation order).  addk_ce_upsample_supported() is 0 for shapes the kernel does not take (C != 19, more than 16 output
 * rows per input row): the caller then runs addk_resize_fwd + addk_ce_fwd_bwd + addk_resize_bwd. */
typedef struct {
  const float* logits; int32_t ld;   /* NHWC, pixel stride ld >= C */
  int32_t N, H, W, C, OH, OW;
  const int64_t* target;             /* [N,OH,OW] */
  const float* class_w;              /* [C] or NULL */
  int32_t ignore_index;
  const float* wsum;                 /* device scalar from addk_ce_count */
  float scale;
  float* loss_out;                   /* device scalar, accumulated */
  float* g; int32_t ldg; int32_t accumulate;
  float* ws;                         /* addk_ce_upsample_ws_floats() floats */
} addk_ce_upsample_args;
int addk_ce_upsample_supported(int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t C);
int64_t addk_ce_upsample_ws_floats(int32_t N, int32_t H, int32_t W);
int addk_ce_upsample_fwd_bwd(const addk_ce_upsample_args* a, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused SGD (torch.optim.SGD(momentum, weight_decay, nesterov), train.py:126) on a flat buffer.
 *   d = g*gscale + wd*p;  buf = first ? d : mom*buf + d;  p -= lr*(nesterov ? d + mom*buf : buf)
 * lr is read from device memory so a captured graph can be replayed with a new poly-LR value.
 * ------------------------------------------------------------------------------------- */
int addk_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* lr_dev, float momentum,
                  float weight_decay, int32_t nesterov, int32_t first, float gscale, void* stream);

/* ---------------------------------------------------------------------------------------
 * Earlier-Decision-Maker head in ONE launch (ADD.py:502-525, the gate of dynamic_inference ADD.py:420-423):
 *   relu?(a*x+b) -> conv 3x3 stride 2 pad 1, C -> 128, no bias -> ReLU -> global average pool -> Linear 128-64 -> ReLU
 *   -> Linear 64-32 -> ReLU -> Linear 32-1.   out[n * ldo] = confidence of image n.
 * conv_w: [128][3][3][C] (a torch [128,C,3,3] tensor in channels_last memory); w1 [64][128], w2 [32][64], w3 [1][32] row-major
 * (torch Linear weights), b1/b2/b3 their biases.  ws: addk_edm_head_ws_bytes() bytes, ZERO-initialised once (the kernel leaves its
 * ticket word at zero, so the launch can be replayed from a hipGraph).  out_host: optional host-mapped (pinned) word the confidence is
 * ALSO written to — the host gate then needs no device-to-host copy, only the completion of the launch.
 * ------------------------------------------------------------------------------------- */
typedef struct addk_edm_args {
  addk_src src; int32_t N, H, W;
  int32_t ldo;                /* out[n * ldo] = confidence of image n (an [N,1,1,1] NHWC tensor with a padded pixel stride); >= 1 */
  const float* conv_w;
  const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;
  float* out; float* out_host;
  void* ws;
} addk_edm_args;
int64_t addk_edm_head_ws_bytes(int32_t N, int32_t H, int32_t W);
int addk_edm_head_supported(const addk_edm_args* a);
int addk_edm_head(const addk_edm_args* a, void* stream);

/* misc */
int addk_fill(float* p, int64_t n, float v, void* stream);
/* Earlier-Decision-Maker support (operations.py:161-170): per-pixel normalised entropy summed */
int addk_entropy_sum(const float* logits_nchw, int32_t N, int32_t C, int64_t HW, float* out1, float* ws, void* stream);
/* argmax over channels of NCHW logits -> int64 [N,HW], and confusion matrix accumulation
 * (utils/metrics.py:34-43) */
int addk_argmax_nchw(const float* logits, int32_t N, int32_t C, int64_t HW, int64_t* out, void* stream);
int addk_confusion(const int64_t* gt, const int64_t* pred, int64_t n, int32_t num_class, int64_t* cm, void* stream);

/* ---------------------------------------------------------------------------------------
 * GPU input pipeline: Cityscapes sample preparation from DECODED 8-bit planes (reference
 * dataloaders/datasets/cityscapes.py:64-91 encode_segmap; dataloaders/custom_transforms.py:238-286 train_preprocess,
 * :322-347 full_image_eval_preprocess).  Bit-exact with the reference's PIL calls: the resampling tables are built on the
 * host with PIL's own arithmetic (addk/data.py) and the kernels apply them in PIL's fixed point.
 * ------------------------------------------------------------------------------------- */
int addk_lut_u8(const uint8_t* in, uint8_t* out, int64_t n, const uint8_t* lut256, void* stream);
/* one pass of the antialiased 8-bit resize (interleaved C channels): out = clip8(((1<<21) + sum_k in*coef[k]) >> 22);
 * bounds[2*o] = first source index, bounds[2*o+1] = tap count of output index o; coef is [out extent][ksize] int32.
 * vertical = 0: along x (mirror = 1 reads the row right-to-left: the random flip); vertical = 1: along y. */
int addk_resample_u8(const uint8_t* in, int32_t IH, int32_t IW, uint8_t* out, int32_t OH, int32_t OW, int32_t C, const int32_t* bounds,
                     const int32_t* coef, int32_t ksize, int32_t vertical, int32_t mirror, void* stream);
/* NEAREST resize of a label plane through source-index tables (xtab[OW], ytab[OH]) */
int addk_nearest_u8(const uint8_t* in, int32_t IH, int32_t IW, uint8_t* out, int32_t OH, int32_t OW, const int32_t* xtab, const int32_t* ytab,
                    int32_t mirror, void* stream);
/* ToTensor + Normalize + pad (image 0, label 255) + crop at (i0, j0): HWC u8 -> CHW float [3][CH][CW], labels -> int64 [CH][CW] (lbl/out_lbl may be NULL) */
int addk_finish_sample(const uint8_t* img, const uint8_t* lbl, int32_t IH, int32_t IW, int32_t i0, int32_t j0, int32_t CH, int32_t CW,
                       const float* mean3, const float* std3, float* out_img, int64_t* out_lbl, void* stream);

/* ---- small-message all-reduce of the SyncBN statistics inside one node (csrc/comm.hip) -------------------------------------------------
 * Replaces, for the (sum, sumsq) / (dmean, dvar) vectors of the BatchNorms of one dependency level, what the reference moves through its
 * master / slave pipes per BatchNorm call (modeling/sync_batchnorm/batchnorm.py:95-108, comm.py:56-129) and what rounds 2-4 sent through a
 * stock RCCL all_reduce.  Every rank owns a mailbox in its own HBM (fine-grained allocation, exported as a hipIpc handle, mapped by every
 * peer); an exchange is ONE single-workgroup launch: push the vector into every mailbox, flag it, poll the own mailbox (bounded: a missing
 * flag sets an error word and the kernel returns), add the ranks' vectors in rank order (bit-identical on every rank).  Capturable in a
 * hipGraph (the sequence number lives in device memory).  Every rank issues the same sequence of addk_comm_allreduce calls.
 *   host flow: addk_comm_alloc -> exchange the 64-byte handles (any out-of-band channel, e.g. the process group) -> addk_comm_open
 *              -> addk_comm_allreduce ... -> addk_comm_status (error word) -> addk_comm_close.
 * Error behaviour: negative status + addk_last_error() for invalid arguments / HIP failures; a timed-out exchange is reported by
 * addk_comm_status (err != 0: bit 63 | sequence number << 8 | rank whose flag never arrived), never by a hang. */
int64_t addk_comm_mailbox_bytes(int32_t world, int64_t max_bytes);
int addk_comm_alloc(int32_t world, int64_t max_bytes, void** mailbox, void* handle64);
int addk_comm_open(int32_t rank, int32_t world, int64_t max_bytes, void* my_mailbox, const void* handles, void** comm_out);
int addk_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype /* 0 fp32, 1 fp64 */, void* stream);
int addk_comm_status(void* comm, int64_t* seq, int64_t* err);
int addk_comm_close(void* comm, void* my_mailbox);

/* diagnostic: print the NATIVE stack on SIGABRT / SIGSEGV (stderr), then die as before.  tests/conftest.py and bench.py's child processes call it. */
int addk_debug_trace_fatal_signals(void);

#ifdef __cplusplus
}
#endif
#endif
