// Elementwise / channel-reducing kernels of the path (all HBM-bound, 16 B per lane):
//   affine_sum      — the branch sum of a cell block (ADD.py:108) with the branches' BatchNorms applied on
//                     the fly, written into the block's slot of the cell concat buffer (ADD.py:112)
//   affine_sum_bwd  — its gradient + the per-channel sums that drive the branches' BN backward
//   bn_bwd_apply    — dy = g + c1 + c2*(x - mean) (training-mode BatchNorm backward on the raw conv output)
//   sgd_step, fill
#include "common.h"

namespace {

struct SumK {
  addk_src term[ADDK_MAX_TERMS]; int nterm;
  long P; int C;
  float* out; int ldo; int relu_out; int accumulate;
  const float* dout; int lddo;
  const float* fout; int ldfo;
  float* g[ADDK_MAX_TERMS]; int ldg[ADDK_MAX_TERMS]; int acc[ADDK_MAX_TERMS];
  double* dab[ADDK_MAX_TERMS];
  int nq, npl, vec;
};

__global__ void __launch_bounds__(256) affine_sum_fwd_kernel(const SumK p) {
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  if (pl >= p.npl) return;
  const int c = 4 * q, nrem = p.C - c;
  float4 av[ADDK_MAX_TERMS], bv[ADDK_MAX_TERMS];
#pragma unroll
  for (int i = 0; i < ADDK_MAX_TERMS; ++i) {
    av[i] = make_float4(1.f, 1.f, 1.f, 1.f); bv[i] = zero4();
    if (i < p.nterm && p.term[i].a) { av[i] = ld4g(p.term[i].a + c, nrem, p.vec); bv[i] = ld4g(p.term[i].b + c, nrem, p.vec); }
  }
  for (long pp = (long)blockIdx.x * p.npl + pl; pp < p.P; pp += (long)gridDim.x * p.npl) {
    float4 s = zero4();
#pragma unroll
    for (int i = 0; i < ADDK_MAX_TERMS; ++i) {
      if (i < p.nterm) {
        float4 x = ld4g(p.term[i].x + pp * p.term[i].ld + c, nrem, p.vec);
        float4 z = make_float4(fmaf(av[i].x, x.x, bv[i].x), fmaf(av[i].y, x.y, bv[i].y), fmaf(av[i].z, x.z, bv[i].z), fmaf(av[i].w, x.w, bv[i].w));
        if (p.term[i].relu) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
        s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w;
      }
    }
    if (p.relu_out) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    float* op = p.out + pp * p.ldo + c;
    if (p.accumulate) { float4 o = ld4g(op, nrem, p.vec); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    st4g(op, s, nrem, p.vec);
  }
}

__global__ void __launch_bounds__(256) affine_sum_bwd_kernel(const SumK p) {
  extern __shared__ double redt[];       // [C4][2]
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q, nrem = p.C - c;
  float4 av[ADDK_MAX_TERMS], bv[ADDK_MAX_TERMS];
  double sA[ADDK_MAX_TERMS][4], sB[ADDK_MAX_TERMS][4];
#pragma unroll
  for (int i = 0; i < ADDK_MAX_TERMS; ++i) {
    av[i] = make_float4(1.f, 1.f, 1.f, 1.f); bv[i] = zero4();
#pragma unroll
    for (int e = 0; e < 4; ++e) { sA[i][e] = 0.0; sB[i][e] = 0.0; }
    if (active && i < p.nterm && p.term[i].a) { av[i] = ld4g(p.term[i].a + c, nrem, p.vec); bv[i] = ld4g(p.term[i].b + c, nrem, p.vec); }
  }
  if (active) {
    for (long pp = (long)blockIdx.x * p.npl + pl; pp < p.P; pp += (long)gridDim.x * p.npl) {
      float4 d = ld4g(p.dout + pp * p.lddo + c, nrem, p.vec);
      if (p.relu_out) {
        float4 o = ld4g(p.fout + pp * p.ldfo + c, nrem, p.vec);
        if (!(o.x > 0.f)) d.x = 0.f; if (!(o.y > 0.f)) d.y = 0.f; if (!(o.z > 0.f)) d.z = 0.f; if (!(o.w > 0.f)) d.w = 0.f;
      }
#pragma unroll
      for (int i = 0; i < ADDK_MAX_TERMS; ++i) {
        if (i < p.nterm && (p.g[i] || p.dab[i])) {
          float4 x = ld4g(p.term[i].x + pp * p.term[i].ld + c, nrem, p.vec);
          float4 dm = d;
          if (p.term[i].relu) {
            if (!(fmaf(av[i].x, x.x, bv[i].x) > 0.f)) dm.x = 0.f;
            if (!(fmaf(av[i].y, x.y, bv[i].y) > 0.f)) dm.y = 0.f;
            if (!(fmaf(av[i].z, x.z, bv[i].z) > 0.f)) dm.z = 0.f;
            if (!(fmaf(av[i].w, x.w, bv[i].w) > 0.f)) dm.w = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { sA[i][e] += (double)get4(dm, e) * (double)get4(x, e); sB[i][e] += (double)get4(dm, e); }
          if (p.g[i]) {
            float4 gv = make_float4(dm.x * av[i].x, dm.y * av[i].y, dm.z * av[i].z, dm.w * av[i].w);
            float* gp = p.g[i] + pp * p.ldg[i] + c;
            if (p.acc[i]) { float4 o = ld4g(gp, nrem, p.vec); gv.x += o.x; gv.y += o.y; gv.z += o.z; gv.w += o.w; }
            st4g(gp, gv, nrem, p.vec);
          }
        }
      }
    }
  }
  // Block reduction over the pixel lanes through an [npl][C4][2] fp64 panel: one barrier, then one thread per (channel, A|B)
  // adds the npl rows in fixed order (the former lane-after-lane accumulation cost npl barrier rounds per term).
  const int C4 = p.nq * 4;
#pragma unroll
  for (int i = 0; i < ADDK_MAX_TERMS; ++i) {
    if (i < p.nterm && p.dab[i]) {          // block-uniform
      if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { redt[((pl * C4) + c + e) * 2] = sA[i][e]; redt[((pl * C4) + c + e) * 2 + 1] = sB[i][e]; }
      }
      __syncthreads();
      for (int k = threadIdx.x; k < p.C * 2; k += 256) {
        const int ch = k >> 1, ab = k & 1;
        double acc = 0.0;
        for (int r = 0; r < p.npl; ++r) acc += redt[((r * C4) + ch) * 2 + ab];
        p.dab[i][(long)blockIdx.x * p.C * 2 + k] = acc;
      }
      __syncthreads();
    }
  }
}

// Vector-aligned form of affine_sum_bwd_kernel with the term count a template parameter: plain 16-byte loads, TWO pixels per trip with every load of
// both requested before anything is used (the generic kernel's 2-3 trips per thread were 2-3 dependent round trips on launches that move 15-30 MB).
// Same per-thread pixel order and arithmetic as the generic kernel: bit-identical results.
template <int NT>
__global__ void __launch_bounds__(256) affine_sum_bwd_vec_kernel(const SumK p) {
  extern __shared__ double redt[];       // [npl][C4][2]
  const int q = threadIdx.x % p.nq, pl = threadIdx.x / p.nq;
  const bool active = pl < p.npl;
  const int c = 4 * q;
  float4 av[NT], bv[NT];
  double sA[NT][4], sB[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    av[i] = make_float4(1.f, 1.f, 1.f, 1.f); bv[i] = zero4();
#pragma unroll
    for (int e = 0; e < 4; ++e) { sA[i][e] = 0.0; sB[i][e] = 0.0; }
    if (active && p.term[i].a) { av[i] = ld4(p.term[i].a + c); bv[i] = ld4(p.term[i].b + c); }
  }
  if (active) {
    const long stride = (long)gridDim.x * p.npl;
    auto finish = [&](long pp, float4 d, const float4 o, const float4 (&x)[NT], const float4 (&gold)[NT]) {
      if (p.relu_out) { if (!(o.x > 0.f)) d.x = 0.f; if (!(o.y > 0.f)) d.y = 0.f; if (!(o.z > 0.f)) d.z = 0.f; if (!(o.w > 0.f)) d.w = 0.f; }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (p.g[i] || p.dab[i]) {
          float4 dm = d;
          if (p.term[i].relu) {
            if (!(fmaf(av[i].x, x[i].x, bv[i].x) > 0.f)) dm.x = 0.f;
            if (!(fmaf(av[i].y, x[i].y, bv[i].y) > 0.f)) dm.y = 0.f;
            if (!(fmaf(av[i].z, x[i].z, bv[i].z) > 0.f)) dm.z = 0.f;
            if (!(fmaf(av[i].w, x[i].w, bv[i].w) > 0.f)) dm.w = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { sA[i][e] += (double)get4(dm, e) * (double)get4(x[i], e); sB[i][e] += (double)get4(dm, e); }
          if (p.g[i]) {
            float4 gv = make_float4(dm.x * av[i].x, dm.y * av[i].y, dm.z * av[i].z, dm.w * av[i].w);
            if (p.acc[i]) { gv.x += gold[i].x; gv.y += gold[i].y; gv.z += gold[i].z; gv.w += gold[i].w; }
            st4(p.g[i] + pp * p.ldg[i] + c, gv);
          }
        }
      }
    };
    for (long p0 = (long)blockIdx.x * p.npl + pl; p0 < p.P; p0 += 2 * stride) {
      const long p1 = p0 + stride;
      const bool ok1 = p1 < p.P;
      const long q1 = ok1 ? p1 : p0;                       // (the second pixel's loads fall on the first when there is none)
      const float4 d0 = ld4(p.dout + p0 * p.lddo + c), d1 = ld4(p.dout + q1 * p.lddo + c);
      float4 o0 = zero4(), o1 = zero4();
      if (p.relu_out) { o0 = ld4(p.fout + p0 * p.ldfo + c); o1 = ld4(p.fout + q1 * p.ldfo + c); }
      float4 x0[NT], x1[NT], g0[NT], g1[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        x0[i] = zero4(); x1[i] = zero4(); g0[i] = zero4(); g1[i] = zero4();
        if (p.g[i] || p.dab[i]) { x0[i] = ld4(p.term[i].x + p0 * p.term[i].ld + c); x1[i] = ld4(p.term[i].x + q1 * p.term[i].ld + c); }
        if (p.g[i] && p.acc[i]) { g0[i] = ld4(p.g[i] + p0 * p.ldg[i] + c); g1[i] = ld4(p.g[i] + q1 * p.ldg[i] + c); }
      }
      finish(p0, d0, o0, x0, g0);
      if (ok1) finish(p1, d1, o1, x1, g1);
    }
  }
  const int C4 = p.nq * 4;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (p.dab[i]) {          // block-uniform
      if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { redt[((pl * C4) + c + e) * 2] = sA[i][e]; redt[((pl * C4) + c + e) * 2 + 1] = sB[i][e]; }
      }
      __syncthreads();
      for (int k = threadIdx.x; k < p.C * 2; k += 256) {
        const int ch = k >> 1, ab = k & 1;
        double acc = 0.0;
        for (int r = 0; r < p.npl; ++r) acc += redt[((r * C4) + ch) * 2 + ab];
        p.dab[i][(long)blockIdx.x * p.C * 2 + k] = acc;
      }
      __syncthreads();
    }
  }
}

// vector-aligned form: every load is a plain 16-byte load issued up front (coefficients and data are one round trip)
__global__ void __launch_bounds__(256) bn_bwd_apply_vec_kernel(const float* g, int ldg, const float* x, int ldx, const float* mean,
                                                               const float* c1, const float* c2, long P, float* out, int ldo,
                                                               int nq, int npl) {
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  if (pl >= npl) return;
  const int c = 4 * q;
  const long pp0 = (long)blockIdx.x * npl + pl;
  if (pp0 >= P) return;
  float4 gv = ld4(g + pp0 * ldg + c);
  float4 xv = c2 ? ld4(x + pp0 * ldx + c) : zero4();
  const float4 mu = mean ? ld4(mean + c) : zero4();
  const float4 k1 = c1 ? ld4(c1 + c) : zero4();
  const float4 k2 = c2 ? ld4(c2 + c) : zero4();
  for (long pp = pp0;;) {
    const long nx = pp + (long)gridDim.x * npl;
    float4 gn = zero4(), xn = zero4();
    if (nx < P) { gn = ld4(g + nx * ldg + c); if (c2) xn = ld4(x + nx * ldx + c); }
    st4(out + pp * ldo + c, make_float4(gv.x + fmaf(k2.x, xv.x - mu.x, k1.x), gv.y + fmaf(k2.y, xv.y - mu.y, k1.y),
                                        gv.z + fmaf(k2.z, xv.z - mu.z, k1.z), gv.w + fmaf(k2.w, xv.w - mu.w, k1.w)));
    if (nx >= P) break;
    pp = nx; gv = gn; xv = xn;
  }
}

// several independent tensors in one launch (vector-aligned items only): block (x, y) strides over the pixels of item y
struct BnApplyItem { const float* g; const float* x; const float* c1; const float* c2; const float* mean; float* out; long P; int ldg, ldx, ldo, C; };
__global__ void __launch_bounds__(256) bn_bwd_apply_batch_kernel(const BnApplyItem* __restrict__ tab) {
  const BnApplyItem it = tab[blockIdx.y];
  const int nq = it.C >> 2, npl = 256 / nq;
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  if (pl >= npl) return;
  const int c = 4 * q;
  long pp = (long)blockIdx.x * npl + pl;
  if (pp >= it.P) return;
  const float4 k1 = ld4(it.c1 + c), k2 = ld4(it.c2 + c);
  const float4 mu = it.mean ? ld4(it.mean + c) : zero4();
  const long step = (long)gridDim.x * npl;
  float4 gv = ld4(it.g + pp * it.ldg + c), xv = ld4(it.x + pp * it.ldx + c);
  for (;;) {
    const long nx = pp + step;
    float4 gn = zero4(), xn = zero4();
    if (nx < it.P) { gn = ld4(it.g + nx * it.ldg + c); xn = ld4(it.x + nx * it.ldx + c); }
    st4(it.out + pp * it.ldo + c, make_float4(gv.x + fmaf(k2.x, xv.x - mu.x, k1.x), gv.y + fmaf(k2.y, xv.y - mu.y, k1.y),
                                              gv.z + fmaf(k2.z, xv.z - mu.z, k1.z), gv.w + fmaf(k2.w, xv.w - mu.w, k1.w)));
    if (nx >= it.P) break;
    pp = nx; gv = gn; xv = xn;
  }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* g, int ldg, const float* x, int ldx, const float* mean,
                                                           const float* c1, const float* c2, long P, int C, float* out, int ldo,
                                                           int nq, int npl, int vec) {
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  if (pl >= npl) return;
  const int c = 4 * q, nrem = C - c;
  float4 mu = mean ? ld4g(mean + c, nrem, vec) : zero4();
  float4 k1 = c1 ? ld4g(c1 + c, nrem, vec) : zero4();
  float4 k2 = c2 ? ld4g(c2 + c, nrem, vec) : zero4();
  for (long pp = (long)blockIdx.x * npl + pl; pp < P; pp += (long)gridDim.x * npl) {
    float4 gv = ld4g(g + pp * ldg + c, nrem, vec);
    float4 xv = c2 ? ld4g(x + pp * ldx + c, nrem, vec) : zero4();
    float4 o = make_float4(gv.x + fmaf(k2.x, xv.x - mu.x, k1.x), gv.y + fmaf(k2.y, xv.y - mu.y, k1.y),
                           gv.z + fmaf(k2.z, xv.z - mu.z, k1.z), gv.w + fmaf(k2.w, xv.w - mu.w, k1.w));
    st4g(out + pp * ldo + c, o, nrem, vec);
  }
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void sgd_kernel(float* p, const float* g, float* buf, long n, const float* lr_dev, float mom, float wd,
                           int nesterov, int first, float gscale) {
  const float lr = *lr_dev;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = fmaf(wd, w, g[i] * gscale);
    float b = first ? d : fmaf(mom, buf[i], d);
    buf[i] = b;
    float step = nesterov ? fmaf(mom, b, d) : b;
    p[i] = w - lr * step;
  }
}

int ew_blocks(long P, int npl) {
  long b = cdiv(P, (long)npl);
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}
int ew_rows(long P, int C) {
  EwMap m = ew_map(C);
  long r = P / ((long)m.npl * 2);
  if (r < 1) r = 1;
  if (r > 1024) r = 1024;
  return (int)r;
}

}  // namespace

extern "C" int addk_ew_rows(int64_t P, int32_t C) { return ew_rows(P, C); }

extern "C" int addk_affine_sum_fwd(const addk_affine_sum_args* a, void* stream) {
  ADDK_REQUIRE(a && a->nterm >= 1 && a->nterm <= ADDK_MAX_TERMS && a->P > 0 && a->C > 0 && a->C <= 1024, "affine_sum: bad args");
  ADDK_REQUIRE(a->out && a->ldo >= a->C, "affine_sum: bad output");
  SumK k{};
  k.vec = aligned16(a->out) && a->ldo % 4 == 0;
  for (int i = 0; i < a->nterm; ++i) {
    ADDK_REQUIRE(a->term[i].x && a->term[i].ld >= a->C && a->term[i].C == a->C, "affine_sum: bad term %d", i);
    ADDK_REQUIRE((a->term[i].a == nullptr) == (a->term[i].b == nullptr), "affine_sum: a/b must come together");
    k.term[i] = a->term[i];
    if (!src_vec_ok(a->term[i])) k.vec = 0;
  }
  k.nterm = a->nterm; k.P = a->P; k.C = a->C; k.out = a->out; k.ldo = a->ldo; k.relu_out = a->relu_out; k.accumulate = a->accumulate;
  EwMap m = ew_map(a->C); k.nq = m.nq; k.npl = m.npl;
  hipLaunchKernelGGL(affine_sum_fwd_kernel, dim3(ew_blocks(a->P, m.npl)), dim3(256), 0, (hipStream_t)stream, k);
  return addk_check_launch("affine_sum_fwd");
}

extern "C" int addk_affine_sum_bwd(const addk_affine_sum_bwd_args* a, void* stream) {
  ADDK_REQUIRE(a && a->nterm >= 1 && a->nterm <= ADDK_MAX_TERMS && a->P > 0 && a->C > 0 && a->C <= 1024, "affine_sum_bwd: bad args");
  ADDK_REQUIRE(a->dout && a->lddo >= a->C && (!a->relu_out || (a->out && a->ldo >= a->C)), "affine_sum_bwd: bad dout/out");
  SumK k{};
  k.vec = aligned16(a->dout) && a->lddo % 4 == 0 && (!a->relu_out || (aligned16(a->out) && a->ldo % 4 == 0));
  for (int i = 0; i < a->nterm; ++i) {
    ADDK_REQUIRE(a->term[i].x && a->term[i].ld >= a->C && a->term[i].C == a->C, "affine_sum_bwd: bad term %d", i);
    k.term[i] = a->term[i]; k.g[i] = a->g[i]; k.ldg[i] = a->ldg[i]; k.acc[i] = a->accumulate[i]; k.dab[i] = (double*)a->dab[i];
    ADDK_REQUIRE(!a->g[i] || a->ldg[i] >= a->C, "affine_sum_bwd: short ldg %d", i);
    if (!src_vec_ok(a->term[i]) || (a->g[i] && (!aligned16(a->g[i]) || a->ldg[i] % 4))) k.vec = 0;
  }
  k.nterm = a->nterm; k.P = a->P; k.C = a->C; k.dout = a->dout; k.lddo = a->lddo; k.fout = a->out; k.ldfo = a->ldo; k.relu_out = a->relu_out;
  EwMap m = ew_map(a->C); k.nq = m.nq; k.npl = m.npl;
  size_t sh = (size_t)m.npl * m.nq * 4 * 2 * sizeof(double);
  const int vec2 = 1;
  bool vec_ok = vec2 && k.vec && a->C % 4 == 0 && a->C == m.nq * 4;
  for (int i = 0; i < a->nterm && vec_ok; ++i) if (a->term[i].a && (!aligned16(a->term[i].a) || !aligned16(a->term[i].b))) vec_ok = false;
  const dim3 grid(ew_rows(a->P, a->C));
  if (vec_ok) {
    switch (a->nterm) {
      case 1: hipLaunchKernelGGL(affine_sum_bwd_vec_kernel<1>, grid, dim3(256), sh, (hipStream_t)stream, k); break;
      case 2: hipLaunchKernelGGL(affine_sum_bwd_vec_kernel<2>, grid, dim3(256), sh, (hipStream_t)stream, k); break;
      case 3: hipLaunchKernelGGL(affine_sum_bwd_vec_kernel<3>, grid, dim3(256), sh, (hipStream_t)stream, k); break;
      default: hipLaunchKernelGGL(affine_sum_bwd_vec_kernel<4>, grid, dim3(256), sh, (hipStream_t)stream, k); break;
    }
    return addk_check_launch("affine_sum_bwd");
  }
  hipLaunchKernelGGL(affine_sum_bwd_kernel, grid, dim3(256), sh, (hipStream_t)stream, k);
  return addk_check_launch("affine_sum_bwd");
}

extern "C" int addk_bn_bwd_apply(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* mean, const float* c1,
                                 const float* c2, int64_t P, int32_t C, float* out, int32_t ldo, void* stream) {
  ADDK_REQUIRE(g && out && P > 0 && C > 0 && C <= 1024 && ldg >= C && ldo >= C && (!c2 || (x && ldx >= C)), "bn_bwd_apply: bad args");
  ADDK_REQUIRE((c1 == nullptr) == (c2 == nullptr), "bn_bwd_apply: c1/c2 come together");
  int vec = aligned16(g) && aligned16(out) && ldg % 4 == 0 && ldo % 4 == 0 && C % 4 == 0 && (!x || (aligned16(x) && ldx % 4 == 0)) &&
            (!mean || aligned16(mean)) && (!c1 || (aligned16(c1) && aligned16(c2)));
  EwMap m = ew_map(C);
  if (vec && C == m.nq * 4) {
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(ew_blocks(P, m.npl)), dim3(256), 0, (hipStream_t)stream, g, ldg, x, ldx, mean, c1, c2,
                       (long)P, out, ldo, m.nq, m.npl);
    return addk_check_launch("bn_bwd_apply");
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(P, m.npl)), dim3(256), 0, (hipStream_t)stream, g, ldg, x, ldx, mean, c1, c2,
                     (long)P, C, out, ldo, m.nq, m.npl, vec);
  return addk_check_launch("bn_bwd_apply");
}

extern "C" int addk_fill(float* p, int64_t n, float v, void* stream) {
  ADDK_REQUIRE(p && n >= 0, "fill: bad args");
  if (n == 0) return 0;
  int b = cdiv(n, 1024); if (b > 2048) b = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, p, (long)n, v);
  return addk_check_launch("fill");
}

extern "C" int addk_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* lr_dev, float momentum, float weight_decay,
                             int32_t nesterov, int32_t first, float gscale, void* stream) {
  ADDK_REQUIRE(p && g && buf && lr_dev && n > 0, "sgd_step: bad args");
  int b = cdiv(n, 1024); if (b > 4096) b = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long)n, lr_dev, momentum, weight_decay, nesterov,
                     first, gscale);
  return addk_check_launch("sgd_step");
}

// table entries as addk_bn_apply_item (include/addk.h); every item must be vector-aligned (16-byte pointers, ld % 4 == 0,
// C % 4 == 0, C <= 1024) with c1, c2 given (mean optional)
extern "C" int addk_bn_bwd_apply_batch(const addk_bn_apply_item* dev_table, int32_t n, int64_t max_P, void* stream) {
  ADDK_REQUIRE(dev_table && n > 0 && max_P > 0, "bn_bwd_apply_batch: bad args");
  static_assert(sizeof(addk_bn_apply_item) == sizeof(BnApplyItem), "item layout");
  long b = cdiv(max_P, 6); if (b > 2048) b = 2048; if (b < 1) b = 1;
  hipLaunchKernelGGL(bn_bwd_apply_batch_kernel, dim3((unsigned)b, n), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const BnApplyItem*>(dev_table));
  return addk_check_launch("bn_bwd_apply_batch");
}
