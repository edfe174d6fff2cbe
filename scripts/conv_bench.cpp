// Stand-alone timing of addk_conv_fwd / addk_conv_dgrad / wgrad on the heavy shapes of config 2 (through the C ABI).
// hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "addk.h"

static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  unsigned s = 12345u + (unsigned)n;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((s >> 8) * (1.0f / 8388608.0f) - 1.0f); }
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
struct Shape { const char* name; int N, H, W, Cin, Cout, k, dil, s; };      // s: stride (0 = 1)

static void dw_bench(hipStream_t st, hipEvent_t e0, hipEvent_t e1, int reps) {
  struct D { const char* name; int N, H, W, C, k, s; } ds[] = {
    {"dw 5x5 C40 @125x253", 2, 125, 253, 40, 5, 1}, {"dw 3x3 C40 @125x253", 2, 125, 253, 40, 3, 1},
    {"dw 5x5 C80 @63x127", 2, 63, 127, 80, 5, 1}, {"dw 3x3 C80 @63x127", 2, 63, 127, 80, 3, 1},
    {"dw 5x5 C160 @32x64", 2, 32, 64, 160, 5, 1}, {"dw 5x5 C40 s2 @125x253", 2, 125, 253, 40, 5, 2}};
  for (auto& d : ds) {
    long P = (long)d.N * d.H * d.W; int pad = d.k / 2;
    int OH = (d.H + 2 * pad - d.k) / d.s + 1, OW = (d.W + 2 * pad - d.k) / d.s + 1; long PO = (long)d.N * OH * OW;
    float* x = dev_rand(P * d.C, 1.f); float* a = dev_rand(d.C, 1.f); float* b = dev_rand(d.C, .5f); float* w = dev_rand(d.C * d.k * d.k, .3f);
    float* y; hipMalloc(&y, PO * d.C * 4); float* g; hipMalloc(&g, P * d.C * 4); float* dw; hipMalloc(&dw, d.C * d.k * d.k * 4);
    int rows = addk_dw_rows(P, d.C); double* dab; hipMalloc(&dab, (size_t)rows * d.C * 16); float* ws; hipMalloc(&ws, (size_t)rows * d.C * d.k * d.k * 4);
    addk_dw_args ar; memset(&ar, 0, sizeof ar);
    ar.src.x = x; ar.src.a = a; ar.src.b = b; ar.src.ld = d.C; ar.src.C = d.C; ar.src.relu = 1;
    ar.N = d.N; ar.H = d.H; ar.W = d.W; ar.OH = OH; ar.OW = OW; ar.KH = ar.KW = d.k; ar.stride = d.s; ar.pad = pad; ar.dil = 1; ar.w = w; ar.y = y; ar.ldy = d.C;
    addk_dw_bwd_args ba; memset(&ba, 0, sizeof ba);
    ba.dy = y; ba.lddy = d.C; ba.N = d.N; ba.H = d.H; ba.W = d.W; ba.OH = OH; ba.OW = OW; ba.KH = ba.KW = d.k; ba.stride = d.s; ba.pad = pad; ba.dil = 1;
    ba.src = ar.src; ba.w = w; ba.g = g; ba.ldg = d.C; ba.dab = dab; ba.dw = dw; ba.ws = ws;
    for (int mode = 0; mode < 2; ++mode) {
      auto run = [&] { return mode == 0 ? addk_dw_fwd(&ar, st) : addk_dw_bwd(&ba, st); };
      if (run() != 0) { printf("%s: error %s\n", d.name, addk_last_error()); return; }
      hipStreamSynchronize(st);
      hipEventRecord(e0, st); for (int r = 0; r < reps; ++r) run(); hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      double mb = 4.0 * d.C * (mode == 0 ? P + PO : 2 * P + PO) * 1e-6;
      printf("%-34s %-5s %8.1f us  %6.0f GB/s of min traffic (%.1f MB)\n", d.name, mode ? "bwd" : "fwd", ms * 1e3, mb / ms, mb);
    }
    hipFree(x); hipFree(a); hipFree(b); hipFree(w); hipFree(y); hipFree(g); hipFree(dw); hipFree(dab); hipFree(ws);
  }
}

int main(int argc, char** argv) {
  Shape shapes[] = {
    {"decoder 3x3 304->256 @128x256", 2, 128, 256, 304, 256, 3, 1},
    {"decoder 3x3 256->256 @128x256", 2, 128, 256, 256, 256, 3, 1},
    {"aspp 3x3 d6 256->256 @64x128", 2, 64, 128, 256, 256, 3, 6},
    {"stem 3x3 64->64 @512x1024", 2, 512, 1024, 64, 64, 3, 1},
    {"aspp 1x1 1280->256 @64x128", 2, 64, 128, 1280, 256, 1, 1},
    {"cell 1x1 40->40 @125x253", 2, 125, 253, 40, 40, 1, 1},
    {"cell 1x1 80->80 @63x127", 2, 63, 127, 80, 80, 1, 1},
    {"cell 1x1 160->160 @32x64", 2, 32, 64, 160, 160, 1, 1},
    {"cell 1x1 200->40 @125x253", 2, 125, 253, 200, 40, 1, 1},
    {"cell 5x5 40->40 @125x253", 2, 125, 253, 40, 40, 5, 1},
    {"dil 5x5 d2 80->80 @63x127", 2, 63, 127, 80, 80, 5, 2},
    {"dil 3x3 d2 80->80 @63x127", 2, 63, 127, 80, 80, 3, 2},
    {"dil 5x5 d2 80->80 @64x128", 2, 64, 128, 80, 80, 5, 2},
    {"dil 5x5 d2 160->160 @32x64", 2, 32, 64, 160, 160, 5, 2},
    {"dil 3x3 d2 160->160 @32x64", 2, 32, 64, 160, 160, 3, 2},
    {"dil 5x5 d2 40->40 @125x253", 2, 125, 253, 40, 40, 5, 2},
    {"dil 3x3 d2 40->40 @125x253", 2, 125, 253, 40, 40, 3, 2},
    {"glue 1x1 400->80 @63x127", 2, 63, 127, 400, 80, 1, 1},
    {"glue 1x1 800->160 @32x64", 2, 32, 64, 800, 160, 1, 1},
    {"stem2 3x3s2 64->128 @512x1024", 2, 512, 1024, 64, 128, 3, 1, 2},
  };
  const char* only = getenv("SHAPES");
  int reps = argc > 1 ? atoi(argv[1]) : 20;
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (only && strstr(only, "dw")) { dw_bench(st, e0, e1, reps); return 0; }
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const int str = s.s ? s.s : 1;
    int taps = s.k * s.k, pad = s.dil * (s.k - 1) / 2;
    const int OH = (s.H + 2 * pad - s.dil * (s.k - 1) - 1) / str + 1, OW = (s.W + 2 * pad - s.dil * (s.k - 1) - 1) / str + 1;
    long P = (long)s.N * s.H * s.W; const long PI = P; (void)PI;
    const long PO = (long)s.N * OH * OW;
    float* x = dev_rand(P * s.Cin, 1.f); float* a = dev_rand(s.Cin, 1.f); float* b = dev_rand(s.Cin, 0.5f);
    float* w = dev_rand((size_t)s.Cout * taps * s.Cin, 0.05f);
    float* y; hipMalloc(&y, PO * s.Cout * 4);
    float* g; hipMalloc(&g, P * s.Cin * 4);
    int rows = addk_conv_rows(PO, s.Cout);
    double* slab; hipMalloc(&slab, (size_t)rows * s.Cout * 2 * 8);
    double* dab; hipMalloc(&dab, (size_t)addk_conv_rows(P, s.Cin) * s.Cin * 2 * 8);
    addk_conv_args ar; memset(&ar, 0, sizeof ar);
    ar.src[0].x = x; ar.src[0].a = a; ar.src[0].b = b; ar.src[0].ld = s.Cin; ar.src[0].C = s.Cin; ar.src[0].relu = 1; ar.nsrc = 1;
    ar.N = s.N; ar.H = s.H; ar.W = s.W; ar.OH = OH; ar.OW = OW; ar.KH = ar.KW = s.k; ar.stride = str; ar.pad = pad; ar.dil = s.dil;
    ar.Cout = s.Cout; ar.ldw = taps * s.Cin; ar.cin_total = s.Cin; ar.ldy = s.Cout; ar.w = w; ar.y = y; ar.stats = slab; ar.stats_ld = s.Cout;
    addk_conv_dgrad_args dg; memset(&dg, 0, sizeof dg);
    dg.dy = y; dg.lddy = s.Cout; dg.Cout = s.Cout; dg.N = s.N; dg.H = s.H; dg.W = s.W; dg.OH = OH; dg.OW = OW; dg.KH = dg.KW = s.k;
    dg.stride = str; dg.pad = pad; dg.dil = s.dil; dg.w = w; dg.ldw = taps * s.Cin; dg.cin_total = s.Cin; dg.dst = ar.src[0];
    dg.g = g; dg.ldg = s.Cin; dg.dab = dab;
    addk_conv_wgrad_args wg; memset(&wg, 0, sizeof wg);
    wg.dy = y; wg.lddy = s.Cout; wg.Cout = s.Cout; wg.N = s.N; wg.H = s.H; wg.W = s.W; wg.OH = OH; wg.OW = OW; wg.KH = wg.KW = s.k;
    wg.stride = str; wg.pad = pad; wg.dil = s.dil; wg.src = ar.src[0]; wg.ldw = taps * s.Cin; wg.cin_total = s.Cin;
    hipMalloc(&wg.dw, (size_t)s.Cout * taps * s.Cin * 4);
    wg.ws_floats = addk_conv_wgrad_ws(PO, s.Cout, s.Cin, taps); hipMalloc(&wg.ws, wg.ws_floats * 4);
    ar.wpack_floats = addk_conv_fwd_pack_floats(&ar); if (ar.wpack_floats) hipMalloc(&ar.wpack, ar.wpack_floats * 4);
    dg.wpack_floats = addk_conv_dgrad_pack_floats(&dg); if (dg.wpack_floats) hipMalloc(&dg.wpack, dg.wpack_floats * 4);
    double gf = 2.0 * PO * s.Cout * taps * s.Cin * 1e-9;
    typedef int (*diag_fn)(unsigned long long*);
    static diag_fn diag = (diag_fn)dlsym(RTLD_DEFAULT, "addk_c3b_diag");
    for (int mode = 0; mode < 3; ++mode) {
      if (getenv("NOWGRAD") && mode == 2) break;
      auto run = [&] { return mode == 0 ? addk_conv_fwd(&ar, st) : mode == 1 ? addk_conv_dgrad(&dg, st) : addk_conv_wgrad(&wg, st); };
      if (run() != 0) { printf("%s: error %s\n", s.name, addk_last_error()); return 1; }
      hipStreamSynchronize(st);
      if (getenv("PACKED")) { ar.wpack_ready = 1; dg.wpack_ready = 1; }      // the weight pack ran in the call above: time the convolution launch alone, as a plan does (hoisted packs)
      if (diag && mode < 2) { unsigned long long z[12]; diag(z); }             // reset the diagnostic counters after the untimed call
      float ms;
      if (getenv("COLD")) {      // every timed run behind a 1 GiB fill: operands come from HBM, not from L2 / the infinity cache
        static char* junk = nullptr; if (!junk) hipMalloc(&junk, 1L << 30);
        ms = 0.f;
        for (int r = 0; r < reps; ++r) {
          hipMemsetAsync(junk, r, 1L << 30, st);
          hipEventRecord(e0, st); run(); hipEventRecord(e1, st); hipEventSynchronize(e1);
          float t; hipEventElapsedTime(&t, e0, e1); ms += t;
        }
        ms /= reps;
      } else {
        hipEventRecord(e0, st); for (int r = 0; r < reps; ++r) run(); hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      }
      double mb = 4.0 * (P * s.Cin + PO * s.Cout) * 1e-6;
      printf("%-34s %-5s %8.1f us  %6.1f TFLOP/s (%.1f GF)  %6.0f GB/s of min traffic\n", s.name, mode == 0 ? "fwd" : mode == 1 ? "dgrad" : "wgrad", ms * 1e3, gf / ms, gf, mb / ms);
      // diagnostic library (built with -DADDK_C3B_DIAG): the clock inside the split kernel's workgroups
      unsigned long long dv[12];
      if (diag && mode < 2 && diag(dv) == 0 && dv[1]) {
        const double life = (double)dv[0];
        printf("      in-kernel clock: %.0f MHz (s_memtime / s_memrealtime over %llu workgroups, mean workgroup life %.1f us); wave 0's life: first patch %.1f %%, matrix phase %.1f %%, "
               "barrier behind it %.1f %%, prologue + split + LDS stores %.1f %%, barrier behind them %.1f %%, epilogue %.1f %%\n",
               100.0 * life / (double)dv[1], dv[2], (double)dv[1] / (double)dv[2] / 100.0, 100.0 * dv[4] / life, 100.0 * dv[5] / life, 100.0 * dv[6] / life, 100.0 * dv[7] / life,
               100.0 * dv[8] / life, 100.0 * dv[9] / life);
      }
      // (-DADDK_WG_DIAG): clock and phase split inside wgrad_h3b_kernel
      static diag_fn wdiag = (diag_fn)dlsym(RTLD_DEFAULT, "addk_wg_diag");
      unsigned long long wv[8];
      if (wdiag && mode == 2 && wdiag(wv) == 0 && wv[1]) {
        const double life = (double)wv[0];
        printf("      in-kernel clock: %.0f MHz over %llu waves, mean wave life %.1f us; of it: address preparation %.1f %%, matrix phase %.1f %%, split %.1f %%, barrier %.1f %%, LDS stores + barrier %.1f %%\n",
               100.0 * life / (double)wv[1], wv[2], (double)wv[1] / (double)wv[2] / 100.0, 100.0 * wv[3] / life, 100.0 * wv[4] / life, 100.0 * wv[5] / life, 100.0 * wv[6] / life, 100.0 * wv[7] / life);
      }
    }
    hipFree(x); hipFree(a); hipFree(b); hipFree(w); hipFree(y); hipFree(g); hipFree(slab); hipFree(dab); hipFree(wg.dw); hipFree(wg.ws);
  }
  return 0;
}
