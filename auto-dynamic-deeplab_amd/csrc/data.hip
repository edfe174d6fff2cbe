// GPU input pipeline of the Cityscapes training / evaluation samples (reference dataloaders/datasets/cityscapes.py:64-91 and
// dataloaders/custom_transforms.py:238-286, 322-347), from the DECODED 8-bit image and labelIds planes to the network's
// input tensors.  Integer / byte work, HBM-bound: every kernel is a coalesced gather with per-row / per-column tables
// built on the host (the tables carry the arithmetic of PIL's resampling, so results are bit-exact with the reference's
// PIL calls — tests/test_data_pipeline.py).
//   lut_u8            encode_segmap: labelId -> train id through a 256-entry table (void classes -> 255)
//   resample_u8       one pass (horizontal or vertical) of PIL's antialiased resize of 8-bit interleaved pixels: fixed-point
//                     coefficients (22 fractional bits), rounding and clipping to 8 bits after EACH pass as ImagingResample does;
//                     the horizontal pass can read the row mirrored (the random left-right flip)
//   nearest_u8        PIL's NEAREST resize of the label plane through host-built source-index tables (+ the same mirror)
//   finish_sample     ToTensor + Normalize + ZeroPad2d / ConstantPad2d(255) + crop: HWC u8 image -> CHW float, u8 labels -> int64
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) lut_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long n, const uint8_t* __restrict__ lut) {
  __shared__ uint8_t l[256];
  l[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uchar4 v = reinterpret_cast<const uchar4*>(in)[i];
    reinterpret_cast<uchar4*>(out)[i] = make_uchar4(l[v.x], l[v.y], l[v.z], l[v.w]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; out[i] = l[in[i]]; }
}

// out[y][x][c] = clip8((1 << 21) + sum_k in[...][c] * coef[k]) >> 22   (horizontal: along x with bounds[x] = (first, count);
// vertical: along y).  ksize = row length of the coefficient table.
__global__ void __launch_bounds__(256) resample_u8_kernel(const uint8_t* __restrict__ in, int IH, int IW, uint8_t* __restrict__ out, int OH, int OW, int C,
                                                          const int* __restrict__ bounds, const int* __restrict__ coef, int ksize, int vertical, int mirror) {
  const long total = (long)OH * OW * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C); const long p = i / C;
    const int x = (int)(p % OW), y = (int)(p / OW);
    const int o = vertical ? y : x;
    const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = coef + (long)o * ksize;
    int ss = 1 << 21;
    if (vertical) {
      for (int j = 0; j < cnt; ++j) ss += (int)in[((long)(first + j) * IW + x) * C + c] * k[j];
    } else {
      for (int j = 0; j < cnt; ++j) { const int sx = first + j; ss += (int)in[((long)y * IW + (mirror ? IW - 1 - sx : sx)) * C + c] * k[j]; }
    }
    ss >>= 22;
    out[i] = (uint8_t)(ss < 0 ? 0 : ss > 255 ? 255 : ss);
  }
}

__global__ void __launch_bounds__(256) nearest_u8_kernel(const uint8_t* __restrict__ in, int IH, int IW, uint8_t* __restrict__ out, int OH, int OW,
                                                         const int* __restrict__ xtab, const int* __restrict__ ytab, int mirror) {
  const long total = (long)OH * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % OW), y = (int)(i / OW);
    const int sx = xtab[x];
    out[i] = in[(long)ytab[y] * IW + (mirror ? IW - 1 - sx : sx)];
  }
}

// image [IH][IW][3] u8, label [IH][IW] u8 -> out_img [3][CH][CW] float = (v/255 - mean)/std inside the image, 0 in the padding;
// out_lbl [CH][CW] int64 = label inside, 255 in the padding; (i0, j0) = crop origin in the padded sample.
__global__ void __launch_bounds__(256) finish_sample_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lbl, int IH, int IW, int i0, int j0,
                                                            int CH, int CW, float m0, float m1, float m2, float s0, float s1, float s2,
                                                            float* __restrict__ out_img, long long* __restrict__ out_lbl) {
  const long total = (long)CH * CW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % CW), y = (int)(i / CW);
    const int sy = y + i0, sx = x + j0;
    const bool in = sy < IH && sx < IW;
    float r = 0.f, g = 0.f, b = 0.f; long long l = 255;
    if (in) {
      const uint8_t* px = img + ((long)sy * IW + sx) * 3;
      // torchvision: ToTensor divides by 255 (float32), Normalize subtracts the mean and divides by the std
      r = ((float)px[0] / 255.f - m0) / s0; g = ((float)px[1] / 255.f - m1) / s1; b = ((float)px[2] / 255.f - m2) / s2;
      if (lbl) l = lbl[(long)sy * IW + sx];
    }
    out_img[i] = r; out_img[total + i] = g; out_img[2 * total + i] = b;
    if (out_lbl) out_lbl[i] = l;
  }
}

inline int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b); }

}  // namespace

extern "C" int addk_lut_u8(const uint8_t* in, uint8_t* out, int64_t n, const uint8_t* lut256, void* stream) {
  ADDK_REQUIRE(in && out && lut256 && n > 0, "lut_u8: bad args");
  ADDK_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 3) == 0, "lut_u8: planes must be 4-byte aligned");
  hipLaunchKernelGGL(lut_u8_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, in, out, (long)n, lut256);
  return addk_check_launch("lut_u8");
}
extern "C" int addk_resample_u8(const uint8_t* in, int32_t IH, int32_t IW, uint8_t* out, int32_t OH, int32_t OW, int32_t C, const int32_t* bounds,
                                const int32_t* coef, int32_t ksize, int32_t vertical, int32_t mirror, void* stream) {
  ADDK_REQUIRE(in && out && bounds && coef && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && ksize > 0, "resample_u8: bad args");
  ADDK_REQUIRE(vertical ? OW == IW : OH == IH, "resample_u8: a pass changes one axis only");
  hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for((long)OH * OW * C)), dim3(256), 0, (hipStream_t)stream, in, IH, IW, out, OH, OW, C, bounds, coef, ksize,
                     vertical, mirror);
  return addk_check_launch("resample_u8");
}
extern "C" int addk_nearest_u8(const uint8_t* in, int32_t IH, int32_t IW, uint8_t* out, int32_t OH, int32_t OW, const int32_t* xtab, const int32_t* ytab,
                               int32_t mirror, void* stream) {
  ADDK_REQUIRE(in && out && xtab && ytab && IH > 0 && IW > 0 && OH > 0 && OW > 0, "nearest_u8: bad args");
  hipLaunchKernelGGL(nearest_u8_kernel, dim3(grid_for((long)OH * OW)), dim3(256), 0, (hipStream_t)stream, in, IH, IW, out, OH, OW, xtab, ytab, mirror);
  return addk_check_launch("nearest_u8");
}
extern "C" int addk_finish_sample(const uint8_t* img, const uint8_t* lbl, int32_t IH, int32_t IW, int32_t i0, int32_t j0, int32_t CH, int32_t CW,
                                  const float* mean3, const float* std3, float* out_img, int64_t* out_lbl, void* stream) {
  ADDK_REQUIRE(img && out_img && mean3 && std3 && IH > 0 && IW > 0 && CH > 0 && CW > 0 && i0 >= 0 && j0 >= 0, "finish_sample: bad args");
  hipLaunchKernelGGL(finish_sample_kernel, dim3(grid_for((long)CH * CW)), dim3(256), 0, (hipStream_t)stream, img, lbl, IH, IW, i0, j0, CH, CW,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_img, (long long*)out_lbl);
  return addk_check_launch("finish_sample");
}
