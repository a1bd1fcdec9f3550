// Input transform of the reference on the GPU (reference src/data_utils.py:24-30):
//   Resize(shorter side -> R, bilinear as Pillow resamples it) -> CenterCrop(R) -> RGB -> ToTensor -> Normalize(0.5, 0.5)
// fed by uint8 images, so only the decoded bytes cross PCIe (3 B per source pixel instead of 12 B per output pixel and
// the per-item CPU resize).  The arithmetic is Pillow's 8-bit resampler (Resample.c): separable, horizontal pass then
// vertical pass, coefficients in 22-bit fixed point, an 8-bit intermediate between the passes -- integer work, so the
// result equals the CPU path bit for bit.  The coefficient tables (which depend only on the sizes) come from the host
// (vaehip/preprocess.py); only the output columns / rows inside the centre crop are computed.
// Both passes are byte streams: one thread per output pixel (3 channels), x fastest.  HBM-bound: per output pixel the
// horizontal pass reads ~3*kx source bytes (neighbouring pixels share them through L2 / the coalescer) and writes 3,
// the vertical pass reads 3*ky and writes 12 (fp32 planes).
#include "common.h"

namespace {

constexpr int PBITS = 22;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// tmp[n][rows][R][3] (rows = source rows row0 .. row0+nrows-1) from src[n][H][W][C]; C = 1 (grey, replicated) or 3
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int H, int W, int R, int row0, int nrows,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ks,
                                                         uint8_t* __restrict__ tmp) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, n = blockIdx.z;
  if (x >= R) return;
  const int lo = bounds[2 * x], cnt = bounds[2 * x + 1];
  const int* k = kk + (int64_t)x * ks;
  const uint8_t* row = src + (((int64_t)n * H + row0 + y) * W + lo) * C;
  int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < cnt; ++t) {
    const int w = k[t];
    if (C == 3) {
      a0 += row[3 * t] * w;
      a1 += row[3 * t + 1] * w;
      a2 += row[3 * t + 2] * w;
    } else {
      a0 += row[t] * w;
    }
  }
  uint8_t* o = tmp + (((int64_t)n * nrows + y) * R + x) * 3;
  if (C == 3) {
    o[0] = (uint8_t)clip8(a0 >> PBITS);
    o[1] = (uint8_t)clip8(a1 >> PBITS);
    o[2] = (uint8_t)clip8(a2 >> PBITS);
  } else {
    const uint8_t v = (uint8_t)clip8(a0 >> PBITS);
    o[0] = v; o[1] = v; o[2] = v;
  }
}

// out[n][3][R][R] fp32 = ((u8 / 255) - 0.5) / 0.5 of the vertical pass over tmp (row index relative to row0)
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int R, int row0, int nrows,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk, int ks,
                                                              float* __restrict__ out) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, n = blockIdx.z;
  if (x >= R) return;
  const int lo = bounds[2 * y] - row0, cnt = bounds[2 * y + 1];
  const int* k = kk + (int64_t)y * ks;
  const uint8_t* col = tmp + (((int64_t)n * nrows + lo) * R + x) * 3;
  int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < cnt; ++t) {
    const int w = k[t];
    const uint8_t* p = col + (int64_t)t * R * 3;
    a0 += p[0] * w;
    a1 += p[1] * w;
    a2 += p[2] * w;
  }
  const int v[3] = {clip8(a0 >> PBITS), clip8(a1 >> PBITS), clip8(a2 >> PBITS)};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float u = __fdiv_rn((float)v[c], 255.0f);      // ToTensor: correctly rounded fp32 division, as torch
    out[(((int64_t)n * 3 + c) * R + y) * R + x] = __fdiv_rn(u - 0.5f, 0.5f);  // Normalize
  }
}

}  // namespace

extern "C" int vae_preprocess_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t C, int32_t R,
                                 const int32_t* bounds_x, const int32_t* kk_x, int32_t ksx,
                                 const int32_t* bounds_y, const int32_t* kk_y, int32_t ksy,
                                 int32_t row0, int32_t nrows, uint8_t* tmp, float* out, void* stream) {
  VAE_CHECK(src && bounds_x && kk_x && bounds_y && kk_y && tmp && out, "vae_preprocess_u8: null pointer");
  VAE_CHECK(n > 0 && H > 0 && W > 0 && R > 0 && (C == 1 || C == 3), "vae_preprocess_u8: bad shape n=%d H=%d W=%d C=%d R=%d", n, H, W, C, R);
  VAE_CHECK(ksx > 0 && ksy > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= H, "vae_preprocess_u8: bad row window %d+%d of %d", row0, nrows, H);
  VAE_CHECK(n <= 65535 && nrows <= 65535 && R <= 65535, "vae_preprocess_u8: grid dimension too large");
  hipStream_t st = (hipStream_t)stream;
  dim3 gh((unsigned)((R + 255) / 256), (unsigned)nrows, (unsigned)n), gv((unsigned)((R + 255) / 256), (unsigned)R, (unsigned)n);
  if (C == 3) hipLaunchKernelGGL(resample_h_kernel<3>, gh, dim3(256), 0, st, src, H, W, R, row0, nrows, bounds_x, kk_x, ksx, tmp);
  else hipLaunchKernelGGL(resample_h_kernel<1>, gh, dim3(256), 0, st, src, H, W, R, row0, nrows, bounds_x, kk_x, ksx, tmp);
  hipLaunchKernelGGL(resample_v_norm_kernel, gv, dim3(256), 0, st, tmp, R, row0, nrows, bounds_y, kk_y, ksy, out);
  VAE_LAUNCH_CHECK("vae_preprocess_u8");
  return VAE_OK;
}
