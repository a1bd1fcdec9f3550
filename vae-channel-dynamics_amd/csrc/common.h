// Shared device helpers for libvaehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vaehip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void vae_set_error(const char* fmt, ...);

#define VAE_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      vae_set_error(__VA_ARGS__);            \
      return VAE_EINVAL;                     \
    }                                        \
  } while (0)

#define VAE_LAUNCH_CHECK(name)                                                 \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      vae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
      return VAE_ELAUNCH;                                                      \
    }                                                                          \
  } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float silu_f(float u) {
  // u * sigmoid(u); v_exp_f32 + v_rcp_f32 (1 ulp), ~1e-7 relative; __frcp_rn would expand to a full IEEE division
  return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
}
__device__ __forceinline__ float silu_grad_f(float u) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// row (b,y,x) + tap (kh,kw) -> source pixel; returns validity (zero padding otherwise)
__device__ __forceinline__ bool src_pixel(const vae_conv_geom& g, int y, int x, int kh, int kw,
                                          int& sy, int& sx) {
  if (g.mode == VAE_MODE_FWD) {
    sy = y * g.stride + kh - g.pad_t;
    sx = x * g.stride + kw - g.pad_l;
    return ((unsigned)sy < (unsigned)g.Hs) && ((unsigned)sx < (unsigned)g.Ws);
  } else if (g.mode == VAE_MODE_UP2X) {
    int uy = y + kh - 1, ux = x + kw - 1;
    bool ok = ((unsigned)uy < (unsigned)(2 * g.Hs)) && ((unsigned)ux < (unsigned)(2 * g.Ws));
    sy = uy >> 1;
    sx = ux >> 1;
    return ok;
  } else {
    int ty = y + g.pad_t - kh, tx = x + g.pad_l - kw;
    if (ty < 0 || tx < 0) return false;
    if (g.stride == 2) {
      if ((ty | tx) & 1) return false;
      ty >>= 1;
      tx >>= 1;
    }
    sy = ty;
    sx = tx;
    return (sy < g.Hs) && (sx < g.Ws);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
