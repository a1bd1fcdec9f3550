// Streaming kernels of the train step: attention softmax, posterior sample + KL, MSE,
// layout conversion, nearest-up2x dgrad pooling, grad-norm, fused clip+AdamW, dead-weight scan.
// All reductions are two-stage (per-workgroup partial, fixed-order final) => bitwise reproducible.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// ---------------- softmax ----------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int cols) {
  __shared__ float red[4];
  float* row = S + (int64_t)blockIdx.x * cols;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < cols; j += 256) mx = fmaxf(mx, row[j]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) {
    float e = __expf(row[j] - mx);
    row[j] = e;
    s += e;
  }
  s = block_sum(s, red);
  const float inv = 1.0f / s;
  for (int j = threadIdx.x; j < cols; j += 256) row[j] *= inv;
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, int cols) {
  __shared__ float red[4];
  const float* p = P + (int64_t)blockIdx.x * cols;
  float* d = dP + (int64_t)blockIdx.x * cols;
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) s += p[j] * d[j];
  s = block_sum(s, red);
  for (int j = threadIdx.x; j < cols; j += 256) d[j] = p[j] * (d[j] - s);
}

// ---------------- posterior sample + KL ----------------
__global__ __launch_bounds__(256) void sample_kl_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                                        int hw, int L, float* __restrict__ z,
                                                        float* __restrict__ kl_partial) {
  __shared__ float red[4];
  const int b = blockIdx.y, nblk = gridDim.x;
  const int i = blockIdx.x * 256 + threadIdx.x;  // over hw*L
  float kl = 0.f;
  if (i < hw * L) {
    const int p = i / L, l = i - p * L;
    const float* m = mom + ((int64_t)b * hw + p) * (2 * L);
    const float mu = m[l];
    const float lv = fminf(fmaxf(m[L + l], -30.f), 20.f);
    const float sd = expf(0.5f * lv), var = expf(lv);
    const float e = eps ? eps[((int64_t)b * hw + p) * L + l] : 0.f;
    z[((int64_t)b * hw + p) * L + l] = mu + sd * e;
    kl = 0.5f * (mu * mu + var - 1.0f - lv);
  }
  kl = block_sum(kl, red);
  if (threadIdx.x == 0) kl_partial[b * nblk + blockIdx.x] = kl;
}

__global__ __launch_bounds__(256) void sample_kl_bwd_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                                            const float* __restrict__ dz, int64_t n, int L, float klw_over_b,
                                                            float* __restrict__ dmom) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B*hw*L
  if (i >= n) return;
  const int64_t p = i / L;
  const int l = (int)(i - p * L);
  const float* m = mom + p * (2 * L);
  const float mu = m[l], lvr = m[L + l];
  const bool inside = (lvr >= -30.f) && (lvr <= 20.f);
  const float lv = fminf(fmaxf(lvr, -30.f), 20.f);
  const float g = dz ? dz[i] : 0.f;
  const float e = eps ? eps[i] : 0.f;
  dmom[p * (2 * L) + l] = g + klw_over_b * mu;
  dmom[p * (2 * L) + L + l] = inside ? (g * e * 0.5f * expf(0.5f * lv) + klw_over_b * 0.5f * (expf(lv) - 1.0f)) : 0.f;
}

// ---------------- MSE ----------------
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                          float* __restrict__ ws) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s += d * d;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ mse_ws, int mse_nblk, double mse_n,
                                                         const float* __restrict__ kl_partial, int B, int kl_nblk,
                                                         float kl_weight, float* __restrict__ scalars) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < mse_nblk; i += 256) s += (double)mse_ws[i];
  s = block_sum_d(s, red);
  double k = 0.0;
  for (int i = threadIdx.x; i < B * kl_nblk; i += 256) k += (double)kl_partial[i];
  k = block_sum_d(k, red);
  if (threadIdx.x == 0) {
    const float mse = (float)(s / mse_n);
    const float kl = (float)(k / (double)B);
    scalars[0] = mse;
    scalars[1] = kl;
    scalars[2] = mse + kl_weight * kl;
  }
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      float two_over_n, float* __restrict__ d) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    d[i] = two_over_n * (a[i] - b[i]);
}

// ---------------- layout ----------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int HW, int Cpad, int64_t npix,
                                                           float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (b, pix)
  if (i >= npix) return;
  const int64_t b = i / HW, p = i - b * HW;
  for (int c = 0; c < Cpad; ++c) dst[i * Cpad + c] = (c < C) ? src[(b * C + c) * HW + p] : 0.f;
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int C, int HW, int64_t npix,
                                                           float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int64_t b = i / HW, p = i - b * HW;
  for (int c = 0; c < C; ++c) dst[(b * C + c) * HW + p] = src[i * C + c];
}

__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float* __restrict__ src, int H, int W, int Q, int64_t n4,
                                                         float* __restrict__ dst) {
  // dst [B][H][W][C], src [B][2H][2W][C]; i indexes float4 of dst
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % Q);
    int64_t t = i / Q;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int64_t b = t / H;
    const int64_t W2 = 2 * (int64_t)W;
    const float* s00 = src + (((b * 2 * H + 2 * y) * W2 + 2 * x) * Q + q) * 4;
    const float* s10 = s00 + W2 * Q * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(s00) + *reinterpret_cast<const f32x4*>(s00 + Q * 4) +
              *reinterpret_cast<const f32x4*>(s10) + *reinterpret_cast<const f32x4*>(s10 + Q * 4);
    *reinterpret_cast<f32x4*>(dst + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float* __restrict__ o) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = a[i] + b[i];
}

// bf16 tensors: o = bf16(float(a) + float(b)), 4 elements per thread per trip
__device__ __forceinline__ f32x4 unpack4_bf16(uint2 r) {
  f32x4 v;
  v[0] = __builtin_bit_cast(float, r.x << 16);
  v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, r.y << 16);
  v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
  return v;
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b, int64_t n4,
                                                       unsigned short* __restrict__ o) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = unpack4_bf16(*reinterpret_cast<const uint2*>(a + i * 4)) + unpack4_bf16(*reinterpret_cast<const uint2*>(b + i * 4));
    bf16x4_t h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    *reinterpret_cast<uint2*>(o + i * 4) = __builtin_bit_cast(uint2, h);
  }
}
__global__ __launch_bounds__(256) void unpack_bf16_kernel(const unsigned short* __restrict__ src, int64_t n, float* __restrict__ dst) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    *reinterpret_cast<f32x4*>(dst + i * 4) = unpack4_bf16(*reinterpret_cast<const uint2*>(src + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    dst[i] = __builtin_bit_cast(float, (unsigned)src[i] << 16);
  }
}

// fp32 -> bf16 image (round to nearest even), 8 elements per thread per trip: 32 B in, 16 B out
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, int64_t n, unsigned short* __restrict__ dst) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    bf16x8_t h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (__bf16)a[e];
      h[4 + e] = (__bf16)b[e];
    }
    *reinterpret_cast<uint4*>(dst + i * 8) = __builtin_bit_cast(uint4, h);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (n8 << 3) + threadIdx.x;
    const __bf16 h = (__bf16)src[i];
    dst[i] = __builtin_bit_cast(unsigned short, h);
  }
}

// effective 3x3 kernels of conv3x3(nearest_upsample_2x(x)) on the low-resolution x, one per output parity (vaehip.h)
__global__ __launch_bounds__(256) void upconv_phase_weights_kernel(const float* __restrict__ W, int Co, int Ci, float* __restrict__ We) {
  const int64_t n = (int64_t)Co * Ci;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i / Ci), ci = (int)(i - (int64_t)co * Ci);
    float w[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) w[kh][kw] = W[((int64_t)co * 9 + kh * 3 + kw) * Ci + ci];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // parity a: upsampled rows 2i+a+kh-1 -> low rows {i-1: kh0 | i: kh1,kh2} (a=0), {i: kh0,kh1 | i+1: kh2} (a=1)
        float r[3][3];  // row-combined [kh'][kw]
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          r[0][kw] = a == 0 ? w[0][kw] : 0.f;
          r[1][kw] = a == 0 ? w[1][kw] + w[2][kw] : w[0][kw] + w[1][kw];
          r[2][kw] = a == 0 ? 0.f : w[2][kw];
        }
        float* o = We + (((int64_t)(a * 2 + b) * Co + co) * 9) * Ci + ci;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          o[(int64_t)(kh * 3 + 0) * Ci] = b == 0 ? r[kh][0] : 0.f;
          o[(int64_t)(kh * 3 + 1) * Ci] = b == 0 ? r[kh][1] + r[kh][2] : r[kh][0] + r[kh][1];
          o[(int64_t)(kh * 3 + 2) * Ci] = b == 0 ? 0.f : r[kh][2];
        }
      }
  }
}

// transpose of upconv_phase_weights_kernel: every original tap (kh,kw) collects the effective tap it went into, in each phase
__global__ __launch_bounds__(256) void upconv_fold_wgrad_kernel(const float* __restrict__ dWe, const float* __restrict__ dbe, int Co,
                                                                int Ci, float* __restrict__ dW, float* __restrict__ db) {
  const int64_t n = (int64_t)Co * Ci, slab = (int64_t)Co * 9 * Ci;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i / Ci), ci = (int)(i - (int64_t)co * Ci);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int khe = a == 0 ? (kh == 0 ? 0 : 1) : (kh == 2 ? 2 : 1);
            const int kwe = b == 0 ? (kw == 0 ? 0 : 1) : (kw == 2 ? 2 : 1);
            t += dWe[(a * 2 + b) * slab + ((int64_t)co * 9 + khe * 3 + kwe) * Ci + ci];
          }
        dW[((int64_t)co * 9 + kh * 3 + kw) * Ci + ci] = t;
      }
    if (db && dbe && ci == 0) db[co] = (dbe[co] + dbe[Co + co]) + (dbe[2 * Co + co] + dbe[3 * Co + co]);
  }
}

// ---------------- grad norm + AdamW ----------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ ws) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)ws[i];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) out[0] = (float)s;
}

struct AdamArgs {
  float max_norm, decay, omb1, beta2, omb2, step_size, bc2_sqrt, eps;
};
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ sqnorm,
                                                    AdamArgs a) {
  float clip = 1.0f;
  if (a.max_norm > 0.f) clip = fminf(1.0f, a.max_norm / (sqrtf(sqnorm[0]) + 1e-6f));
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * clip;
      float pp = pv[e] * a.decay;
      const float mm = mv[e] + (gg - mv[e]) * a.omb1;
      const float v2 = vv[e] * a.beta2 + a.omb2 * gg * gg;
      const float denom = sqrtf(v2) / a.bc2_sqrt + a.eps;
      pp = pp - a.step_size * (mm / denom);
      pv[e] = pp; mv[e] = mm; vv[e] = v2;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * clip;
    float pp = p[i] * a.decay;
    const float mm = m[i] + (gg - m[i]) * a.omb1;
    const float v2 = v[i] * a.beta2 + a.omb2 * gg * gg;
    const float denom = sqrtf(v2) / a.bc2_sqrt + a.eps;
    p[i] = pp - a.step_size * (mm / denom);
    m[i] = mm;
    v[i] = v2;
  }
}

// ---------------- dead-weight scan ----------------
// One launch over every parameter segment: the segments are cut into chunks of DEAD_CHUNK elements, one workgroup per chunk
// (a 2.4 M-element conv weight is 72 chunks; one workgroup per SEGMENT left 36 large segments on 36 CUs), and a second
// small kernel adds each segment's chunk partials in chunk order (fixed order: reproducible counts AND sums).
// seg_chunk0 [nseg+1]: prefix sum of ceil(len/DEAD_CHUNK) (host); a workgroup finds its segment by bisection.
constexpr int DEAD_CHUNK = 32768;

__device__ __forceinline__ int dead_find_segment(const int32_t* __restrict__ seg_chunk0, int nseg, int chunk) {
  int lo = 0, hi = nseg;  // seg_chunk0[lo] <= chunk < seg_chunk0[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_chunk0[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

// ADAPT: count |w| < athr[seg] (and < thr when use_fixed) instead of |w| < thr, no sum
template <bool ADAPT>
__global__ __launch_bounds__(256) void dead_scan_chunk_kernel(const float* __restrict__ w, const int64_t* __restrict__ seg_off,
                                                              const int32_t* __restrict__ seg_chunk0, int nseg, float thr, int use_fixed,
                                                              const float* __restrict__ athr, unsigned long long* __restrict__ pcnt,
                                                              double* __restrict__ psum) {
  __shared__ double red[4];
  const int chunk = blockIdx.x;
  const int s = dead_find_segment(seg_chunk0, nseg, chunk);
  const int64_t b = seg_off[2 * s] + (int64_t)(chunk - seg_chunk0[s]) * DEAD_CHUNK;
  const int64_t e = min(seg_off[2 * s + 1], b + DEAD_CHUNK);
  const float at = ADAPT ? athr[s] : 0.f;
  double sum = 0.0;
  unsigned cnt = 0;
  // segment starts are 32-byte aligned in the arena and DEAD_CHUNK is a multiple of 4: float4 body, scalar tail
  const int64_t n4 = (e - b) >> 2;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w + b);
  const bool vec = ((reinterpret_cast<uintptr_t>(w + b)) & 15u) == 0;
  if (vec) {
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = w4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = fabsf(v[k]);
        if (ADAPT) cnt += ((a < at) && (!use_fixed || a < thr)) ? 1u : 0u;
        else { cnt += (a < thr) ? 1u : 0u; sum += (double)a; }
      }
    }
  }
  for (int64_t i = b + (vec ? (n4 << 2) : 0) + threadIdx.x; i < e; i += 256) {
    const float a = fabsf(w[i]);
    if (ADAPT) cnt += ((a < at) && (!use_fixed || a < thr)) ? 1u : 0u;
    else { cnt += (a < thr) ? 1u : 0u; sum += (double)a; }
  }
  const double c = block_sum_d((double)cnt, red);  // exact: counts are far below 2^53
  if (!ADAPT) sum = block_sum_d(sum, red);
  if (threadIdx.x == 0) {
    pcnt[chunk] = (unsigned long long)(c + 0.5);
    if (!ADAPT) psum[chunk] = sum;
  }
}

__global__ __launch_bounds__(64) void dead_scan_final_kernel(const int32_t* __restrict__ seg_chunk0, const unsigned long long* __restrict__ pcnt,
                                                             const double* __restrict__ psum, unsigned long long* __restrict__ counts,
                                                             double* __restrict__ abssum) {
  const int s = blockIdx.x;
  if (threadIdx.x != 0) return;
  unsigned long long c = 0;
  double sum = 0.0;
  for (int k = seg_chunk0[s]; k < seg_chunk0[s + 1]; ++k) {  // chunk order
    c += pcnt[k];
    if (psum) sum += psum[k];
  }
  counts[s] = c;
  if (abssum) abssum[s] = sum;
}

}  // namespace

extern "C" int vae_softmax_rows(float* S, int64_t rows, int32_t cols, void* stream) {
  VAE_CHECK(S && rows > 0 && rows < (1ll << 31) && cols > 0, "softmax_rows: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, cols);
  VAE_LAUNCH_CHECK("softmax_rows");
  return VAE_OK;
}
extern "C" int vae_softmax_bwd_rows(const float* P, float* dP, int64_t rows, int32_t cols, void* stream) {
  VAE_CHECK(P && dP && rows > 0 && rows < (1ll << 31) && cols > 0, "softmax_bwd_rows: bad args");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, P, dP, cols);
  VAE_LAUNCH_CHECK("softmax_bwd_rows");
  return VAE_OK;
}

extern "C" int vae_sample_kl(const float* moments, const float* eps, int32_t B, int32_t hw, int32_t L, float* z,
                             float* kl_partial, void* stream) {
  VAE_CHECK(moments && z && kl_partial && B > 0 && hw > 0 && L > 0 && B <= 65535, "sample_kl: bad args");
  const int nblk = (hw * L + 255) / 256;
  hipLaunchKernelGGL(sample_kl_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, moments, eps, hw, L, z, kl_partial);
  VAE_LAUNCH_CHECK("sample_kl");
  return VAE_OK;
}
extern "C" int vae_sample_kl_bwd(const float* moments, const float* eps, const float* dz, int32_t B, int32_t hw, int32_t L,
                                 float kl_weight, float* dmoments, void* stream) {
  VAE_CHECK(moments && dmoments && B > 0 && hw > 0 && L > 0, "sample_kl_bwd: bad args");
  const int64_t n = (int64_t)B * hw * L;
  hipLaunchKernelGGL(sample_kl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, eps,
                     dz, n, L, kl_weight / (float)B, dmoments);
  VAE_LAUNCH_CHECK("sample_kl_bwd");
  return VAE_OK;
}

extern "C" int vae_mse_partial(const float* recon, const float* target, int64_t n, float* ws, int32_t nblk, void* stream) {
  VAE_CHECK(recon && target && ws && n > 0 && nblk > 0 && nblk <= 65535, "mse_partial: bad args");
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, recon, target, n, ws);
  VAE_LAUNCH_CHECK("mse_partial");
  return VAE_OK;
}
extern "C" int vae_loss_final(const float* mse_ws, int32_t mse_nblk, int64_t mse_n, const float* kl_partial, int32_t B,
                              int32_t kl_nblk, float kl_weight, float* scalars, void* stream) {
  VAE_CHECK(mse_ws && kl_partial && scalars && mse_nblk > 0 && mse_n > 0 && B > 0 && kl_nblk > 0, "loss_final: bad args");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mse_ws, mse_nblk, (double)mse_n, kl_partial,
                     B, kl_nblk, kl_weight, scalars);
  VAE_LAUNCH_CHECK("loss_final");
  return VAE_OK;
}
extern "C" int vae_mse_bwd(const float* recon, const float* target, int64_t n, float scale, float* drecon, void* stream) {
  VAE_CHECK(recon && target && drecon && n > 0, "mse_bwd: bad args");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, recon, target, n,
                     (float)(2.0 * (double)scale / (double)n), drecon);
  VAE_LAUNCH_CHECK("mse_bwd");
  return VAE_OK;
}

extern "C" int vae_nchw_to_nhwc(const float* src, int32_t B, int32_t C, int32_t HW, int32_t Cpad, float* dst, void* stream) {
  VAE_CHECK(src && dst && B > 0 && C > 0 && HW > 0 && Cpad >= C && Cpad <= 64, "nchw_to_nhwc: bad args");
  const int64_t npix = (int64_t)B * HW;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, C, HW,
                     Cpad, npix, dst);
  VAE_LAUNCH_CHECK("nchw_to_nhwc");
  return VAE_OK;
}
extern "C" int vae_nhwc_to_nchw(const float* src, int32_t B, int32_t C, int32_t HW, float* dst, void* stream) {
  VAE_CHECK(src && dst && B > 0 && C > 0 && HW > 0 && C <= 64, "nhwc_to_nchw: bad args");
  const int64_t npix = (int64_t)B * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, C, HW,
                     npix, dst);
  VAE_LAUNCH_CHECK("nhwc_to_nchw");
  return VAE_OK;
}
extern "C" int vae_sumpool2x2(const float* src, int32_t B, int32_t H, int32_t W, int32_t C, float* dst, void* stream) {
  VAE_CHECK(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && aligned16(src) && aligned16(dst),
            "sumpool2x2: bad args");
  const int64_t n4 = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, src, H, W, C / 4, n4, dst);
  VAE_LAUNCH_CHECK("sumpool2x2");
  return VAE_OK;
}
extern "C" int vae_add(const float* a, const float* b, int64_t n, float* out, void* stream) {
  VAE_CHECK(a && b && out && n > 0, "add: bad args");
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
  VAE_LAUNCH_CHECK("add");
  return VAE_OK;
}

extern "C" int vae_add_bf16(const void* a, const void* b, int64_t n, void* out, void* stream) {
  VAE_CHECK(a && b && out && n > 0 && n % 4 == 0, "add_bf16: bad args (n %% 4 == 0)");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                     (const unsigned short*)b, n / 4, (unsigned short*)out);
  VAE_LAUNCH_CHECK("add_bf16");
  return VAE_OK;
}
extern "C" int vae_unpack_bf16(const void* src16, int64_t n, float* dst, void* stream) {
  VAE_CHECK(src16 && dst && n > 0 && (((uintptr_t)src16) & 7u) == 0 && aligned16(dst), "unpack_bf16: bad args");
  hipLaunchKernelGGL(unpack_bf16_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src16, n, dst);
  VAE_LAUNCH_CHECK("unpack_bf16");
  return VAE_OK;
}

extern "C" int vae_upconv_phase_weights(const float* W, int32_t Co, int32_t Ci, float* Weff, void* stream) {
  VAE_CHECK(W && Weff && Co > 0 && Ci > 0, "upconv_phase_weights: bad args");
  hipLaunchKernelGGL(upconv_phase_weights_kernel, dim3(ew_blocks((int64_t)Co * Ci)), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, Weff);
  VAE_LAUNCH_CHECK("upconv_phase_weights");
  return VAE_OK;
}

extern "C" int vae_upconv_fold_wgrad(const float* dWeff, const float* dbeff, int32_t Co, int32_t Ci, float* dW, float* db, void* stream) {
  VAE_CHECK(dWeff && dW && Co > 0 && Ci > 0, "upconv_fold_wgrad: bad args");
  hipLaunchKernelGGL(upconv_fold_wgrad_kernel, dim3(ew_blocks((int64_t)Co * Ci)), dim3(256), 0, (hipStream_t)stream, dWeff, dbeff, Co,
                     Ci, dW, db);
  VAE_LAUNCH_CHECK("upconv_fold_wgrad");
  return VAE_OK;
}

extern "C" int vae_pack_bf16(const float* src, int64_t n, void* dst, void* stream) {
  VAE_CHECK(src && dst && n > 0 && aligned16(src) && aligned16(dst), "pack_bf16: bad args");
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(ew_blocks((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, src, n,
                     reinterpret_cast<unsigned short*>(dst));
  VAE_LAUNCH_CHECK("pack_bf16");
  return VAE_OK;
}

extern "C" int vae_sqnorm(const float* g, int64_t n, float* ws, int32_t nblk, float* out, void* stream) {
  VAE_CHECK(g && ws && out && n > 0 && nblk > 0 && nblk <= 65535 && aligned16(g), "sqnorm: bad args");
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, n, ws);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, nblk, out);
  VAE_LAUNCH_CHECK("sqnorm");
  return VAE_OK;
}

extern "C" int vae_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm, float max_norm, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  VAE_CHECK(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
  VAE_CHECK(max_norm <= 0.f || sqnorm != nullptr, "adamw: clipping needs sqnorm");
  VAE_CHECK(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw: unaligned");
  // scalar prep exactly as torch.optim.adamw._single_tensor_adamw (python doubles)
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamArgs a;
  a.max_norm = max_norm;
  a.decay = (float)(1.0 - (double)lr * (double)weight_decay);
  a.omb1 = (float)(1.0 - (double)beta1);
  a.beta2 = beta2;
  a.omb2 = (float)(1.0 - (double)beta2);
  a.step_size = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = eps;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sqnorm, a);
  VAE_LAUNCH_CHECK("adamw");
  return VAE_OK;
}

extern "C" int vae_dead_scan_chunk(void) { return DEAD_CHUNK; }

extern "C" int vae_dead_scan(const float* w, const int64_t* seg_off, const int32_t* seg_chunk0, int32_t nseg, int32_t nchunk, float thr,
                             unsigned long long* part_counts, double* part_abssum, unsigned long long* out_counts,
                             double* out_abssum, void* stream) {
  VAE_CHECK(w && seg_off && seg_chunk0 && part_counts && part_abssum && out_counts && out_abssum && nseg > 0 && nchunk >= nseg,
            "dead_scan: bad args");
  hipLaunchKernelGGL(dead_scan_chunk_kernel<false>, dim3(nchunk), dim3(256), 0, (hipStream_t)stream, w, seg_off, seg_chunk0, nseg, thr, 0,
                     (const float*)nullptr, part_counts, part_abssum);
  hipLaunchKernelGGL(dead_scan_final_kernel, dim3(nseg), dim3(64), 0, (hipStream_t)stream, seg_chunk0, part_counts,
                     (const double*)part_abssum, out_counts, out_abssum);
  VAE_LAUNCH_CHECK("dead_scan");
  return VAE_OK;
}
extern "C" int vae_dead_scan_adaptive(const float* w, const int64_t* seg_off, const int32_t* seg_chunk0, int32_t nseg, int32_t nchunk,
                                      float thr, int32_t use_fixed, const float* adaptive_thr, unsigned long long* part_counts,
                                      unsigned long long* out_counts, void* stream) {
  VAE_CHECK(w && seg_off && seg_chunk0 && part_counts && out_counts && adaptive_thr && nseg > 0 && nchunk >= nseg, "dead_scan_adaptive: bad args");
  hipLaunchKernelGGL(dead_scan_chunk_kernel<true>, dim3(nchunk), dim3(256), 0, (hipStream_t)stream, w, seg_off, seg_chunk0, nseg, thr,
                     use_fixed, adaptive_thr, part_counts, (double*)nullptr);
  hipLaunchKernelGGL(dead_scan_final_kernel, dim3(nseg), dim3(64), 0, (hipStream_t)stream, seg_chunk0, part_counts,
                     (const double*)nullptr, out_counts, (double*)nullptr);
  VAE_LAUNCH_CHECK("dead_scan_adaptive");
  return VAE_OK;
}
