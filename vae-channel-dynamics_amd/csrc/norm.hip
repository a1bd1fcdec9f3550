// GroupNorm(32 groups) statistics, tracker reduction and backward for NHWC activations stored as fp32 or as bf16 (x_bf16:
// bf16 mode keeps conv outputs and the residual stream as bf16, as autocast does for the reference; statistics, sums and
// the arithmetic stay fp32).
// All kernels are HBM-bound streaming passes: lanes run across channels (float4 per lane,
// fully coalesced rows), each workgroup owns one pixel chunk of one image, and every
// cross-workgroup reduction is "per-workgroup partial + fixed-order final pass" so results
// are bitwise reproducible (the inactivity mask of classifier.py:135 depends on it).
#include "common.h"

namespace {

// 4 consecutive elements of a gradient tensor stored as fp32 or as bf16 (bf16 mode keeps the conv dgrad outputs and the
// GroupNorm-backward outputs that only feed convolutions as bf16: their consumers round them to bf16 anyway)
template <bool BF>
__device__ __forceinline__ f32x4 load4g(const void* base, int64_t idx4) {
  if (!BF) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx4 * 4);
  const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx4 * 4);
  f32x4 v;
  v[0] = __builtin_bit_cast(float, r.x << 16);
  v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, r.y << 16);
  v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
  return v;
}
__device__ __forceinline__ uint2 pack4_bf16(f32x4 v) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  bf16x4_t h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
  return __builtin_bit_cast(uint2, h);
}

// activation tensors: element quad `idx4` of a tensor stored as fp32 or bf16
template <bool BF>
__device__ __forceinline__ f32x4 load4x(const void* base, int64_t idx4) { return load4g<BF>(base, idx4); }
template <bool BF>
__device__ __forceinline__ void store4x(void* base, int64_t idx4, f32x4 v) {
  if (!BF) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx4 * 4) = v;
  else *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx4 * 4) = pack4_bf16(v);
}

// layout helper: 256 threads = PR pixel rows x Q float4 lanes (Q = C/4)
struct Lay {
  int Q, PR, cq, pr;
};
__device__ __forceinline__ Lay make_lay(int C) {
  Lay l;
  l.Q = C >> 2;
  l.PR = 256 / l.Q;
  l.cq = threadIdx.x % l.Q;
  l.pr = threadIdx.x / l.Q;
  return l;
}

template <bool XBF>
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const void* __restrict__ x, int HW, int C, int G,
                                                               int nchunk, float* __restrict__ ws) {
  __shared__ float red[3][256];
  const Lay l = make_lay(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int64_t xb4 = (int64_t)b * HW * l.Q;
  // shifted sums around the thread's first value (no cancellation), then centred moments, merged over the group's threads
  float pv = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
  for (int pix = p0 + l.pr; pix < p1; pix += l.PR) {
    f32x4 v = load4x<XBF>(x, xb4 + (int64_t)pix * l.Q + l.cq);
    if (cnt == 0.f) pv = v[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[e] - pv;
      s1 += d;
      s2 += d * d;
    }
    cnt += 4.f;
  }
  const MeanM2 mine = (cnt > 0.f) ? mm2_from_shifted(pv, s1, s2, cnt) : MeanM2{0.f, 0.f};
  red[0][threadIdx.x] = mine.m;
  red[1][threadIdx.x] = mine.M2;
  red[2][threadIdx.x] = cnt;
  __syncthreads();
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    const int lanes = (C / G) >> 2;  // float4 lanes per group
    MeanM2 acc{0.f, 0.f};
    float n = 0.f;
    for (int pr = 0; pr < l.PR; ++pr)
      for (int j = 0; j < lanes; ++j) {  // fixed order
        const int t = pr * l.Q + g * lanes + j;
        const float nt = red[2][t];
        if (nt > 0.f) {
          acc = (n > 0.f) ? mm2_merge(acc, n, MeanM2{red[0][t], red[1][t]}, nt) : MeanM2{red[0][t], red[1][t]};
          n += nt;
        }
      }
    float* o = ws + (((int64_t)b * nchunk + chunk) * G + g) * 2;
    o[0] = acc.m;
    o[1] = acc.M2;
  }
}

__global__ __launch_bounds__(256) void gn_stats_final_kernel(const float* __restrict__ ws, int HW, int C, int G,
                                                             int nchunk, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ float smean[64], srstd[64];
  __shared__ double sS[256], sQ[256], sMean[64];
  const int b = blockIdx.x;
  // ws[b][chunk][g] = (mean, M2) of chunk `chunk` (pixels [chunk*per, min(HW, (chunk+1)*per)) x the group's channels; a conv
  // epilogue leaves one chunk per output tile: hundreds to thousands).  Chan's merge in fp64: first the count-weighted mean, then
  // M2 = sum_k M2_k + n_k (mean_k - mean)^2.  Workgroup (b, y) finishes groups [y*gw, (y+1)*gw), gw = G / gridDim.y; the chunk
  // range of a group is dealt out to 256 / gw threads, fixed order (round 3: with one workgroup per image and 8 threads per
  // group the 4096 chunks of a 1024^2 image took 17 us x 400 launches per step).
  const int gw = G / (int)gridDim.y, g0 = (int)blockIdx.y * gw;
  const int parts = 256 / gw;  // gw <= 64
  const int per = (HW + nchunk - 1) / nchunk;
  const int cpg_ = C / G;
  const int gl = threadIdx.x % gw, part_ = threadIdx.x / gw;
  const int g_ = g0 + gl;
  {
    double S = 0.0;
    if (part_ < parts)
      for (int c = part_; c < nchunk; c += parts) {
        const float* o = ws + (((int64_t)b * nchunk + c) * G + g_) * 2;
        const double nk = (double)max(0, min(HW, (c + 1) * per) - c * per) * (double)cpg_;  // trailing chunks can start beyond HW: empty
        S += nk * (double)o[0];
      }
    sS[threadIdx.x] = S;
  }
  __syncthreads();
  if (threadIdx.x < gw) {
    double S = 0.0;
    for (int part = 0; part < parts; ++part) S += sS[part * gw + threadIdx.x];
    sMean[threadIdx.x] = S / ((double)HW * (double)cpg_);
  }
  __syncthreads();
  {
    const double m = sMean[gl];
    double Q = 0.0;
    if (part_ < parts)
      for (int c = part_; c < nchunk; c += parts) {
        const float* o = ws + (((int64_t)b * nchunk + c) * G + g_) * 2;
        const double nk = (double)max(0, min(HW, (c + 1) * per) - c * per) * (double)cpg_;
        const double d = (double)o[0] - m;
        Q += (double)o[1] + nk * d * d;
      }
    sQ[threadIdx.x] = Q;
  }
  __syncthreads();
  if (threadIdx.x < gw) {
    const int g = g0 + threadIdx.x;
    double Q = 0.0;
    for (int part = 0; part < parts; ++part) Q += sQ[part * gw + threadIdx.x];
    const double n = (double)HW * (double)(C / G);
    const double m = sMean[threadIdx.x];
    double var = Q / n;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    smean[threadIdx.x] = mf;
    srstd[threadIdx.x] = rf;
    mean[b * G + g] = mf;
    rstd[b * G + g] = rf;
  }
  __syncthreads();
  const int cpg = C / G;
  for (int cl = threadIdx.x; cl < gw * cpg; cl += 256) {
    const int c = g0 * cpg + cl;
    const float sc = srstd[cl / cpg] * gamma[c];
    scale[(int64_t)b * C + c] = sc;
    shift[(int64_t)b * C + c] = beta[c] - smean[cl / cpg] * sc;
  }
}

template <bool XBF>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int64_t n4, int HWQ, int Q,
                                                       int C, int xf, float* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    const int b = (int)(i / HWQ);
    const int c = (int)(i % Q) * 4;
    f32x4 v = load4x<XBF>(x, i);
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (int64_t)b * C + c);
    f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (int64_t)b * C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float u = v[e] * sc[e] + sh[e];
      o[e] = (xf == VAE_XF_AFFINE_SILU) ? silu_f(u) : u;
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

// same transform, output rounded to bf16 (the activation image the bf16 conv / wgrad kernels read): 8 elements per thread
template <bool XBF>
__global__ __launch_bounds__(256) void gn_apply_bf16_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int64_t n8, int HWQ, int Q,
                                                            int C, int xf, unsigned short* __restrict__ y) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n8; i += stride) {
    const int b = (int)(i / HWQ);
    const int c = (int)(i % Q) * 8;
    bf16x8_t h;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4 v = load4x<XBF>(x, i * 2 + half);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (int64_t)b * C + c + half * 4);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (int64_t)b * C + c + half * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = v[e] * sc[e] + sh[e];
        h[half * 4 + e] = (__bf16)((xf == VAE_XF_AFFINE_SILU) ? silu_f(u) : u);
      }
    }
    *reinterpret_cast<uint4*>(y + i * 8) = __builtin_bit_cast(uint4, h);
  }
}

// The backward apply pass is laid out by ROWS (C / 4 divides 256 for every GroupNorm of the model): one workgroup per (image,
// run of pixels), a thread owns ONE channel quad for the whole run, so mean / rstd / gamma / beta / the coefficients are loaded
// once instead of per element, there is no 64-bit division per element, and GN_UN independent quads are in flight per thread.
// Measured against the grid-stride form it replaced (tools/gn_stream_bench.py, batch 16): bf16 storage 329 -> 275 us at
// 256x256x128 (4.1 -> 4.9 TB/s), fp32 storage unchanged (5.3 -> 5.4 TB/s); the forward apply passes gained nothing from the
// same layout and keep the grid-stride form.  Same arithmetic per element.
constexpr int GN_UN = 4;
struct RowPlan { int per, nchunk; };
inline RowPlan row_plan(int B, int HW, int threads_per_pixel) {
  const int PR = 256 / threads_per_pixel;  // pixels per pass of the workgroup
  int per = PR * GN_UN * 4;                // 16 element groups per thread ...
  while (per > PR * GN_UN && (int64_t)B * ((HW + per - 1) / per) < 2048) per /= 2;  // ... fewer on small maps (>= 2048 workgroups)
  return {per, (HW + per - 1) / per};
}

template <bool XBF>
__global__ __launch_bounds__(256) void gn_track_partial_kernel(const void* __restrict__ x,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int HW, int C,
                                                               int nchunk, float* __restrict__ ws) {
  __shared__ f32x4 red[256];
  const Lay l = make_lay(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int64_t xb4 = (int64_t)b * HW * l.Q;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (int64_t)b * C + l.cq * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (int64_t)b * C + l.cq * 4);
  f32x4 s = {0, 0, 0, 0};
  for (int pix = p0 + l.pr; pix < p1; pix += l.PR) {
    f32x4 v = load4x<XBF>(x, xb4 + (int64_t)pix * l.Q + l.cq);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += fabsf(v[e] * sc[e] + sh[e]);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < l.Q) {
    f32x4 t = {0, 0, 0, 0};
    for (int pr = 0; pr < l.PR; ++pr) t += red[pr * l.Q + threadIdx.x];
    *reinterpret_cast<f32x4*>(ws + ((int64_t)b * nchunk + chunk) * C + threadIdx.x * 4) = t;
  }
}

__global__ __launch_bounds__(256) void track_final_kernel(const float* __restrict__ ws, int rows, int C,
                                                          float inv_count, float* __restrict__ out) {
  // one workgroup per 4 channels; thread t owns rows t, t+256, ... (fixed order), then a fixed LDS tree
  __shared__ double red[256][4];
  const int c0 = blockIdx.x * 4;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int r = threadIdx.x; r < rows; r += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < C) s[e] += (double)ws[(int64_t)r * C + c0 + e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[threadIdx.x][e] += red[threadIdx.x + o][e];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4 && c0 + (int)threadIdx.x < C) out[c0 + threadIdx.x] = (float)(red[0][threadIdx.x] * (double)inv_count);
}

template <bool SILU, bool GBF, bool XBF>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const void* __restrict__ x, const void* __restrict__ g,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int HW, int C, int G,
                                                             int nchunk, float* __restrict__ ws) {
  __shared__ f32x4 red[2][256];
  const Lay l = make_lay(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int c = l.cq * 4, grp = c / (C / G);
  const float mu = mean[b * G + grp], rs = rstd[b * G + grp];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
  const int64_t gb4 = (int64_t)b * HW * (C / 4);
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  for (int pix = p0 + l.pr; pix < p1; pix += l.PR) {
    f32x4 v = load4x<XBF>(x, gb4 + (int64_t)pix * (C / 4) + l.cq);
    f32x4 gv = load4g<GBF>(g, gb4 + (int64_t)pix * (C / 4) + l.cq);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      float du = gv[e];
      if (SILU) du *= silu_grad_f(xh * ga[e] + be[e]);
      s1[e] += du;
      s2[e] += du * xh;
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < l.Q) {
    f32x4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
    for (int pr = 0; pr < l.PR; ++pr) {
      t1 += red[0][pr * l.Q + threadIdx.x];
      t2 += red[1][pr * l.Q + threadIdx.x];
    }
    float* o = ws + (((int64_t)b * nchunk + chunk) * C + threadIdx.x * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e * 2 + 0] = t1[e];
      o[e * 2 + 1] = t2[e];
    }
  }
}

// stage 2b (same launch as 2a, workgroups B ..): dgamma/dbeta = sum over batch items and chunks of the stage-1 partials.
// A workgroup takes 8 channels; 32 threads per channel split the B * nchunk rows (row r to thread r % 32), fp64, and
// thread 0 of a channel adds the 32 partial sums in order: a fixed association.  (It cannot use 2a's per-image totals:
// the two roles run concurrently.)
__device__ __forceinline__ void gn_bwd_dparam_role(const float* __restrict__ ws, int blk, int B, int C, int nchunk,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, double* red /*[2][256]*/) {
  const int cl = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int c = blk * 8 + cl;
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    const int rows = B * nchunk;
    for (int r = part; r < rows; r += 32) {
      const float* o = ws + ((int64_t)r * C + c) * 2;
      a1 += (double)o[0];
      a2 += (double)o[1];
    }
  }
  red[threadIdx.x] = a1;
  red[256 + threadIdx.x] = a2;
  __syncthreads();
  if (part == 0 && c < C) {
#pragma unroll 8
    for (int j = 1; j < 32; ++j) {
      a1 += red[j * 8 + cl];
      a2 += red[256 + j * 8 + cl];
    }
    dbeta[c] = (float)a1;
    dgamma[c] = (float)a2;
  }
}

// stage 2a: workgroup (b, group block gb): chunk totals per (b, c) and the group coefficients of groups [gb*GPW, (gb+1)*GPW).
// The block's channels (GPW * cpg <= 64) each get 256 / (GPW * cpg) threads that split the chunk range (thread t of a channel
// takes chunks t, t + T, ...), fp64, then thread 0 of the channel adds the T partial sums in order: a fixed association.
// (Round 3: with one workgroup per image and a serial loop over 512 chunks the launch took 40-200 us at batch 2.)
constexpr int GN_GPW = 4;  // groups per workgroup of stage 2a
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(const float* __restrict__ ws, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, int B, int HW, int C, int G,
                                                           int nchunk, float* __restrict__ coef, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  __shared__ __attribute__((aligned(16))) double sred[2][256];
  __shared__ float sg1[64], sg2[64];
  const int cpg = C / G;
  const int gblocks = G / GN_GPW, nwa = B * gblocks;
  if ((int)blockIdx.x >= nwa) {  // uniform per workgroup
    gn_bwd_dparam_role(ws, blockIdx.x - nwa, B, C, nchunk, dgamma, dbeta, &sred[0][0]);  // 512 doubles
    return;
  }
  const int b = blockIdx.x / gblocks, gb = blockIdx.x % gblocks;
  const int nc = GN_GPW * cpg;           // channels of the block: 16, 32 or 64
  const int T = 256 / nc;                // threads per channel
  const int cl = threadIdx.x % nc, part = threadIdx.x / nc;
  const int c = gb * nc + cl;
  double a1 = 0.0, a2 = 0.0;
  for (int k = part; k < nchunk; k += T) {
    const float* o = ws + (((int64_t)b * nchunk + k) * C + c) * 2;
    a1 += (double)o[0];
    a2 += (double)o[1];
  }
  sred[0][threadIdx.x] = a1;
  sred[1][threadIdx.x] = a2;
  __syncthreads();
  if (part == 0) {
    for (int j = 1; j < T; ++j) {
      a1 += sred[0][j * nc + cl];
      a2 += sred[1][j * nc + cl];
    }
    sg1[cl] = (float)(a1 * (double)gamma[c]);
    sg2[cl] = (float)(a2 * (double)gamma[c]);
  }
  __syncthreads();
  if (threadIdx.x < GN_GPW) {
    const int g = gb * GN_GPW + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      s1 += (double)sg1[threadIdx.x * cpg + j];
      s2 += (double)sg2[threadIdx.x * cpg + j];
    }
    const double n = (double)HW * (double)cpg;
    const double r = (double)rstd[b * G + g];
    coef[((int64_t)b * G + g) * 2 + 0] = (float)(r * s2 / n);
    coef[((int64_t)b * G + g) * 2 + 1] = (float)(r * s1 / n);
  }
}
template <bool SILU, bool GBF, bool XBF>
__global__ __launch_bounds__(256) void gn_bwd_apply_rows_kernel(const void* __restrict__ x, const void* __restrict__ g,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ coef, const void* __restrict__ add,
                                                                int HW, int C, int G, int per, float* __restrict__ dx,
                                                                unsigned short* __restrict__ dx16) {
  const int Q = C >> 2, PR = 256 / Q, q = threadIdx.x % Q, r = threadIdx.x / Q;
  const int b = blockIdx.y, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  const int c = q * 4, grp = c / (C / G);
  const float mu = mean[b * G + grp], rs = rstd[b * G + grp];
  const float k0 = coef[((int64_t)b * G + grp) * 2 + 0], k1 = coef[((int64_t)b * G + grp) * 2 + 1];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  const int64_t img = (int64_t)b * HW * Q + q;
  for (int p = p0 + r; p < p1; p += PR * GN_UN) {
    f32x4 v[GN_UN], gv[GN_UN], o[GN_UN];
#pragma unroll
    for (int u = 0; u < GN_UN; ++u) {
      const int64_t i = img + (int64_t)min(p + u * PR, p1 - 1) * Q;  // (a tail re-reads the last pixel)
      v[u] = load4x<XBF>(x, i);
      gv[u] = load4g<GBF>(g, i);
      if (add) o[u] = load4x<XBF>(add, i);  // (the residual-path gradient is stored like x)
      else o[u] = f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < GN_UN; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[u][e] - mu) * rs;
        float du = gv[u][e];
        if (SILU) du *= silu_grad_f(xh * ga[e] + be[e]);
        o[u][e] += du * (rs * ga[e]) - xh * k0 - k1;
      }
      if (p + u * PR < p1) {
        const int64_t i = img + (int64_t)(p + u * PR) * Q;
        if (dx) *reinterpret_cast<f32x4*>(dx + i * 4) = o[u];
        if (dx16) *reinterpret_cast<uint2*>(dx16 + i * 4) = pack4_bf16(o[u]);
      }
    }
  }
}

int check_gn(const char* who, int B, int HW, int C, int G, int nchunk) {
  VAE_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && nchunk > 0, "%s: non-positive size", who);
  VAE_CHECK(C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0, "%s: C=%d must be 4*{1,2,4,...,256}", who, C);
  VAE_CHECK(C % G == 0 && (C / G) % 4 == 0, "%s: channels per group must be a multiple of 4 (C=%d G=%d)", who, C, G);
  VAE_CHECK(G <= 64, "%s: G=%d > 64", who, G);
  VAE_CHECK(nchunk <= 65535 && B <= 65535, "%s: grid too large", who);
  return 0;
}
inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int vae_gn_stats_partial(const void* x, int32_t x_bf16, int32_t B, int32_t HW, int32_t C, int32_t G, int32_t nchunk,
                                    float* ws, void* stream) {
  if (int e = check_gn("gn_stats_partial", B, HW, C, G, nchunk)) return e;
  VAE_CHECK(x && ws && aligned16(x), "gn_stats_partial: bad pointers");
  if (x_bf16) hipLaunchKernelGGL(gn_stats_partial_kernel<true>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, HW, C, G, nchunk, ws);
  else hipLaunchKernelGGL(gn_stats_partial_kernel<false>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, HW, C, G, nchunk, ws);
  VAE_LAUNCH_CHECK("gn_stats_partial");
  return VAE_OK;
}

extern "C" int vae_gn_stats_final(const float* ws, int32_t B, int32_t HW, int32_t C, int32_t G, int32_t nchunk,
                                  const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                  float* scale, float* shift, void* stream) {
  if (int e = check_gn("gn_stats_final", B, HW, C, G, nchunk)) return e;
  VAE_CHECK(ws && gamma && beta && mean && rstd && scale && shift, "gn_stats_final: null pointer");
  // few images with many chunks each (large feature maps at small batch): several workgroups per image
  const int gy = (nchunk >= 256 && G % 8 == 0 && B < 64) ? 8 : 1;
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3(B, gy), dim3(256), 0, (hipStream_t)stream, ws, HW, C, G, nchunk, gamma, beta,
                     eps, mean, rstd, scale, shift);
  VAE_LAUNCH_CHECK("gn_stats_final");
  return VAE_OK;
}

extern "C" int vae_gn_apply(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW, int32_t C,
                            int32_t xf, float* y, void* stream) {
  VAE_CHECK(x && scale && shift && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "gn_apply: bad args");
  VAE_CHECK(aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift), "gn_apply: unaligned");
  const int Q = C / 4;
  const int64_t n4 = (int64_t)B * HW * Q;
  if (x_bf16) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, n4, HW * Q, Q, C, xf, y);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, n4, HW * Q, Q, C, xf, y);
  VAE_LAUNCH_CHECK("gn_apply");
  return VAE_OK;
}

extern "C" int vae_gn_apply_bf16(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW, int32_t C,
                                 int32_t xf, void* y16, void* stream) {
  VAE_CHECK(x && scale && shift && y16 && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "gn_apply_bf16: bad args (C %% 8 == 0)");
  VAE_CHECK(aligned16(x) && aligned16(y16) && aligned16(scale) && aligned16(shift), "gn_apply_bf16: unaligned");
  const int Q = C / 8;
  const int64_t n8 = (int64_t)B * HW * Q;
  if (x_bf16) hipLaunchKernelGGL(gn_apply_bf16_kernel<true>, dim3(ew_blocks(n8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, n8,
                                 HW * Q, Q, C, xf, reinterpret_cast<unsigned short*>(y16));
  else hipLaunchKernelGGL(gn_apply_bf16_kernel<false>, dim3(ew_blocks(n8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, n8,
                          HW * Q, Q, C, xf, reinterpret_cast<unsigned short*>(y16));
  VAE_LAUNCH_CHECK("gn_apply_bf16");
  return VAE_OK;
}

extern "C" int vae_gn_track_partial(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW,
                                    int32_t C, int32_t nchunk, float* ws, void* stream) {
  VAE_CHECK(B > 0 && HW > 0 && nchunk > 0 && nchunk <= 65535 && B <= 65535, "gn_track_partial: bad sizes");
  VAE_CHECK(C > 0 && C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0, "gn_track_partial: C=%d unsupported", C);
  VAE_CHECK(x && scale && shift && ws && aligned16(x) && aligned16(scale) && aligned16(shift) && aligned16(ws),
            "gn_track_partial: bad pointers");
  if (x_bf16) hipLaunchKernelGGL(gn_track_partial_kernel<true>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, scale, shift, HW, C, nchunk, ws);
  else hipLaunchKernelGGL(gn_track_partial_kernel<false>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, scale, shift, HW, C, nchunk, ws);
  VAE_LAUNCH_CHECK("gn_track_partial");
  return VAE_OK;
}

extern "C" int vae_track_final(const float* ws, int32_t rows, int32_t C, float inv_count, float* out, void* stream) {
  VAE_CHECK(ws && out && rows > 0 && C > 0, "track_final: bad args");
  hipLaunchKernelGGL(track_final_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, ws, rows, C, inv_count, out);
  VAE_LAUNCH_CHECK("track_final");
  return VAE_OK;
}

extern "C" int vae_gn_bwd_partial(const void* x, int32_t x_bf16, const void* g, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, int32_t B, int32_t HW, int32_t C, int32_t G,
                                  int32_t nchunk, int32_t silu, int32_t g_bf16, float* ws, void* stream) {
  if (int e = check_gn("gn_bwd_partial", B, HW, C, G, nchunk)) return e;
  VAE_CHECK(x && g && mean && rstd && gamma && beta && ws, "gn_bwd_partial: null pointer");
  VAE_CHECK(aligned16(x) && aligned16(g) && aligned16(gamma) && aligned16(beta), "gn_bwd_partial: unaligned");
#define GNP(S, BF, XB) hipLaunchKernelGGL((gn_bwd_partial_kernel<S, BF, XB>), dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, g, mean, rstd, gamma, beta, HW, C, G, nchunk, ws)
#define GNP2(S, BF) do { if (x_bf16) GNP(S, BF, true); else GNP(S, BF, false); } while (0)
  if (silu) { if (g_bf16) GNP2(true, true); else GNP2(true, false); }
  else { if (g_bf16) GNP2(false, true); else GNP2(false, false); }
#undef GNP2
#undef GNP
  VAE_LAUNCH_CHECK("gn_bwd_partial");
  return VAE_OK;
}

extern "C" int vae_gn_bwd_final(const float* ws, const float* rstd, const float* gamma, int32_t B, int32_t HW, int32_t C,
                                int32_t G, int32_t nchunk, float* dgamma, float* dbeta, float* coef, void* stream) {
  if (int e = check_gn("gn_bwd_final", B, HW, C, G, nchunk)) return e;
  VAE_CHECK(ws && rstd && gamma && dgamma && dbeta && coef, "gn_bwd_final: null pointer");
  VAE_CHECK(G % GN_GPW == 0 && GN_GPW * (C / G) <= 64 && 256 % (GN_GPW * (C / G)) == 0, "gn_bwd_final: %d channels per group unsupported (G=%d)", C / G, G);
  hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(B * (G / GN_GPW) + (C + 7) / 8), dim3(256), 0, (hipStream_t)stream, ws, rstd, gamma, B, HW, C, G,
                     nchunk, coef, dgamma, dbeta);
  VAE_LAUNCH_CHECK("gn_bwd_final");
  return VAE_OK;
}

extern "C" int vae_gn_bwd_apply(const void* x, int32_t x_bf16, const void* g, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, const float* coef, const void* add, int32_t B,
                                int32_t HW, int32_t C, int32_t G, int32_t silu, int32_t g_bf16, float* dx, void* dx16, void* stream) {
  if (int e = check_gn("gn_bwd_apply", B, HW, C, G, 1)) return e;
  VAE_CHECK(x && g && mean && rstd && gamma && beta && coef && (dx || dx16), "gn_bwd_apply: null pointer");
  VAE_CHECK(aligned16(x) && aligned16(g) && aligned16(dx) && aligned16(dx16) && aligned16(gamma) && aligned16(beta) &&
                (add == nullptr || aligned16(add)),
            "gn_bwd_apply: unaligned");
  const int Q = C / 4;
  const int64_t n4 = (int64_t)B * HW * Q;
  const RowPlan pl = row_plan(B, HW, Q);  // (check_gn: C / 4 divides 256)
  (void)n4;
#define GNA(S, BF, XB) hipLaunchKernelGGL((gn_bwd_apply_rows_kernel<S, BF, XB>), dim3(pl.nchunk, B), dim3(256), 0, (hipStream_t)stream, x, g, mean, rstd, gamma, beta, coef, add, HW, C, G, pl.per, dx, (unsigned short*)dx16)
#define GNA2(S, BF) do { if (x_bf16) GNA(S, BF, true); else GNA(S, BF, false); } while (0)
  if (silu) { if (g_bf16) GNA2(true, true); else GNA2(true, false); }
  else { if (g_bf16) GNA2(false, true); else GNA2(false, false); }
#undef GNA2
#undef GNA
  VAE_LAUNCH_CHECK("gn_bwd_apply");
  return VAE_OK;
}
