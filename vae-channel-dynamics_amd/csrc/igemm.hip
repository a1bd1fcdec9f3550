// Implicit-GEMM convolution / GEMM family on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
// Three operand forms cover every contraction of the SDXL-VAE train step:
//   rows   (A rows = pixels gathered through the conv geometry)
//     - weight tile k-contiguous : conv/linear forward, Q.K^T, dO.V^T
//     - weight tile n-contiguous : conv/linear dgrad, P.V, dS.K
//   wgrad  (contraction over pixels; both tiles k-major) : conv/linear wgrad, P^T.dO, dS^T.Q
//
// Tiling: 256 threads = 4 waves, BK = 32, wave tile = (BM/WM) x (BN/WN) built from
// 32x32 MFMA tiles.  LDS tiles are padded so every ds_read_b128 fragment read is
// bank-conflict free (row stride 36 dwords: 36*i mod 64 hits 16 distinct 16-B slots
// for the 16 rows of a b128 lane group).  Global loads of step s+1 are issued before
// the MFMA block of step s and written to LDS after it (register-staged prefetch);
// GroupNorm+SiLU is applied to the A operand in that write pass, so the normalised
// activation never exists in HBM.
#include "common.h"

namespace {

constexpr int BK = 32;

template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* p, int c, int C) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (c < C) v = *reinterpret_cast<const f32x4*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) v[e] = p[e];
  }
  return v;
}

template <bool VEC, int XF>
__device__ __forceinline__ f32x4 xform4(f32x4 v, const float* scale, const float* shift, int c, int C) {
  if (XF == VAE_XF_NONE) return v;
  f32x4 sc = load4<VEC>(scale, c, C);
  f32x4 sh = load4<VEC>(shift, c, C);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float u = v[e] * sc[e] + sh[e];
    if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
    v[e] = (c + e < C) ? u : 0.f;
  }
  return v;
}

// ---------------------------------------------------------------------------------------
// rows kernel
// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool BKM, bool VEC, int XF>
__global__ __launch_bounds__(256) void igemm_rows_kernel(vae_igemm_args p) {
  constexpr int LDA = BK + 4;
  constexpr int LDB = BKM ? (BN + 4) : (BK + 4);
  constexpr int SA = BM * LDA;
  constexpr int SB = BKM ? BK * LDB : BN * LDB;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AR = BM / 32;                       // A rows per thread
  constexpr int BR = BKM ? (BK / (256 / (BN / 4))) : (BN / 32);
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float smem[SA + SB];
  float* sA = smem;
  float* sB = smem + SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const float* __restrict__ A = p.A + (int64_t)z * p.sAb;
  const float* __restrict__ W = p.W + (int64_t)z * p.sWb;

  // per-thread A rows
  const int k4 = tid & 7, r0 = tid >> 3;
  int rb[AR], ry[AR], rx[AR];
  {
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int m = m0 + r0 + 32 * i;
      if (m < p.M) {
        int b = m / hw, rem = m - b * hw;
        int y = rem / g.Wo;
        rb[i] = b; ry[i] = y; rx[i] = rem - y * g.Wo;
      } else {
        rb[i] = -1; ry[i] = 0; rx[i] = 0;
      }
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int kchunks = (p.K + BK - 1) / BK;
  const int steps = g.taps * kchunks;

  f32x4 ra[AR], rbw[BR];
  int a_b[AR];  // batch index of the loaded row (for scale/shift), -1 = padding
  int cur_c0 = 0;

  auto load_regs = [&](int s) {
    const int tap = s / kchunks;
    const int c0 = (s - tap * kchunks) * BK;
    cur_c0 = c0;
    const int kh = (g.taps == 9) ? tap / 3 : 0, kw = (g.taps == 9) ? tap - kh * 3 : 0;
    const int c = c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      a_b[i] = -1;
      if (rb[i] >= 0) {
        int sy, sx;
        if (src_pixel(g, ry[i], rx[i], kh, kw, sy, sx)) {
          const float* src = A + (((int64_t)rb[i] * g.Hs + sy) * g.Ws + sx) * g.Cs + c;
          v = load4<VEC>(src, c, p.K);
          a_b[i] = rb[i];
        }
      }
      ra[i] = v;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int n = n0 + r0 + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) v = load4<VEC>(W + (int64_t)n * p.sn + (int64_t)tap * p.st + c, c, p.K);
        rbw[i] = v;
      }
    } else {
      constexpr int NQ = BN / 4, KR = 256 / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int k = c0 + kq + KR * i;
        int n = n0 + n4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < p.K) v = load4<VEC>(W + (int64_t)k * p.sk + (int64_t)tap * p.st + n, n, p.N);
        rbw[i] = v;
      }
    }
  };

  auto store_lds = [&]() {
    const int c = cur_c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = ra[i];
      if (XF != VAE_XF_NONE) {
        if (a_b[i] >= 0) {
          int64_t o = (int64_t)a_b[i] * g.Cs + c;
          v = xform4<VEC, XF>(v, p.scale + o, p.shift + o, c, p.K);
        }
      }
      *reinterpret_cast<f32x4*>(&sA[(r0 + 32 * i) * LDA + k4 * 4]) = v;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(r0 + 32 * i) * LDB + k4 * 4]) = rbw[i];
    } else {
      constexpr int NQ = BN / 4, KR = 256 / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(kq + KR * i) * LDB + n4 * 4]) = rbw[i];
    }
  };

  load_regs(0);
  const int lr = lane & 31, lh = lane >> 5;
  for (int s = 0; s < steps; ++s) {
    __syncthreads();
    store_lds();
    __syncthreads();
    if (s + 1 < steps) load_regs(s + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a[mi] = *reinterpret_cast<const f32x4*>(&sA[(wm * TM + mi * 32 + lr) * LDA + kk * 8 + lh * 4]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!BKM) {
          b[ni] = *reinterpret_cast<const f32x4*>(&sB[(wn * TN + ni * 32 + lr) * LDB + kk * 8 + lh * 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[ni][j] = sB[(kk * 8 + lh * 4 + j) * LDB + wn * TN + ni * 32 + lr];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---------------- epilogue ----------------
  float* __restrict__ C = p.C + (int64_t)z * p.sCb;
  const float* __restrict__ R = p.res ? p.res + (int64_t)z * p.sCb : nullptr;
  float tsum[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) tsum[ni] = 0.f;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * TN + ni * 32 + lr;
    const bool colok = col < p.N;
    const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (colok && row < p.M) {
          float v = p.alpha * acc[mi][ni][r] + bv;
          const int64_t o = (int64_t)row * p.ldc + col;
          if (R) v += R[o];
          C[o] = v;
          tsum[ni] += fabsf(v);
        }
      }
    }
  }
  if (p.track && z == 0) {
    float* red = smem;  // [WM][BN]
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s2 = tsum[ni] + __shfl_xor(tsum[ni], 32, 64);
      if (lh == 0) red[wm * BN + wn * TN + ni * 32 + lr] = s2;
    }
    __syncthreads();
    if (tid < BN) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) t += red[w * BN + tid];
      if (n0 + tid < p.N) p.track[(int64_t)tm * p.N + n0 + tid] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------
// wgrad kernel: out[m][tap][n] = sum_pix dY[pix][m] * XF(X[src(pix,tap)][n])
// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VEC, int XF>
__global__ __launch_bounds__(256) void wgrad_kernel(vae_wgrad_args p) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int SA = BK * LDA, SB = BK * LDB;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AQ = BM / 4, AKR = 256 / AQ, AI = BK / AKR;  // dY tile: AQ float4 per row
  constexpr int BQ = BN / 4, BKR = 256 / BQ, BI = BK / BKR;
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float smem[SA + SB];
  float* sA = smem;
  float* sB = smem + SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tap = blockIdx.y / p.nsplit, split = blockIdx.y % p.nsplit;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const int kh = (g.taps == 9) ? tap / 3 : 0, kw = (g.taps == 9) ? tap - kh * 3 : 0;
  const float* __restrict__ dY = p.dY + (int64_t)z * p.sYb;
  const float* __restrict__ X = p.X + (int64_t)z * p.sXb;

  int chunk = (p.npix + p.nsplit - 1) / p.nsplit;
  chunk = ((chunk + BK - 1) / BK) * BK;
  const int pbeg = split * chunk;
  const int pend = min(p.npix, pbeg + chunk);
  const int steps = (pend > pbeg) ? (pend - pbeg + BK - 1) / BK : 0;
  const int hw = g.Ho * g.Wo;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int a4 = tid % AQ, akq = tid / AQ;
  const int b4 = tid % BQ, bkq = tid / BQ;
  f32x4 ra[AI], rx[BI];
  int xb[BI];

  auto load_regs = [&](int s) {
    const int pb = pbeg + s * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int pix = pb + akq + AKR * i;
      int c = m0 + a4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < pend) v = load4<VEC>(dY + (int64_t)pix * p.ldy + c, c, p.M);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int pix = pb + bkq + BKR * i;
      int c = n0 + b4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      xb[i] = -1;
      if (pix < pend) {
        int b = pix / hw, rem = pix - b * hw;
        int y = rem / g.Wo, x = rem - y * g.Wo;
        int sy, sx;
        if (src_pixel(g, y, x, kh, kw, sy, sx)) {
          v = load4<VEC>(X + (((int64_t)b * g.Hs + sy) * g.Ws + sx) * g.Cs + c, c, p.N);
          xb[i] = b;
        }
      }
      rx[i] = v;
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&sA[(akq + AKR * i) * LDA + a4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      f32x4 v = rx[i];
      if (XF != VAE_XF_NONE) {
        if (xb[i] >= 0) {
          int c = n0 + b4 * 4;
          int64_t o = (int64_t)xb[i] * g.Cs + c;
          v = xform4<VEC, XF>(v, p.scale + o, p.shift + o, c, p.N);
        }
      }
      *reinterpret_cast<f32x4*>(&sB[(bkq + BKR * i) * LDB + b4 * 4]) = v;
    }
  };

  const int lr = lane & 31, lh = lane >> 5;
  if (steps > 0) load_regs(0);
  for (int s = 0; s < steps; ++s) {
    __syncthreads();
    store_lds();
    __syncthreads();
    if (s + 1 < steps) load_regs(s + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kk * 8 + lh * 4 + j;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi][j] = sA[k * LDA + wm * TM + mi * 32 + lr];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni][j] = sB[k * LDB + wn * TN + ni * 32 + lr];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    }
  }

  const int64_t ld = (int64_t)g.taps * p.N;
  float* __restrict__ O = (p.nsplit == 1 ? p.out : p.partial + (int64_t)split * p.M * ld) + (int64_t)z * p.sOb;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * TN + ni * 32 + lr;
    if (col >= p.N) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M) O[(int64_t)row * ld + (int64_t)tap * p.N + col] = p.alpha * acc[mi][ni][r];
      }
  }
}

__global__ void reduce_splits_kernel(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * n + i];
    out[i] = s;
  }
}

template <int BM, int BN, int WM, int WN, bool BKM, bool VEC>
int launch_rows_xf(const vae_igemm_args& a, dim3 grid, hipStream_t st) {
  if (BKM) {
    if (a.xf != VAE_XF_NONE) { vae_set_error("igemm_rows: xf unsupported with n-contiguous weights"); return VAE_EINVAL; }
    hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, BKM, VEC, VAE_XF_NONE>), grid, dim3(256), 0, st, a);
    return 0;
  }
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_NONE>), grid, dim3(256), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_AFFINE>), grid, dim3(256), 0, st, a); break;
    case VAE_XF_AFFINE_SILU: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_AFFINE_SILU>), grid, dim3(256), 0, st, a); break;
    default: vae_set_error("igemm_rows: bad xf %d", a.xf); return VAE_EINVAL;
  }
  return 0;
}

template <int BM, int BN, int WM, int WN>
int launch_rows(const vae_igemm_args& a, bool bkm, bool vec, hipStream_t st) {
  dim3 grid((unsigned)(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN)), 1, (unsigned)a.batch);
  if (bkm) return vec ? launch_rows_xf<BM, BN, WM, WN, true, true>(a, grid, st) : launch_rows_xf<BM, BN, WM, WN, true, false>(a, grid, st);
  return vec ? launch_rows_xf<BM, BN, WM, WN, false, true>(a, grid, st) : launch_rows_xf<BM, BN, WM, WN, false, false>(a, grid, st);
}

template <int BM, int BN, int WM, int WN, bool VEC>
int launch_wgrad_xf(const vae_wgrad_args& a, dim3 grid, hipStream_t st) {
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_NONE>), grid, dim3(256), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_AFFINE>), grid, dim3(256), 0, st, a); break;
    case VAE_XF_AFFINE_SILU: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_AFFINE_SILU>), grid, dim3(256), 0, st, a); break;
    default: vae_set_error("wgrad: bad xf %d", a.xf); return VAE_EINVAL;
  }
  return 0;
}
template <int BM, int BN, int WM, int WN>
int launch_wgrad(const vae_wgrad_args& a, bool vec, hipStream_t st) {
  dim3 grid((unsigned)(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN)), (unsigned)(a.g.taps * a.nsplit), (unsigned)a.batch);
  return vec ? launch_wgrad_xf<BM, BN, WM, WN, true>(a, grid, st) : launch_wgrad_xf<BM, BN, WM, WN, false>(a, grid, st);
}

int check_geom(const char* who, const vae_conv_geom& g) {
  VAE_CHECK(g.B > 0 && g.Hs > 0 && g.Ws > 0 && g.Cs > 0 && g.Ho > 0 && g.Wo > 0, "%s: non-positive geometry", who);
  VAE_CHECK(g.taps == 1 || g.taps == 9, "%s: taps must be 1 or 9 (got %d)", who, g.taps);
  VAE_CHECK(g.stride == 1 || g.stride == 2, "%s: stride must be 1 or 2", who);
  VAE_CHECK(g.mode >= 0 && g.mode <= 2, "%s: bad mode", who);
  VAE_CHECK(g.mode != VAE_MODE_UP2X || (g.taps == 9 && g.stride == 1), "%s: up2x needs 3x3 stride 1", who);
  return 0;
}

}  // namespace

extern "C" int vae_igemm_rows(const vae_igemm_args* ap, void* stream) {
  VAE_CHECK(ap != nullptr, "igemm_rows: null args");
  const vae_igemm_args& a = *ap;
  if (int e = check_geom("igemm_rows", a.g)) return e;
  VAE_CHECK(a.A && a.W && a.C, "igemm_rows: null operand");
  VAE_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "igemm_rows: bad sizes M=%d N=%d K=%d", a.M, a.N, a.K);
  VAE_CHECK(a.K <= a.g.Cs, "igemm_rows: K=%d exceeds source channels %d", a.K, a.g.Cs);
  VAE_CHECK((int64_t)a.g.B * a.g.Ho * a.g.Wo == a.M, "igemm_rows: M=%d != B*Ho*Wo", a.M);
  VAE_CHECK(a.ldc >= a.N, "igemm_rows: ldc < N");
  VAE_CHECK(a.sk == 1 || a.sn == 1, "igemm_rows: one of sn, sk must be 1 (sn=%lld sk=%lld)", (long long)a.sn,
            (long long)a.sk);
  VAE_CHECK(a.xf == VAE_XF_NONE || (a.scale && a.shift), "igemm_rows: xf needs scale/shift");
  const bool bkm = (a.sn == 1) && (a.sk != 1);
  bool vec = aligned16(a.A) && aligned16(a.W) && (a.g.Cs % 4 == 0) && (a.K % 4 == 0) && (a.st % 4 == 0) &&
             (a.sAb % 4 == 0) && (a.sWb % 4 == 0);
  if (bkm) vec = vec && (a.sk % 4 == 0) && (a.N % 4 == 0);
  else vec = vec && (a.sn % 4 == 0);
  if (a.xf != VAE_XF_NONE) vec = vec && aligned16(a.scale) && aligned16(a.shift);
  hipStream_t st = (hipStream_t)stream;
  int rc = (a.N <= 32) ? launch_rows<128, 32, 4, 1>(a, bkm, vec, st) : launch_rows<128, 128, 2, 2>(a, bkm, vec, st);
  if (rc) return rc;
  VAE_LAUNCH_CHECK("igemm_rows");
  return VAE_OK;
}

extern "C" int vae_wgrad(const vae_wgrad_args* ap, void* stream) {
  VAE_CHECK(ap != nullptr, "wgrad: null args");
  const vae_wgrad_args& a = *ap;
  if (int e = check_geom("wgrad", a.g)) return e;
  VAE_CHECK(a.dY && a.X, "wgrad: null operand");
  VAE_CHECK(a.M > 0 && a.N > 0 && a.npix > 0 && a.nsplit > 0 && a.batch > 0, "wgrad: bad sizes");
  VAE_CHECK(a.N <= a.g.Cs, "wgrad: N exceeds source channels");
  VAE_CHECK((int64_t)a.g.B * a.g.Ho * a.g.Wo == a.npix, "wgrad: npix != B*Ho*Wo");
  VAE_CHECK(a.ldy >= a.M, "wgrad: ldy < M");
  VAE_CHECK(a.g.mode != VAE_MODE_DGRAD, "wgrad: dgrad geometry not valid here");
  VAE_CHECK(a.nsplit == 1 ? a.out != nullptr : a.partial != nullptr, "wgrad: missing output buffer");
  VAE_CHECK(a.xf == VAE_XF_NONE || (a.scale && a.shift), "wgrad: xf needs scale/shift");
  bool vec = aligned16(a.dY) && aligned16(a.X) && (a.g.Cs % 4 == 0) && (a.ldy % 4 == 0) && (a.M % 4 == 0) &&
             (a.N % 4 == 0) && (a.sYb % 4 == 0) && (a.sXb % 4 == 0);
  if (a.xf != VAE_XF_NONE) vec = vec && aligned16(a.scale) && aligned16(a.shift);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (a.M <= 32) rc = launch_wgrad<32, 128, 1, 4>(a, vec, st);
  else if (a.N <= 32) rc = launch_wgrad<128, 32, 4, 1>(a, vec, st);
  else rc = launch_wgrad<128, 128, 2, 2>(a, vec, st);
  if (rc) return rc;
  VAE_LAUNCH_CHECK("wgrad");
  return VAE_OK;
}

extern "C" int vae_reduce_splits(const float* partial, int32_t nsplit, int64_t n, float* out, void* stream) {
  VAE_CHECK(partial && out && nsplit > 0 && n > 0, "reduce_splits: bad args");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, nsplit, n, out);
  VAE_LAUNCH_CHECK("reduce_splits");
  return VAE_OK;
}
