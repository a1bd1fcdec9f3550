// Implicit-GEMM convolution / GEMM family on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
// Three operand forms cover every contraction of the SDXL-VAE train step:
//   rows   (A rows = pixels gathered through the conv geometry)
//     - weight tile k-contiguous : conv/linear forward, Q.K^T, dO.V^T
//     - weight tile n-contiguous : conv/linear dgrad, P.V, dS.K
//   wgrad  (contraction over pixels; both tiles k-major) : conv/linear wgrad, P^T.dO, dS^T.Q
//
// Tiling: 64*WM*WN threads (8 waves for the 128x128 tile, 4 for the skinny ones), BK = 32, wave tile =
// (BM/WM) x (BN/WN) built from 32x32 MFMA tiles.  LDS tiles are padded so every ds_read_b128 fragment read is
// bank-conflict free (row stride 36 dwords: 36*i mod 64 hits 16 distinct 16-B slots
// for the 16 rows of a b128 lane group).  Global loads of step s+1 are issued before
// the MFMA block of step s and written to LDS after it (register-staged prefetch);
// GroupNorm+SiLU is applied to the A operand in that write pass, so the normalised
// activation never exists in HBM.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

bool conv3_tile_eligible(const vae_igemm_args& a, bool vec, bool bkm);
int launch_conv3_tile(const vae_igemm_args& a, bool bkm, hipStream_t st);
int launch_conv3_tile_bf16(const vae_igemm_args& a, bool bkm, hipStream_t st);
bool conv3_tile_bf16_packed(const vae_igemm_args& a);
bool conv3_wino_eligible(const vae_igemm_args& a);                      // conv3_wino.hip (fp32 Winograd F(2x2,3x3))
int launch_wino_weights(const vae_igemm_args& a, float* U, hipStream_t st);
int launch_conv3_wino(const vae_igemm_args& a, const float* U, hipStream_t st);
int conv3_wino_gstat_chunks(const vae_igemm_args& a);
int conv3_wino_gnb_chunks(const vae_igemm_args& a);
bool conv1_bf16_eligible(const vae_igemm_args& a);                      // conv1_bf16.hip (bf16 1x1 convolutions, weights resident in LDS)
int launch_conv1_bf16(const vae_igemm_args& a, hipStream_t st);
int conv3_tile_bf16_gstat_chunks(const vae_igemm_args& a);
bool conv3_upwino_eligible(const vae_igemm_args& a);                    // conv3_upwino.hip (fp32 upsampler convolution, 9 positions)
int launch_upwino_weights(const vae_igemm_args& a, float* U, hipStream_t st);
int launch_conv3_upwino(const vae_igemm_args& a, const float* U, hipStream_t st);
int conv3_wino_nb();
bool conv_thin_bf16_eligible(const vae_igemm_args& a);                  // conv_thin_bf16.hip (bf16: <= 4-channel contraction on the matrix pipe)
int launch_conv_thin_bf16(const vae_igemm_args& a, hipStream_t st);
bool conv_thinn_bf16_eligible(const vae_igemm_args& a);                 // (<= 4 output channels)
int launch_conv_thinn_bf16(const vae_igemm_args& a, hipStream_t st);
bool conv3_wino4_eligible(const vae_igemm_args& a);                     // conv3_wino4.hip (fp32 Winograd F(4x4,3x3))
int launch_wino4_weights(const vae_igemm_args& a, float* U, hipStream_t st);
int launch_conv3_wino4(const vae_igemm_args& a, const float* U, hipStream_t st);
int conv3_wino4_gstat_chunks(const vae_igemm_args& a);
int conv3_wino4_gnb_chunks(const vae_igemm_args& a);
bool conv3_wide_bf16_eligible(const vae_igemm_args& a);                 // conv3_wide_bf16.hip (both operands bf16 images, 8x32 tiles)
int conv3_wide_bf16_gstat_chunks(const vae_igemm_args& a);
int launch_conv3_wide_bf16(const vae_igemm_args& a, hipStream_t st);
int conv3_tile_gstat_chunks(const vae_igemm_args& a);  // both tile kernels share the tile shape and the epilogue layout
bool wgrad3_tile_eligible(const vae_wgrad_args& a, bool vec);
int64_t wgrad3_tile_units(const vae_conv_geom& g);
int launch_wgrad3_tile(const vae_wgrad_args& a, hipStream_t st);
bool wgrad3_wino_eligible(const vae_wgrad_args& a);
int64_t wgrad3_wino_units(const vae_conv_geom& g);
int launch_wgrad3_wino(const vae_wgrad_args& a, hipStream_t st);
int launch_wino_wgrad_reduce(const float* slab, int nsplit, int N, int M, float* dW, const float* bpart, float* db, hipStream_t st);
bool wgrad3_upwino_eligible(const vae_wgrad_args& a);                   // wgrad3_upwino.hip (fp32 upsampler convolution, 9 positions)
int64_t wgrad3_upwino_units(const vae_conv_geom& g);
int launch_wgrad3_upwino(const vae_wgrad_args& a, hipStream_t st);
int launch_upwino_wgrad_reduce(const float* slab, int nsplit, int N, int M, float* dW, const float* bpart, float* db, hipStream_t st);
bool wgrad3_tile_bf16_eligible(const vae_wgrad_args& a, bool vec);
int64_t wgrad3_tile_bf16_units(const vae_conv_geom& g);
int wgrad3_tile_bf16_columns(const vae_wgrad_args& a);
int launch_wgrad3_tile_bf16(const vae_wgrad_args& a, hipStream_t st);
bool wgrad3_tile_bf16_dma(const vae_wgrad_args& a);
bool conv_smallk_eligible(const vae_igemm_args& a);                      // skinny.hip (<= 4-channel sides on the VALU)
int launch_conv_smallk(const vae_igemm_args& a, hipStream_t st);
bool conv_smalln_eligible(const vae_igemm_args& a);
int launch_conv_smalln(const vae_igemm_args& a, hipStream_t st);
int wgrad_smallk_kind(const vae_wgrad_args& a);
int wgrad_smallk_tiles(const vae_wgrad_args& a);
int launch_wgrad_smallk(const vae_wgrad_args& a, hipStream_t st);
bool wgrad_smallk_on_mfma(const vae_wgrad_args& a);  // bf16 mode: the launch runs on wgrad_thin_bf16.hip instead
int launch_rows_bf16(const vae_igemm_args& a, bool bkm, hipStream_t st);   // igemm_bf16.hip (vectorised shapes only)
int launch_wgrad_bf16(const vae_wgrad_args& a, hipStream_t st);

namespace {

constexpr int BK = 32;
__device__ __forceinline__ int b_lo_of(int m0, int hw) { return m0 / hw; }

// ---------------------------------------------------------------------------------------
// rows kernel.  Pipeline: LDS is double buffered; the global loads of K-step s+2 are issued and
// the registers of step s+1 are transformed + written to the other LDS stage in the MIDDLE of
// step s's MFMA block, so one workgroup barrier per K-step suffices and the VALU/LDS-write work
// sits in the shadow of MFMAs already issued.  The GroupNorm scale/shift rows the tile needs are
// staged in LDS once per workgroup (they used to be 8 dependent global loads per thread per step).
// ---------------------------------------------------------------------------------------

template <int BM, int BN, int WM, int WN, bool BKM, bool VEC, int XF>
__global__ __launch_bounds__(64 * WM * WN) void igemm_rows_kernel(vae_igemm_args p) {
  constexpr int NT = 64 * WM * WN;  // 4 waves (skinny tiles) or 8 waves (128x128: 4 waves/SIMD at 2 workgroups/CU)
  constexpr int NL = NT;
  constexpr int RP = NT / 8;        // tile rows covered by one pass of the float4 loaders
  constexpr int LDA = BK + 4;
  constexpr int LDB = BKM ? (BN + 4) : (BK + 4);
  constexpr int SA = BM * LDA;
  constexpr int SB = BKM ? BK * LDB : BN * LDB;
  constexpr int STAGE = SA + SB;
  constexpr int SS = (XF != VAE_XF_NONE) ? 2 * SS_HALF : 0;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AR = BM / RP;                       // A rows per thread
  constexpr int BR = BKM ? (BK / (NL / (BN / 4))) : (BN / RP);
  static_assert(AR >= 1 && BR >= 1 && TM % 32 == 0 && TN % 32 == 0, "tile/wave layout");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + SS];
  float* sS = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lt = tid;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const SrcMap smap = make_srcmap(g);
  const float* __restrict__ A = p.A + (int64_t)z * p.sAb;
  const float* __restrict__ W = p.W + (int64_t)z * p.sWb;
  const int hw = g.Ho * g.Wo;

  // stride-2 dgrad, parity-class-major rows: the tile's class fixes the taps it meets
  const bool s2c = (g.mode == VAE_MODE_DGRAD_S2);
  const int hh = g.Ho >> 1, wh = g.Wo >> 1;
  const int cls_rows = s2c ? p.M >> 2 : 1;
  const int cls = s2c ? m0 / cls_rows : 0;
  const int cpy = cls >> 1, cpx = cls & 1;
  const int nkw = s2c ? (cpx ? 1 : 2) : 3;
  const int ntaps = s2c ? (cpy ? 1 : 2) * nkw : g.taps;
  auto row_pixel = [&](int m, int& b, int& y, int& x) {  // GEMM row -> (image, y, x) of the row grid
    if (s2c) {
      const int r = m - cls * cls_rows;
      b = r / (hh * wh);
      const int rem = r - b * (hh * wh);
      const int i = rem / wh;
      y = 2 * i + cpy;
      x = 2 * (rem - i * wh) + cpx;
    } else {
      b = m / hw;
      const int rem = m - b * hw;
      y = rem / g.Wo;
      x = rem - y * g.Wo;
    }
  };

  // vectorised instantiations read both operands through buffer descriptors (common.h): out-of-range offsets read
  // zeros, so no select sits on a loaded value.  The activation descriptor starts at the first image this tile's
  // rows touch (the host checked that the images one tile can span fit 32-bit offsets).
  const int b_base = s2c ? (m0 - cls * cls_rows) / (hh * wh) : b_lo_of(m0, hw);
  const size_t img = (size_t)g.Hs * g.Ws * g.Cs;
  const size_t abytes = (size_t)(g.B - b_base) * img * 4u, wbytes = (size_t)(BKM ? (int64_t)p.K * p.sk : (int64_t)p.N * p.sn) * 4u;
  const auto rsA = VAE_BUF_RSRC(A + (int64_t)b_base * img, abytes < BUF_MAX ? abytes : BUF_MAX);
  const auto rsW = VAE_BUF_RSRC(W, wbytes < BUF_MAX ? wbytes : BUF_MAX);

  // per-thread A rows
  const int k4 = lt & 7, r0 = lt >> 3;
  int rb[AR], ry[AR], rx[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + r0 + RP * i;
    if (m < p.M) {
      row_pixel(m, rb[i], ry[i], rx[i]);
    } else {
      rb[i] = -1; ry[i] = 0; rx[i] = 0;
    }
  }

  // scale/shift table for the batches this tile touches
  // (the host checks with vae_xf_fusable_rows that the rows of one tile never need more than SS_HALF entries)
  const int b_lo = m0 / hw;
  if (XF != VAE_XF_NONE) {
    const int b_hi = (min(p.M, m0 + BM) - 1) / hw;
    const int nent = min((b_hi - b_lo + 1) * p.K, SS_HALF);
    for (int i = tid; i < nent; i += NT) {
      const int j = i / p.K, c = i - j * p.K;
      sS[i] = p.scale[(int64_t)(b_lo + j) * g.Cs + c];
      sS[SS_HALF + i] = p.shift[(int64_t)(b_lo + j) * g.Cs + c];
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
  const int steps = ntaps * kchunks;

  f32x4 ra[AR], rbw[BR];
  int a_b[AR];  // batch index of the loaded row (for scale/shift), -1 = padding
  int reg_c0 = 0;

  auto load_regs = [&](int s) {
    const int ord = s / kchunks;  // ordinal of the tap among the taps this tile meets
    const int c0 = (s - ord * kchunks) * BK;
    reg_c0 = c0;
    int kh, kw;
    if (s2c) {
      const int a = ord / nkw;
      kh = cpy ? 1 : 2 * a;
      kw = cpx ? 1 : 2 * (ord - a * nkw);
    } else {
      kh = (g.taps == 9) ? ord / 3 : 0;
      kw = (g.taps == 9) ? ord - kh * 3 : 0;
    }
    const int tap = kh * 3 + kw;
    const int c = c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int sy = 0, sx = 0;
      const bool ok = src_pixel(smap, ry[i], rx[i], kh, kw, sy, sx) && (rb[i] >= 0);
      if (VEC) {
        ra[i] = VAE_BUF_LOAD4(rsA, oob_unless(ok && c < p.K, ((unsigned)(((rb[i] - b_base) * g.Hs + sy) * g.Ws + sx) * (unsigned)g.Cs + (unsigned)c) * 4u));
      } else {
        ra[i] = load4s(A + (((int64_t)rb[i] * g.Hs + sy) * g.Ws + sx) * g.Cs + c, ok, c, p.K);
      }
      a_b[i] = ok ? rb[i] : -1;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + RP * i;
        if (VEC) rbw[i] = VAE_BUF_LOAD4(rsW, oob_unless(n < p.N && c < p.K, ((unsigned)n * (unsigned)p.sn + (unsigned)tap * (unsigned)p.st + (unsigned)c) * 4u));
        else rbw[i] = load4s(W + (int64_t)n * p.sn + (int64_t)tap * p.st + c, n < p.N, c, p.K);
      }
    } else {
      constexpr int NQ = BN / 4, KR = NL / NQ;
      const int n4 = lt % NQ, kq = lt / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int k = c0 + kq + KR * i;
        const int n = n0 + n4 * 4;
        if (VEC) rbw[i] = VAE_BUF_LOAD4(rsW, oob_unless(k < p.K && n < p.N, ((unsigned)k * (unsigned)p.sk + (unsigned)tap * (unsigned)p.st + (unsigned)n) * 4u));
        else rbw[i] = load4s(W + (int64_t)k * p.sk + (int64_t)tap * p.st + n, k < p.K, n, p.N);
      }
    }
  };

  auto store_lds = [&](float* sA, float* sB) {
    const int c = reg_c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = ra[i];
      if (XF != VAE_XF_NONE) {
        const bool ok = (a_b[i] >= 0) && (c < p.K);
        const int o = ok ? (a_b[i] - b_lo) * p.K + c : 0;
        v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
      }
      *reinterpret_cast<f32x4*>(&sA[(r0 + RP * i) * LDA + k4 * 4]) = v;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(r0 + RP * i) * LDB + k4 * 4]) = rbw[i];
    } else {
      constexpr int NQ = BN / 4, KR = NL / NQ;
      const int n4 = lt % NQ, kq = lt / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(kq + KR * i) * LDB + n4 * 4]) = rbw[i];
    }
  };

  const int lr = lane & 31, lh = lane >> 5;
  // fragments of k-group kk+1 are requested before the MFMAs of kk are issued (pinned with sched_barrier)
  f32x4 fa[2][MI], fb[2][NI];
  auto fetch = [&](const float* sA, const float* sB, int kk, f32x4* a, f32x4* b) {
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
  };
  auto compute = [&](const float* sA, const float* sB, int kk) {  // fragments of kk already requested
    if (kk + 1 < BK / 8) fetch(sA, sB, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][mi][j], fb[kk & 1][ni][j], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  load_regs(0);
  __syncthreads();  // scale/shift table visible
  store_lds(smem, smem + SA);
  if (steps > 1) load_regs(1);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const float* cA = smem + (s & 1) * STAGE;
    const float* cB = cA + SA;
    fetch(cA, cB, 0, fa[0], fb[0]);
    compute(cA, cB, 0);
    compute(cA, cB, 1);
    if (s + 1 < steps) {  // staged in the shadow of the MFMAs already issued
      float* nA = smem + ((s + 1) & 1) * STAGE;
      store_lds(nA, nA + SA);
      if (s + 2 < steps) load_regs(s + 2);
    }
    compute(cA, cB, 2);
    compute(cA, cB, 3);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // outputs and the residual through buffer descriptors (common.h): a row / column outside the matrix is an
  // out-of-range offset (load reads 0, store is dropped): no branch per element, residual loads issued back to back
  float* __restrict__ C = p.C + (int64_t)z * p.sCb;
  const size_t obytes = (size_t)p.M * p.ldc * 4u;
  const auto rsC = VAE_BUF_RSRC(C, obytes);
  const auto rsR = VAE_BUF_RSRC(p.res ? p.res + (int64_t)z * p.sCb : C, obytes);
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
      unsigned off[16];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        unsigned orow = (unsigned)row;
        if (s2c) {  // class-major row -> pixel-major output row
          int b, y, x;
          row_pixel(row < p.M ? row : m0, b, y, x);
          orow = (unsigned)((b * g.Ho + y) * g.Wo + x);
        }
        off[r] = (colok && row < p.M) ? (orow * (unsigned)p.ldc + (unsigned)col) * 4u : BUF_OOB;
        rv[r] = 0.f;
      }
      if (p.res) {  // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off[r], 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = p.alpha * acc[mi][ni][r] + bv + rv[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off[r], 0, 0);
        tsum[ni] += (off[r] != BUF_OOB) ? fabsf(v) : 0.f;
      }
    }
  }
  if (p.track && z == 0) {
    float* red = smem;  // [WM][BN]; the last loop barrier already separated it from the MFMA reads
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
// Same double-buffered one-barrier pipeline.  The bias gradient (column sums of dY) is folded in:
// workgroups with tn == 0 and tap == 0 add up the dY tiles they stage anyway.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VEC, int XF>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_kernel(vae_wgrad_args p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int NL = NT;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int SA = BK * LDA, SB = BK * LDB;
  constexpr int STAGE = SA + SB;
  constexpr int SS = (XF != VAE_XF_NONE) ? 2 * SS_HALF : 0;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AQ = BM / 4, AKR = NL / AQ, AI = BK / AKR;  // dY tile: AQ float4 per row
  constexpr int BQ = BN / 4, BKR = NL / BQ, BI = BK / BKR;
  static_assert(AI >= 1 && BI >= 1 && TM % 32 == 0 && TN % 32 == 0, "tile/wave layout");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + SS];
  float* sS = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lt = tid;
  const int wm = wave / WN, wn = wave % WN;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tap = blockIdx.y / p.nsplit, split = blockIdx.y % p.nsplit;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const SrcMap smap = make_srcmap(g);
  const int kh = (g.taps == 9) ? tap / 3 : 0, kw = (g.taps == 9) ? tap - kh * 3 : 0;
  const float* __restrict__ dY = p.dY + (int64_t)z * p.sYb;
  const float* __restrict__ X = p.X + (int64_t)z * p.sXb;

  int chunk = (p.npix + p.nsplit - 1) / p.nsplit;
  chunk = ((chunk + BK - 1) / BK) * BK;
  const int pbeg = split * chunk;
  const int pend = min(p.npix, pbeg + chunk);
  const int steps = (pend > pbeg) ? (pend - pbeg + BK - 1) / BK : 0;
  const int hw = g.Ho * g.Wo;
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0 && tap == 0 && z == 0;

  // (the host checks with vae_xf_fusable_wgrad that one split never spans more than SS_HALF/BN batch items)
  const int b_lo = pbeg / hw;
  if (XF != VAE_XF_NONE && steps > 0) {
    const int nb = (pend - 1) / hw - b_lo + 1;
    const int ncol = min(BN, p.N - n0);
    const int nent = min(nb * BN, SS_HALF);
    for (int i = tid; i < nent; i += NT) {
      const int j = i / BN, c = i - j * BN;
      const bool ok = c < ncol;
      sS[i] = ok ? p.scale[(int64_t)(b_lo + j) * g.Cs + n0 + c] : 0.f;
      sS[SS_HALF + i] = ok ? p.shift[(int64_t)(b_lo + j) * g.Cs + n0 + c] : 0.f;
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // buffer descriptors (vectorised instantiations): dY from this split's first pixel, X from its first image
  const size_t img = (size_t)g.Hs * g.Ws * g.Cs;
  const size_t ybytes = (size_t)(steps > 0 ? pend - pbeg : 0) * p.ldy * 4u, xbytes = (size_t)(g.B - b_lo) * img * 4u;
  const auto rsY = VAE_BUF_RSRC(dY + (int64_t)pbeg * p.ldy, ybytes < BUF_MAX ? ybytes : BUF_MAX);
  const auto rsX = VAE_BUF_RSRC(X + (int64_t)b_lo * img, xbytes < BUF_MAX ? xbytes : BUF_MAX);

  const int a4 = lt % AQ, akq = lt / AQ;
  const int b4 = lt % BQ, bkq = lt / BQ;
  f32x4 ra[AI], rx[BI];
  int xb[BI];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  auto load_regs = [&](int s) {
    const int pb = pbeg + s * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int pix = pb + akq + AKR * i;
      const int c = m0 + a4 * 4;
      if (VEC) ra[i] = VAE_BUF_LOAD4(rsY, oob_unless(pix < pend && c < p.M, ((unsigned)(pix - pbeg) * (unsigned)p.ldy + (unsigned)c) * 4u));
      else ra[i] = load4s(dY + (int64_t)pix * p.ldy + c, pix < pend, c, p.M);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int pix = pb + bkq + BKR * i;
      const int c = n0 + b4 * 4;
      const int b = pix / hw, rem = pix - b * hw;
      const int y = rem / g.Wo, x = rem - y * g.Wo;
      int sy = 0, sx = 0;
      const bool ok = src_pixel(smap, y, x, kh, kw, sy, sx) && (pix < pend);
      if (VEC) rx[i] = VAE_BUF_LOAD4(rsX, oob_unless(ok && c < p.N, ((unsigned)(((b - b_lo) * g.Hs + sy) * g.Ws + sx) * (unsigned)g.Cs + (unsigned)c) * 4u));
      else rx[i] = load4s(X + (((int64_t)b * g.Hs + sy) * g.Ws + sx) * g.Cs + c, ok, c, p.N);
      xb[i] = ok ? b : -1;
    }
  };
  auto store_lds = [&](float* sA, float* sB) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      *reinterpret_cast<f32x4*>(&sA[(akq + AKR * i) * LDA + a4 * 4]) = ra[i];
      if (do_bias) bsum += ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      f32x4 v = rx[i];
      if (XF != VAE_XF_NONE) {
        const bool ok = xb[i] >= 0;
        const int o = ok ? (xb[i] - b_lo) * BN + b4 * 4 : 0;
        v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
      }
      *reinterpret_cast<f32x4*>(&sB[(bkq + BKR * i) * LDB + b4 * 4]) = v;
    }
  };

  const int lr = lane & 31, lh = lane >> 5;
  f32x4 fa[2][MI], fb[2][NI];
  auto fetch = [&](const float* sA, const float* sB, int kk, f32x4* a, f32x4* b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kk * 8 + lh * 4 + j;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi][j] = sA[k * LDA + wm * TM + mi * 32 + lr];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni][j] = sB[k * LDB + wn * TN + ni * 32 + lr];
    }
  };
  auto compute = [&](const float* sA, const float* sB, int kk) {  // fragments of kk already requested
    if (kk + 1 < BK / 8) fetch(sA, sB, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][mi][j], fb[kk & 1][ni][j], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (steps > 0) {
    load_regs(0);
    __syncthreads();  // scale/shift table visible
    store_lds(smem, smem + SA);
    if (steps > 1) load_regs(1);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const float* cA = smem + (s & 1) * STAGE;
      const float* cB = cA + SA;
      fetch(cA, cB, 0, fa[0], fb[0]);
      compute(cA, cB, 0);
      compute(cA, cB, 1);
      if (s + 1 < steps) {
        float* nA = smem + ((s + 1) & 1) * STAGE;
        store_lds(nA, nA + SA);
        if (s + 2 < steps) load_regs(s + 2);
      }
      compute(cA, cB, 2);
      compute(cA, cB, 3);
      __syncthreads();
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
  if (do_bias) {  // uniform per workgroup
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [AKR][AQ]
    red[akq * AQ + a4] = bsum;
    __syncthreads();
    if (tid < AQ) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < AKR; ++r) t += red[r * AQ + tid];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + tid * 4 + e;
        if (m < p.M) p.bias_partial[(int64_t)split * p.M + m] = t[e];
      }
    }
  }
}

// out[i] = sum_k partial[k][i], fixed association (reproducible).  16-byte loads, 8 independent loads in flight per
// thread; KP threads share a column quad and split the k range (a 128-channel layer has 128 splits of only 147 K elements:
// one thread per element leaves too few bytes in flight to fill HBM), combined through LDS in k order.
// A second, small reduction (the bias gradient: [nsplit][n2]) rides in the same launch: workgroups main_blocks.. do it.
template <int KP>
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ out,
                                                            int main_blocks, const float* __restrict__ partial2, int n2, float* __restrict__ out2) {
  constexpr int COLS = 256 / KP;
  __shared__ f32x4 red[KP > 1 ? 256 : 1];
  if ((int)blockIdx.x >= main_blocks) {  // uniform per workgroup
    const int i = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (i < n2) {
      float s2 = 0.f;
      for (int k = 0; k < nsplit; ++k) s2 += partial2[(int64_t)k * n2 + i];
      out2[i] = s2;
    }
    return;
  }
  const int col = threadIdx.x % COLS, kp = threadIdx.x / COLS;
  const int64_t n4 = n >> 2;
  const int per = (nsplit + KP - 1) / KP;
  const int k0 = kp * per, k1 = min(nsplit, k0 + per);
  auto column = [&](int64_t i) {  // sum of splits [k0, k1) of column quad i
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + i;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(k + j) * n4];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; k < k1; ++k) s += p[(int64_t)k * n4];
    return s;
  };
  if (KP == 1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)main_blocks * 256) reinterpret_cast<f32x4*>(out)[i] = column(i);
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * COLS + col;  // one workgroup per COLS column quads (no loop: the barrier below is uniform)
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) s = column(i);
  red[threadIdx.x] = s;
  __syncthreads();
  if (kp == 0 && i < n4) {
#pragma unroll
    for (int j = 1; j < KP; ++j) s += red[j * COLS + col];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}
__global__ void reduce_splits_scalar_kernel(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ out) {
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
    hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, BKM, VEC, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a);
    return 0;
  }
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_AFFINE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE_SILU: hipLaunchKernelGGL((igemm_rows_kernel<BM, BN, WM, WN, false, VEC, VAE_XF_AFFINE_SILU>), grid, dim3(64 * WM * WN), 0, st, a); break;
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
    case VAE_XF_NONE: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_AFFINE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE_SILU: hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, VEC, VAE_XF_AFFINE_SILU>), grid, dim3(64 * WM * WN), 0, st, a); break;
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
  VAE_CHECK(g.mode >= 0 && g.mode <= 4, "%s: bad mode", who);
  VAE_CHECK(g.mode != VAE_MODE_UP2X_DGRAD || (g.taps == 9 && g.stride == 1 && g.Hs == 2 * g.Ho && g.Ws == 2 * g.Wo),
            "%s: UP2X_DGRAD needs 3x3 stride 1, source twice the row grid", who);
  VAE_CHECK(g.mode != VAE_MODE_DGRAD_S2 ||
                (g.taps == 9 && g.stride == 2 && g.pad_t == 0 && g.pad_l == 0 && g.Ho % 2 == 0 && g.Wo % 2 == 0 &&
                 ((int64_t)g.B * g.Ho * g.Wo / 4) % 128 == 0),
            "%s: DGRAD_S2 needs 3x3 stride 2 pad 0, even row grid and B*Ho*Wo/4 %% 128 == 0", who);
  VAE_CHECK(g.mode != VAE_MODE_UP2X || (g.taps == 9 && g.stride == 1), "%s: up2x needs 3x3 stride 1", who);
  return 0;
}

}  // namespace

// rows of one 128-row tile span at most nb batch items; the LDS table holds SS_HALF scale entries
static bool xf_rows_ok(const vae_conv_geom& g, int M, int K) {
  const int hw = g.Ho * g.Wo;
  const int nb = (hw % 128 == 0) ? 1 : (127 / hw + 2);
  return (K % 4 == 0) && ((int64_t)std::min(nb, g.B) * K <= SS_HALF);
}
static bool xf_wgrad_ok(const vae_conv_geom& g, int npix, int nsplit, int N) {
  const int hw = g.Ho * g.Wo;
  int chunk = (npix + nsplit - 1) / nsplit;
  chunk = ((chunk + BK - 1) / BK) * BK;
  const int nb = (hw % chunk == 0) ? 1 : ((chunk - 1) / hw + 2);
  const int bn = 128;  // conservative: the widest N tile any instantiation uses
  return (N % 4 == 0) && ((int64_t)std::min(nb, g.B) * bn <= SS_HALF);
}
extern "C" int vae_xf_fusable_rows(const vae_conv_geom* g, int32_t M, int32_t K) { return g && xf_rows_ok(*g, M, K) ? 1 : 0; }
static bool wgrad_vec(const vae_wgrad_args& a) {
  bool vec = aligned16(a.dY) && aligned16(a.X) && (a.g.Cs % 4 == 0) && (a.ldy % 4 == 0) && (a.M % 4 == 0) &&
             (a.N % 4 == 0) && (a.sYb % 4 == 0) && (a.sXb % 4 == 0);
  if (a.xf != VAE_XF_NONE) vec = vec && aligned16(a.scale) && aligned16(a.shift);
  return vec;
}
static bool wgrad_is_phase(const vae_wgrad_args& a) { return a.tapmask != 0 || a.y_step > 1; }
static bool wgrad_use_tile(const vae_wgrad_args& a) { return wgrad3_tile_eligible(a, wgrad_vec(a)) && !vae_opt().flat_conv; }
static bool wgrad_use_tile_bf16(const vae_wgrad_args& a) {
  return a.prec == VAE_PREC_BF16 && wgrad3_tile_bf16_eligible(a, wgrad_vec(a)) && !vae_opt().flat_conv;
}

// operand images (X16 with xf == NONE, dY16) on a layer the bf16 halo-tile kernel does not serve become "X / dY is stored as
// bf16" for the flat / <= 4-channel kernels (as rows_canon does for A16)
static vae_wgrad_args wgrad_canon(const vae_wgrad_args& a) {
  vae_wgrad_args b = a;
  if ((b.X16 != nullptr || b.dY16 != nullptr) && b.prec == VAE_PREC_BF16 && !wgrad_is_phase(b)) {
    vae_wgrad_args t = b;
    if (t.X16 != nullptr && t.xf == VAE_XF_NONE) t.X = reinterpret_cast<const float*>(t.X16);
    if (t.dY16 != nullptr && t.dY == nullptr) t.dY = reinterpret_cast<const float*>(t.dY16);
    if (!wgrad_use_tile_bf16(t)) {
      if (b.X16 != nullptr && b.xf == VAE_XF_NONE) { b.X = reinterpret_cast<const float*>(b.X16); b.X16 = nullptr; b.x_bf16 = 1; }
      if (b.dY16 != nullptr) { b.dY = reinterpret_cast<const float*>(b.dY16); b.dY16 = nullptr; b.y_bf16 = 1; }
    }
  }
  return b;
}
// split-K plan: which nsplit to use for these arguments (a->nsplit is ignored) and whether a->xf can be fused.
// The caller allocates partial[nsplit][M*taps*N] (+ bias_partial[nsplit][M]) accordingly.
extern "C" int vae_wgrad_plan(const vae_wgrad_args* ap, int32_t* nsplit, int32_t* xf_fusable) {
  VAE_CHECK(ap && nsplit && xf_fusable, "wgrad_plan: null argument");
  const vae_wgrad_args a = wgrad_canon(*ap);
  // a workgroup keeps the GroupNorm scale/shift rows of every batch item its unit range touches in LDS
  // (SS_HALF entries): the split count is raised until that fits
  auto min_split = [&](int64_t units, int ci_tile) -> int64_t {
    if (a.xf == VAE_XF_NONE) return 1;
    const int64_t upi = units / a.g.B;                         // units per image
    const int64_t nb_max = SS_HALF / ci_tile;                  // batch items whose rows fit
    const int64_t per_max = std::max<int64_t>(1, (nb_max - 1) * upi);
    return (units + per_max - 1) / per_max;
  };
  if (!wgrad_is_phase(a) && wgrad_smallk_kind(a)) {  // <= 4-channel side: one slab per workgroup, 128-pixel tiles dealt out in ranges
    *nsplit = (int32_t)std::max(1, std::min(1024, wgrad_smallk_tiles(a)));
    *xf_fusable = 1;
    return VAE_OK;
  }
  if (wgrad_use_tile_bf16(a) && (!wgrad_is_phase(a) || a.prec == VAE_PREC_BF16)) {
    const int64_t units = wgrad3_tile_bf16_units(a.g);
    const int64_t cols = wgrad3_tile_bf16_columns(a);
    int64_t ns = std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(cols, 1), units / 4));
    *nsplit = (int32_t)std::max(ns, min_split(units, 64));
    *xf_fusable = 1;
    return VAE_OK;
  }
  if (wgrad_use_tile(a)) {
    const int64_t units = wgrad3_tile_units(a.g);
    const int64_t wgs = (int64_t)((a.M + 127) / 128) * (a.N / 32);
    int64_t ns = std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(wgs, 1), units / 8));  // one 12-wave workgroup per CU
    *nsplit = (int32_t)std::max(ns, min_split(units, 32));
    *xf_fusable = 1;
    return VAE_OK;
  }
  const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.g.taps;
  const int64_t ns = std::max<int64_t>(1, std::min<int64_t>(512 / std::max<int64_t>(tiles, 1), a.npix / 256));
  *nsplit = (int32_t)ns;
  *xf_fusable = xf_wgrad_ok(a.g, a.npix, (int)ns, a.N) ? 1 : 0;
  return VAE_OK;
}

static bool rows_bkm(const vae_igemm_args& a) { return (a.sn == 1) && (a.sk != 1); }
static bool rows_vec(const vae_igemm_args& a, bool bkm) {
  bool vec = aligned16(a.A) && aligned16(a.W) && (a.g.Cs % 4 == 0) && (a.K % 4 == 0) && (a.st % 4 == 0) &&
             (a.sAb % 4 == 0) && (a.sWb % 4 == 0);
  if (bkm) vec = vec && (a.sk % 4 == 0) && (a.N % 4 == 0);
  else vec = vec && (a.sn % 4 == 0);
  if (a.xf != VAE_XF_NONE) vec = vec && aligned16(a.scale) && aligned16(a.shift);
  return vec;
}
static bool rows_use_tile(const vae_igemm_args& a, bool vec, bool bkm) {
  return conv3_tile_eligible(a, vec, bkm) && !vae_opt().flat_conv;
}
// the bf16 halo-tile kernel reads the weights from their bf16 image; without one the bf16 flat kernel serves the layer
static bool rows_use_tile_bf16(const vae_igemm_args& a, bool vec, bool bkm) {
  return a.prec == VAE_PREC_BF16 && rows_use_tile(a, vec, bkm) && conv3_tile_bf16_packed(a);
}

// the wide-tile kernel serves a layer the 128-pixel bf16 halo-tile kernel would serve, when both operands are bf16 images
static bool rows_use_wide_bf16(const vae_igemm_args& a, bool vec, bool bkm) {
  return rows_use_tile_bf16(a, vec, bkm) && conv3_wide_bf16_eligible(a) && !vae_opt().no_wide;
}

// both the forward and the wgrad of this 3x3 stride-1 layer run on the bf16 halo-tile kernels (which can read a bf16
// activation image); pointers are placeholders with the alignment the real ones must have
extern "C" int vae_bf16_act_image_ok(const vae_conv_geom* gp, int32_t Cout, int32_t Cin) {
  if (!gp || vae_opt().flat_conv) return 0;
  const vae_conv_geom& g = *gp;
  if (g.mode != VAE_MODE_FWD || Cin % 8 != 0 || g.Cs != Cin) return 0;
  static const float dummy[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};
  vae_igemm_args f{};
  f.A = f.W = dummy; f.C = const_cast<float*>(dummy); f.Wh = dummy;
  f.g = g; f.M = g.B * g.Ho * g.Wo; f.N = Cout; f.K = Cin; f.ldc = Cout;
  f.sn = (int64_t)g.taps * Cin; f.sk = 1; f.st = Cin; f.batch = 1; f.alpha = 1.f; f.prec = VAE_PREC_BF16; f.xf = VAE_XF_NONE;
  if (!rows_use_tile_bf16(f, rows_vec(f, false), false)) return 0;
  vae_wgrad_args w{};
  w.dY = w.X = dummy; w.g = g; w.M = Cout; w.N = Cin; w.ldy = Cout; w.npix = f.M; w.nsplit = 1; w.batch = 1; w.alpha = 1.f;
  w.prec = VAE_PREC_BF16; w.xf = VAE_XF_NONE;
  return wgrad_use_tile_bf16(w) ? 1 : 0;
}

// the output gradient of this 3x3 stride-1 layer may be handed over as a bf16 image: its dgrad (A16) and its weight
// gradient (dY16) both run on the bf16 halo-tile kernels
extern "C" int vae_bf16_grad_image_ok(const vae_conv_geom* gp, int32_t Cout, int32_t Cin) {
  if (!gp || vae_opt().flat_conv) return 0;
  const vae_conv_geom& g = *gp;
  if (g.mode != VAE_MODE_FWD || g.taps != 9 || g.stride != 1 || Cin % 8 != 0 || Cout % 8 != 0 || g.Ho != g.Hs || g.Wo != g.Ws) return 0;
  static const float dummy[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};
  vae_igemm_args d{};  // the dgrad launch ops.conv_dgrad builds
  d.A = d.W = dummy; d.C = const_cast<float*>(dummy); d.Wh = dummy;
  d.g = g; d.g.Cs = Cout; d.g.mode = VAE_MODE_DGRAD;
  d.M = g.B * g.Ho * g.Wo; d.N = Cin; d.K = Cout; d.ldc = Cin;
  d.sn = 1; d.sk = (int64_t)g.taps * Cin; d.st = Cin; d.batch = 1; d.alpha = 1.f; d.prec = VAE_PREC_BF16; d.xf = VAE_XF_NONE;
  if (!rows_use_tile_bf16(d, rows_vec(d, true), true)) return 0;
  vae_wgrad_args w{};
  w.dY = w.X = dummy; w.g = g; w.g.Cs = Cin; w.M = Cout; w.N = Cin; w.ldy = Cout; w.npix = d.M; w.nsplit = 1; w.batch = 1; w.alpha = 1.f;
  w.prec = VAE_PREC_BF16; w.xf = VAE_XF_NONE;
  return wgrad_use_tile_bf16(w) ? 1 : 0;
}

static bool rows_thin_mfma(const vae_igemm_args& a) { return conv_thin_bf16_eligible(a) && !vae_opt().no_thin_mfma; }
static bool rows_thinn_mfma(const vae_igemm_args& a) { return conv_thinn_bf16_eligible(a) && !vae_opt().no_thin_mfma; }
static bool rows_is_phase(const vae_igemm_args& a) { return a.tapmask != 0 || a.a_step > 1 || a.c_step > 1; }
static bool rows_wino(const vae_igemm_args& a) {
  const bool bkm = rows_bkm(a);
  return conv3_wino_eligible(a) && rows_vec(a, bkm) && !conv_smallk_eligible(a) && !conv_smalln_eligible(a) && !vae_opt().flat_conv &&
         !vae_opt().no_wino;
}
// ... and of those, the layers whose maps are whole 16 x 32 tiles with whole 64-channel blocks run F(4x4,3x3) (36 positions, 36
// instead of 64 multiplications per 4x4 outputs); library option "no_wino4" keeps them on F(2x2,3x3)
static bool rows_wino4(const vae_igemm_args& a) { return rows_wino(a) && conv3_wino4_eligible(a) && !vae_opt().no_wino4; }
// the upsampler convolution (forward over the virtual nearest-2x upsample, or its dgrad with the 2x2 sum-pool folded in) as the
// 9-position scheme of conv3_upwino.hip
static bool rows_upwino(const vae_igemm_args& a) { return conv3_upwino_eligible(a) && !vae_opt().flat_conv && !vae_opt().no_wino; }
extern "C" int vae_wino_ok(const vae_igemm_args* ap) { return (ap && (rows_wino(*ap) || rows_upwino(*ap))) ? 1 : 0; }
extern "C" int64_t vae_wino_weight_floats(const vae_igemm_args* ap) {
  if (!ap) return 0;
  const bool up = ap->g.mode == VAE_MODE_UP2X || ap->g.mode == VAE_MODE_UP2X_DGRAD;
  return (int64_t)(up ? 9 : (rows_wino4(*ap) ? 36 : 16)) * ap->N * ap->K;
}
extern "C" int vae_wino_weights(const vae_igemm_args* ap, float* Wu, void* stream) {
  VAE_CHECK(ap && Wu && ap->W && aligned16(Wu), "wino_weights: null or unaligned pointer");
  if (rows_upwino(*ap)) {
    if (int rc = launch_upwino_weights(*ap, Wu, (hipStream_t)stream)) return rc;
    VAE_LAUNCH_CHECK("upwino_weights");
    return VAE_OK;
  }
  VAE_CHECK(rows_wino(*ap), "wino_weights: the layer is not served by the Winograd kernel (vae_wino_ok)");
  if (int rc = (rows_wino4(*ap) ? launch_wino4_weights : launch_wino_weights)(*ap, Wu, (hipStream_t)stream)) return rc;
  VAE_LAUNCH_CHECK("wino_weights");
  return VAE_OK;
}
// An operand image (A16, xf == NONE) on a layer that no halo-tile kernel serves is, for the flat / <= 4-channel kernels, the
// same thing as "A is stored as bf16": the dispatcher rewrites it that way, so a host may hand over a bf16 tensor as A16
// without knowing which kernel will run.
static vae_igemm_args rows_canon(const vae_igemm_args& a) {
  vae_igemm_args b = a;
  if (b.A16 != nullptr && b.prec == VAE_PREC_BF16 && b.xf == VAE_XF_NONE && !(b.tapmask != 0 || b.a_step > 1 || b.c_step > 1)) {
    vae_igemm_args t = b;  // (the vectorisation test reads the pointer the kernel would read)
    if (t.A == nullptr) t.A = reinterpret_cast<const float*>(t.A16);
    const bool bkm = rows_bkm(t);
    if (!rows_use_tile_bf16(t, rows_vec(t, bkm), bkm)) {
      b.A = reinterpret_cast<const float*>(b.A16);
      b.A16 = nullptr;
      b.a_bf16 = 1;
    }
  }
  return b;
}
// Storage flags (vaehip.h: out_bf16 / a_bf16 / res_bf16): does the kernel that serves `a` honour them as they are set?  Follows
// the dispatch order of vae_igemm_rows.
static bool tile16_flags_ok(const vae_igemm_args& a) {  // the 128-pixel bf16 halo-tile kernel (conv3_tile_bf16.hip)
  return !a.a_bf16 && (!a.out_bf16 || (a.track == nullptr && a.ldc % 2 == 0 && a.N % 2 == 0)) &&
         (a.res == nullptr || (a.res_bf16 != 0) == (a.out_bf16 != 0));
}
static bool rows_io16_ok(const vae_igemm_args& a0) {
  const vae_igemm_args a = rows_canon(a0);
  if (!a.out_bf16 && !a.a_bf16 && !a.res_bf16) return true;
  if (a.prec != VAE_PREC_BF16 || a.Wu != nullptr) return false;  // fp32-arithmetic kernels: fp32 storage
  if (a.res_bf16 && a.res == nullptr) return false;
  const bool bkm = rows_bkm(a), vec = rows_vec(a, bkm);
  if (rows_is_phase(a)) {
    if (rows_use_wide_bf16(a, vec, bkm)) return true;  // (its eligibility covers the flags)
    return a.A16 == nullptr && a.xf == VAE_XF_NONE && rows_use_tile_bf16(a, vec, bkm) && tile16_flags_ok(a);
  }
  if (a.A16 == nullptr && conv_smallk_eligible(a)) return !a.a_bf16 && !a.res_bf16;  // wide side = the output
  if (conv_smalln_eligible(a)) return !a.out_bf16 && !a.res_bf16;                     // wide side = the input
  if (rows_use_wide_bf16(a, vec, bkm)) return true;
  if (rows_use_tile_bf16(a, vec, bkm)) return tile16_flags_ok(a);
  return vec && (a.A16 == nullptr);  // the bf16 flat kernels take any combination; unvectorised shapes run the fp32 kernel
}
extern "C" int vae_conv_io16_ok(const vae_igemm_args* ap) { return (ap && rows_io16_ok(*ap)) ? 1 : 0; }
extern "C" int vae_conv_phase_ok(const vae_igemm_args* ap) {
  if (!ap) return 0;
  const vae_igemm_args& a = *ap;
  const bool bkm = rows_bkm(a), vec = rows_vec(a, bkm);
  if (a.A16 != nullptr) return rows_use_wide_bf16(a, vec, bkm) ? 1 : 0;  // operand image: the wide-tile kernel's 2x2 tap blocks
  if (a.prec == VAE_PREC_BF16) return (a.xf == VAE_XF_NONE && rows_use_tile_bf16(a, vec, bkm)) ? 1 : 0;  // no transform variant there
  return rows_use_tile(a, vec, bkm) ? 1 : 0;
}
extern "C" int vae_conv_gnb_chunks(const vae_igemm_args* ap) {
  if (!ap) return 0;
  const vae_igemm_args a = rows_canon(*ap);
  if (a.Wu != nullptr && rows_wino4(a)) return conv3_wino4_gnb_chunks(a);
  if (a.Wu != nullptr && rows_wino(a)) return conv3_wino_gnb_chunks(a);
  return 0;  // (the other dgrad kernels have no such epilogue yet)
}

extern "C" int vae_conv_gstat_chunks(const vae_igemm_args* ap) {
  if (!ap) return 0;
  const vae_igemm_args a = rows_canon(*ap);
  if (a.Wu != nullptr && rows_wino4(a)) return conv3_wino4_gstat_chunks(a);
  if (a.Wu != nullptr) return conv3_wino_eligible(a) ? conv3_wino_gstat_chunks(a) : 0;
  const bool bkm = rows_bkm(a), vec = rows_vec(a, bkm);
  if ((a.A16 == nullptr && conv_smallk_eligible(a)) || conv_smalln_eligible(a)) return 0;
  if (rows_use_wide_bf16(a, vec, bkm)) return conv3_wide_bf16_gstat_chunks(a);
  if (rows_use_tile_bf16(a, vec, bkm)) return conv3_tile_bf16_gstat_chunks(a);
  if (a.prec != VAE_PREC_BF16 && rows_use_tile(a, vec, bkm)) return conv3_tile_gstat_chunks(a);
  return 0;
}

// name of the kernel instantiation vae_igemm_rows / vae_wgrad dispatch to for these arguments
// (profiling labels that match the rocprofv3 kernel names; no launch)
extern "C" int vae_igemm_kernel_name(const vae_igemm_args* ap, char* buf, int32_t n) {
  VAE_CHECK(ap && buf && n > 0, "igemm_kernel_name: bad args");
  const vae_igemm_args a = rows_canon(*ap);
  const bool bkm = rows_bkm(a), vec = rows_vec(a, bkm);
  const char* tf[2] = {"false", "true"};
  if (a.Wu != nullptr && rows_upwino(a))
    snprintf(buf, n, "conv3_upwino_kernel<%s>", tf[a.g.mode == VAE_MODE_UP2X_DGRAD]);
  else if (a.Wu != nullptr && rows_wino4(a))
    snprintf(buf, n, "conv3_wino4_kernel<%d>", a.xf);
  else if (a.Wu != nullptr && rows_wino(a))
    snprintf(buf, n, "conv3_wino_kernel<%d,%d>", a.xf, conv3_wino_nb());
  else if (rows_is_phase(a) && rows_use_wide_bf16(a, vec, bkm))
    snprintf(buf, n, "conv3_wide_bf16_kernel<%s,2>", tf[a.g.mode == VAE_MODE_DGRAD]);
  else if (rows_is_phase(a) && a.prec == VAE_PREC_BF16)
    snprintf(buf, n, "conv3_tile_bf16_kernel<%s,%s,%d,false>", tf[a.g.mode == VAE_MODE_DGRAD], tf[a.g.mode == VAE_MODE_UP2X], a.xf);
  else if (rows_is_phase(a))
    snprintf(buf, n, "conv3_tile_kernel<%s,%s,%s,%d>", tf[bkm], tf[a.g.mode == VAE_MODE_DGRAD], tf[a.g.mode == VAE_MODE_UP2X], a.xf);
  else if (conv_smallk_eligible(a) && a.A16 == nullptr && rows_thin_mfma(a))
    snprintf(buf, n, "conv_thin_bf16_kernel");
  else if (conv_smallk_eligible(a))
    snprintf(buf, n, "conv_smallk_kernel");
  else if (conv_smalln_eligible(a) && rows_thinn_mfma(a))
    snprintf(buf, n, "conv_thinn_bf16_kernel<%d>", a.xf);
  else if (conv_smalln_eligible(a))
    snprintf(buf, n, "conv_smalln_kernel<%d>", a.xf);
  else if (rows_use_wide_bf16(a, vec, bkm))
    snprintf(buf, n, "conv3_wide_bf16_kernel<%s,3>", tf[a.g.mode == VAE_MODE_DGRAD]);
  else if (rows_use_tile_bf16(a, vec, bkm))
    snprintf(buf, n, "conv3_tile_bf16_kernel<%s,%s,%d,%s>", tf[a.g.mode == VAE_MODE_DGRAD], tf[a.g.mode == VAE_MODE_UP2X], a.xf,
             tf[a.A16 != nullptr]);
  else if (a.prec != VAE_PREC_BF16 && rows_use_tile(a, vec, bkm))
    snprintf(buf, n, "conv3_tile_kernel<%s,%s,%s,%d>", tf[bkm], tf[a.g.mode == VAE_MODE_DGRAD], tf[a.g.mode == VAE_MODE_UP2X], a.xf);
  else if (a.prec == VAE_PREC_BF16 && vec && !vae_opt().flat_conv && conv1_bf16_eligible(a))
    snprintf(buf, n, "conv1_bf16_kernel<%s,%d>", tf[a.g.mode == VAE_MODE_DGRAD], a.K / 16);
  else if (a.prec == VAE_PREC_BF16 && vec)
    snprintf(buf, n, "igemm_rows_bf16_kernel<%s,%s,%d>", a.N <= 32 ? "128,32,4,1" : "128,128,4,2", tf[bkm], a.xf);
  else if (a.N <= 32)
    snprintf(buf, n, "igemm_rows_kernel<128,32,4,1,%s,%s,%d>", tf[bkm], tf[vec], a.xf);
  else
    snprintf(buf, n, "igemm_rows_kernel<128,128,4,2,%s,%s,%d>", tf[bkm], tf[vec], a.xf);
  return VAE_OK;
}
extern "C" int vae_wgrad_kernel_name(const vae_wgrad_args* ap, char* buf, int32_t n) {
  VAE_CHECK(ap && buf && n > 0, "wgrad_kernel_name: bad args");
  const vae_wgrad_args a = wgrad_canon(*ap);
  const bool vec = wgrad_vec(a);
  const char* tf[2] = {"false", "true"};
  if (wgrad_is_phase(a) && a.prec == VAE_PREC_BF16 && wgrad3_tile_bf16_dma(a)) snprintf(buf, n, "wgrad3_dma_bf16_kernel<false,1>");
  else if (wgrad_is_phase(a) && a.prec == VAE_PREC_BF16)
    snprintf(buf, n, "wgrad3_tile_bf16_kernel<false,%d,%s,%s>", a.xf, tf[a.X16 != nullptr], tf[a.dY16 != nullptr]);
  else if (wgrad_is_phase(a)) snprintf(buf, n, "wgrad3_tile_kernel<%s,%d>", tf[a.g.mode == VAE_MODE_UP2X], a.xf);
  else if (wgrad_smallk_kind(a) && wgrad_smallk_on_mfma(a)) snprintf(buf, n, "wgrad_thin_bf16_kernel<%s,%d>", tf[wgrad_smallk_kind(a) == 1], a.xf);
  else if (wgrad_smallk_kind(a)) snprintf(buf, n, "wgrad_smallk_kernel<%s,%d>", tf[wgrad_smallk_kind(a) == 1], a.xf);
  else if (wgrad_use_tile_bf16(a) && wgrad3_tile_bf16_dma(a)) snprintf(buf, n, "wgrad3_dma_bf16_kernel<%s,%d>", tf[a.g.mode == VAE_MODE_UP2X], a.g.stride);
  else if (wgrad_use_tile_bf16(a)) snprintf(buf, n, "wgrad3_tile_bf16_kernel<%s,%d,%s,%s>", tf[a.g.mode == VAE_MODE_UP2X], a.xf, tf[a.X16 != nullptr], tf[a.dY16 != nullptr]);
  else if (wgrad_use_tile(a)) snprintf(buf, n, "wgrad3_tile_kernel<%s,%d>", tf[a.g.mode == VAE_MODE_UP2X], a.xf);
  else if (a.prec == VAE_PREC_BF16 && vec)
    snprintf(buf, n, "wgrad_bf16_kernel<%s,%d>", a.M <= 32 ? "32,128,1,4" : (a.N <= 32 ? "128,32,4,1" : "128,128,4,2"), a.xf);
  else snprintf(buf, n, "wgrad_kernel<%s,%s,%d>", a.M <= 32 ? "32,128,1,4" : (a.N <= 32 ? "128,32,4,1" : "128,128,4,2"), tf[vec], a.xf);
  return VAE_OK;
}

extern "C" int vae_igemm_rows(const vae_igemm_args* ap, void* stream) {
  VAE_CHECK(ap != nullptr, "igemm_rows: null args");
  const vae_igemm_args a = rows_canon(*ap);
  ap = &a;
  if (int e = check_geom("igemm_rows", a.g)) return e;
  VAE_CHECK(a.A && a.W && a.C, "igemm_rows: null operand");
  VAE_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "igemm_rows: bad sizes M=%d N=%d K=%d", a.M, a.N, a.K);
  VAE_CHECK(a.K <= a.g.Cs, "igemm_rows: K=%d exceeds source channels %d", a.K, a.g.Cs);
  VAE_CHECK((int64_t)a.g.B * a.g.Ho * a.g.Wo == a.M, "igemm_rows: M=%d != B*Ho*Wo", a.M);
  VAE_CHECK(a.ldc >= a.N, "igemm_rows: ldc < N");
  VAE_CHECK((size_t)a.M * a.ldc * 4u < BUF_MAX, "igemm_rows: output too large for 32-bit byte offsets");
  VAE_CHECK(a.prec == VAE_PREC_F32 || a.prec == VAE_PREC_BF16, "igemm_rows: bad prec %d", a.prec);
  VAE_CHECK(a.sk == 1 || a.sn == 1, "igemm_rows: one of sn, sk must be 1 (sn=%lld sk=%lld)", (long long)a.sn,
            (long long)a.sk);
  VAE_CHECK(a.xf == VAE_XF_NONE || (a.scale && a.shift), "igemm_rows: xf needs scale/shift");
  VAE_CHECK(a.xf == VAE_XF_NONE || xf_rows_ok(a.g, a.M, a.K),
            "igemm_rows: fused GroupNorm needs the tile's scale/shift rows to fit LDS (see vae_xf_fusable_rows)");
  const bool bkm = rows_bkm(a);
  const bool vec = rows_vec(a, bkm);
  VAE_CHECK(a.gstat == nullptr || vae_conv_gstat_chunks(ap) > 0, "igemm_rows: no statistics epilogue for these arguments (vae_conv_gstat_chunks)");
  VAE_CHECK(a.gnb_ws == nullptr || vae_conv_gnb_chunks(ap) > 0, "igemm_rows: no GroupNorm-backward epilogue for these arguments (vae_conv_gnb_chunks)");
  VAE_CHECK(a.gnb_ws == nullptr || (a.gnb_mean && a.gnb_rstd && a.gnb_gamma && a.gnb_beta && aligned16(a.gnb_x)), "igemm_rows: gnb_* pointers");
  VAE_CHECK(rows_io16_ok(a), "igemm_rows: the kernel serving these arguments does not take this combination of out_bf16 / a_bf16 / res_bf16 (vae_conv_io16_ok)");
  hipStream_t st = (hipStream_t)stream;
  VAE_CHECK(a.g.mode != VAE_MODE_UP2X_DGRAD || a.Wu != nullptr, "igemm_rows: UP2X_DGRAD exists only as the Winograd-type kernel (vae_wino_ok, Wu)");
  if (a.Wu != nullptr && rows_upwino(a)) {  // the upsampler convolution with ITS transformed weights (9 positions)
    VAE_CHECK(aligned16(a.Wu), "igemm_rows: unaligned Wu");
    if (int rc2 = launch_conv3_upwino(a, a.Wu, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_upwino");
    return VAE_OK;
  }
  if (a.Wu != nullptr && rows_wino4(a)) {  // Winograd F(4x4,3x3): Wu holds 36 positions (vae_wino_weights built it under the same options)
    VAE_CHECK(aligned16(a.Wu), "igemm_rows: unaligned Wu");
    if (int rc2 = launch_conv3_wino4(a, a.Wu, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_wino4");
    return VAE_OK;
  }
  if (a.Wu != nullptr) {  // Winograd F(2x2,3x3) with the transformed weights the caller built for THIS geometry
    VAE_CHECK(rows_wino(a) && aligned16(a.Wu), "igemm_rows: Wu needs a layer vae_wino_ok accepts");
    if (int rc2 = launch_conv3_wino(a, a.Wu, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_wino");
    return VAE_OK;
  }
  if (rows_is_phase(a)) {  // sub-sampled views / tap subsets: only the fp32 halo-tile kernel implements them
    VAE_CHECK(vae_conv_phase_ok(ap), "igemm_rows: tapmask / a_step / c_step need a halo-tile kernel (vae_conv_phase_ok)");
    VAE_CHECK(a.track == nullptr && a.gstat == nullptr, "igemm_rows: no tracker / statistics epilogue on a sub-sampled output");
    if (rows_use_wide_bf16(a, vec, bkm)) {
      if (int rc2 = launch_conv3_wide_bf16(a, st)) return rc2;
      VAE_LAUNCH_CHECK("conv3_wide_bf16");
      return VAE_OK;
    }
    VAE_CHECK(a.A16 == nullptr, "igemm_rows: a sub-sampled view of an operand image needs the wide-tile kernel (vae_conv_phase_ok)");
    if (int rc2 = (a.prec == VAE_PREC_BF16) ? launch_conv3_tile_bf16(a, bkm, st) : launch_conv3_tile(a, bkm, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_tile");
    return VAE_OK;
  }
  if (a.A16 == nullptr && conv_smallk_eligible(a) && rows_thin_mfma(a)) {  // bf16 mode, bf16 output: the same launch on the matrix pipe
    if (int rc2 = launch_conv_thin_bf16(a, st)) return rc2;
    VAE_LAUNCH_CHECK("conv_thin_bf16");
    return VAE_OK;
  }
  if (a.A16 == nullptr && conv_smallk_eligible(a)) {
    if (int rc2 = launch_conv_smallk(a, st)) return rc2;
    VAE_LAUNCH_CHECK("conv_smallk");
    return VAE_OK;
  }
  if (conv_smalln_eligible(a) && rows_thinn_mfma(a)) {  // bf16 mode, bf16 input: the same launch on the matrix pipe
    if (int rc2 = launch_conv_thinn_bf16(a, st)) return rc2;
    VAE_LAUNCH_CHECK("conv_thinn_bf16");
    return VAE_OK;
  }
  if (conv_smalln_eligible(a)) {
    if (int rc2 = launch_conv_smalln(a, st)) return rc2;
    VAE_LAUNCH_CHECK("conv_smalln");
    return VAE_OK;
  }
  VAE_CHECK(a.A16 == nullptr || (a.xf == VAE_XF_NONE && rows_use_tile_bf16(a, vec, bkm) && aligned16(a.A16) && a.g.Cs % 8 == 0),
            "igemm_rows: A16 needs bf16 mode, xf == NONE and a layer vae_bf16_act_image_ok accepts");
  if (rows_use_wide_bf16(a, vec, bkm)) {
    if (int rc2 = launch_conv3_wide_bf16(a, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_wide_bf16");
    return VAE_OK;
  }
  if (rows_use_tile_bf16(a, vec, bkm)) {
    if (int rc2 = launch_conv3_tile_bf16(a, bkm, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_tile_bf16");
    return VAE_OK;
  }
  if (a.prec != VAE_PREC_BF16 && rows_use_tile(a, vec, bkm)) {  // 3x3 stride-1: LDS halo tile shared by the 9 taps
    if (int rc2 = launch_conv3_tile(a, bkm, st)) return rc2;
    VAE_LAUNCH_CHECK("conv3_tile");
    return VAE_OK;
  }
  if (vec) {  // flat vectorised kernels address one tile's images / the weights with 32-bit byte offsets
    const int64_t rows_per_img = (a.g.mode == VAE_MODE_DGRAD_S2) ? (int64_t)a.g.Ho * a.g.Wo / 4 : (int64_t)a.g.Ho * a.g.Wo;
    const int64_t span = std::min<int64_t>(a.g.B, rows_per_img % 128 == 0 ? 1 : 127 / rows_per_img + 2);
    VAE_CHECK((size_t)span * a.g.Hs * a.g.Ws * a.g.Cs * 4u < BUF_MAX && (size_t)std::max(a.K * a.sk, a.N * a.sn) * 4u < BUF_MAX,
              "igemm_rows: operand too large for 32-bit byte offsets");
  }
  int rc;
  if (a.prec == VAE_PREC_BF16 && vec && !vae_opt().flat_conv && conv1_bf16_eligible(a)) {
    rc = launch_conv1_bf16(a, st);
  } else if (a.prec == VAE_PREC_BF16 && vec) {
    VAE_CHECK(!bkm || a.xf == VAE_XF_NONE, "igemm_rows: xf unsupported with n-contiguous weights");
    rc = launch_rows_bf16(a, bkm, st);
  } else {
    rc = (a.N <= 32) ? launch_rows<128, 32, 4, 1>(a, bkm, vec, st) : launch_rows<128, 128, 4, 2>(a, bkm, vec, st);
  }
  if (rc) return rc;
  VAE_LAUNCH_CHECK("igemm_rows");
  return VAE_OK;
}

// Winograd F(3x3,2x2) weight gradient (wgrad3_wino.hip): plan (nsplit = 0: not served), launch into the transform-domain slab
// [nsplit][16][Cin][Cout] (a->partial; a->bias_partial optional), and the reduction + output transform into OHWI
extern "C" int vae_wgrad_wino_plan(const vae_wgrad_args* ap, int32_t* nsplit) {
  VAE_CHECK(ap && nsplit, "wgrad_wino_plan: null argument");
  *nsplit = 0;
  if (vae_opt().flat_conv || vae_opt().no_wino) return VAE_OK;
  if (wgrad3_upwino_eligible(*ap)) {  // the upsampler convolution: 9 positions, one 8-wave workgroup per CU (158 registers)
    const int64_t units = wgrad3_upwino_units(ap->g);
    const int64_t wgs = (int64_t)(ap->M / 128) * (ap->N / 32);
    *nsplit = (int32_t)std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(wgs, 1), units / 8));
    return VAE_OK;
  }
  if (!wgrad3_wino_eligible(*ap)) return VAE_OK;
  const int64_t units = wgrad3_wino_units(ap->g);
  const int64_t wgs = (int64_t)(ap->M / 128) * (ap->N / 32);
  *nsplit = (int32_t)std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(wgs, 1), units / 8));  // one 8-wave workgroup per CU
  return VAE_OK;
}
extern "C" int vae_wgrad_wino(const vae_wgrad_args* ap, void* stream) {
  VAE_CHECK(ap != nullptr, "wgrad_wino: null args");
  const vae_wgrad_args& a = *ap;
  if (int e = check_geom("wgrad_wino", a.g)) return e;
  VAE_CHECK(a.dY && a.X && a.partial, "wgrad_wino: null operand");
  VAE_CHECK(a.nsplit > 0 && a.nsplit <= 65535, "wgrad_wino: bad nsplit");
  VAE_CHECK((int64_t)a.g.B * a.g.Ho * a.g.Wo == a.npix && a.N <= a.g.Cs && a.ldy >= a.M, "wgrad_wino: inconsistent sizes");
  if (wgrad3_upwino_eligible(a)) {
    if (int rc = launch_wgrad3_upwino(a, (hipStream_t)stream)) return rc;
    VAE_LAUNCH_CHECK("wgrad3_upwino");
    return VAE_OK;
  }
  VAE_CHECK(wgrad3_wino_eligible(a), "wgrad_wino: the layer is not served by the Winograd kernel (vae_wgrad_wino_plan)");
  VAE_CHECK(a.xf == VAE_XF_NONE || (a.scale && a.shift), "wgrad_wino: xf needs scale/shift");
  if (int rc = launch_wgrad3_wino(a, (hipStream_t)stream)) return rc;
  VAE_LAUNCH_CHECK("wgrad3_wino");
  return VAE_OK;
}
static int reduce_splits_impl(const float* partial, int32_t nsplit, int64_t n, float* out, const float* partial2, int32_t n2, float* out2,
                              hipStream_t st);
extern "C" int vae_wgrad_wino_positions(const vae_wgrad_args* ap) { return (ap && wgrad3_upwino_eligible(*ap)) ? 9 : 16; }
extern "C" int vae_wgrad_wino_reduce(const float* slab, int32_t nsplit, int32_t npos, int32_t Cin, int32_t Cout, float* scratch, float* dW,
                                     const float* bias_partial, float* db, void* stream) {
  VAE_CHECK(slab && dW && nsplit > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, "wgrad_wino_reduce: bad args");
  VAE_CHECK(npos == 16 || npos == 9, "wgrad_wino_reduce: npos must be 16 (plain 3x3 layer) or 9 (upsampler convolution)");
  VAE_CHECK((bias_partial == nullptr) == (db == nullptr), "wgrad_wino_reduce: bias_partial and db go together");
  VAE_CHECK(nsplit == 1 || scratch != nullptr, "wgrad_wino_reduce: nsplit > 1 needs the [16*Cin*Cout] scratch buffer");
  hipStream_t st = (hipStream_t)stream;
  if (nsplit > 1) {  // wide fixed-order sum over the splits first (the slab of a 128-channel layer is 64 x 1 MB), then the transform
    if (int rc = reduce_splits_impl(slab, nsplit, (int64_t)npos * Cin * Cout, scratch, bias_partial, bias_partial ? Cout : 0, db, st)) return rc;
    VAE_LAUNCH_CHECK("reduce_splits");
    if (int rc = (npos == 9 ? launch_upwino_wgrad_reduce : launch_wino_wgrad_reduce)(scratch, 1, Cin, Cout, dW, nullptr, nullptr, st)) return rc;
  } else {
    if (int rc = (npos == 9 ? launch_upwino_wgrad_reduce : launch_wino_wgrad_reduce)(slab, 1, Cin, Cout, dW, bias_partial, db, st)) return rc;
  }
  VAE_LAUNCH_CHECK("wino_wgrad_reduce");
  return VAE_OK;
}

extern "C" int vae_wgrad_phase_ok(const vae_wgrad_args* ap) {
  if (!ap || wgrad_smallk_kind(*ap)) return 0;
  if (ap->prec == VAE_PREC_BF16) return (ap->xf == VAE_XF_NONE && wgrad_use_tile_bf16(*ap)) ? 1 : 0;  // the bf16 halo-tile kernel
  return (ap->X16 == nullptr && wgrad_use_tile(*ap)) ? 1 : 0;
}
// storage flags of the weight gradient's operands (vaehip.h: x_bf16 / y_bf16); follows vae_wgrad's dispatch order
static bool wgrad_io16_ok(const vae_wgrad_args& a0) {
  const vae_wgrad_args a = wgrad_canon(a0);
  if (!a.x_bf16 && !a.y_bf16) return true;
  if (a.prec != VAE_PREC_BF16 || wgrad_is_phase(a)) return false;  // the halo-tile kernels take images through X16 / dY16
  if (a.X16 == nullptr) {
    const int kind = wgrad_smallk_kind(a);
    if (kind) return kind == 1 ? !a.x_bf16 : !a.y_bf16;  // only the wide side may be bf16 (kind 1: dY, kind 2: X)
  }
  if (a.X16 != nullptr || a.dY16 != nullptr || wgrad_use_tile_bf16(a) || wgrad_use_tile(a)) return false;
  return wgrad_vec(a);  // the bf16 flat kernel
}
extern "C" int vae_wgrad_io16_ok(const vae_wgrad_args* ap) { return (ap && wgrad_io16_ok(*ap)) ? 1 : 0; }

extern "C" int vae_wgrad(const vae_wgrad_args* ap, void* stream) {
  VAE_CHECK(ap != nullptr, "wgrad: null args");
  const vae_wgrad_args a = wgrad_canon(*ap);
  ap = &a;
  VAE_CHECK(wgrad_io16_ok(a), "wgrad: the kernel serving these arguments does not take x_bf16 / y_bf16 as set (vae_wgrad_io16_ok)");
  if (int e = check_geom("wgrad", a.g)) return e;
  VAE_CHECK((a.dY || a.dY16) && a.X, "wgrad: null operand");
  VAE_CHECK(a.dY16 == nullptr || (wgrad_use_tile_bf16(a) && !(a.X16 == nullptr && wgrad_smallk_kind(a)) && aligned16(a.dY16) && a.ldy % 8 == 0 && a.M % 8 == 0),
            "wgrad: dY16 needs bf16 mode and a layer vae_bf16_grad_image_ok accepts");
  VAE_CHECK(a.M > 0 && a.N > 0 && a.npix > 0 && a.nsplit > 0 && a.batch > 0, "wgrad: bad sizes");
  VAE_CHECK(a.N <= a.g.Cs, "wgrad: N exceeds source channels");
  VAE_CHECK((int64_t)a.g.B * a.g.Ho * a.g.Wo == a.npix, "wgrad: npix != B*Ho*Wo");
  VAE_CHECK(a.ldy >= a.M, "wgrad: ldy < M");
  VAE_CHECK(a.g.mode != VAE_MODE_DGRAD && a.g.mode != VAE_MODE_DGRAD_S2, "wgrad: dgrad geometry not valid here");
  VAE_CHECK(a.nsplit == 1 ? a.out != nullptr : a.partial != nullptr, "wgrad: missing output buffer");
  VAE_CHECK(a.xf == VAE_XF_NONE || (a.scale && a.shift), "wgrad: xf needs scale/shift");
  VAE_CHECK(a.bias_partial == nullptr || a.batch == 1, "wgrad: bias_partial is for batch == 1 only");
  const bool vec = wgrad_vec(a);
  hipStream_t st = (hipStream_t)stream;
  VAE_CHECK(a.prec == VAE_PREC_F32 || a.prec == VAE_PREC_BF16, "wgrad: bad prec %d", a.prec);
  if (wgrad_is_phase(a)) {  // sub-sampled dY / tap subsets: the halo-tile kernels implement them
    VAE_CHECK(vae_wgrad_phase_ok(ap), "wgrad: tapmask / y_step need a halo-tile kernel (vae_wgrad_phase_ok)");
    VAE_CHECK(a.nsplit <= 65535, "wgrad: nsplit too large");
    if (a.prec == VAE_PREC_BF16) {
      VAE_CHECK(a.X16 == nullptr || (aligned16(a.X16) && a.g.Cs % 8 == 0), "wgrad: unaligned X16");
      VAE_CHECK(a.dY16 == nullptr || (aligned16(a.dY16) && a.ldy % 8 == 0 && a.M % 8 == 0), "wgrad: unaligned dY16");
      if (int rc2 = launch_wgrad3_tile_bf16(a, st)) return rc2;
      VAE_LAUNCH_CHECK("wgrad3_tile_bf16");
      return VAE_OK;
    }
    if (int rc2 = launch_wgrad3_tile(a, st)) return rc2;
    VAE_LAUNCH_CHECK("wgrad3_tile");
    return VAE_OK;
  }
  if (a.X16 == nullptr && wgrad_smallk_kind(a)) {
    VAE_CHECK(a.nsplit <= 65535, "wgrad: nsplit too large");
    if (int rc2 = launch_wgrad_smallk(a, st)) return rc2;
    VAE_LAUNCH_CHECK("wgrad_smallk");
    return VAE_OK;
  }
  VAE_CHECK(a.X16 == nullptr || (a.xf == VAE_XF_NONE && wgrad_use_tile_bf16(a) && aligned16(a.X16) && a.g.Cs % 8 == 0),
            "wgrad: X16 needs bf16 mode, xf == NONE and a layer vae_bf16_act_image_ok accepts");
  if (wgrad_use_tile_bf16(a)) {
    VAE_CHECK(a.nsplit <= 65535, "wgrad: nsplit too large");
    if (int rc2 = launch_wgrad3_tile_bf16(a, st)) return rc2;
    VAE_LAUNCH_CHECK("wgrad3_tile_bf16");
    return VAE_OK;
  }
  if (wgrad_use_tile(a)) {  // 3x3 stride-1: the nine taps share one staged dY tile + X halo
    VAE_CHECK(a.nsplit <= 65535, "wgrad: nsplit too large");
    if (int rc2 = launch_wgrad3_tile(a, st)) return rc2;
    VAE_LAUNCH_CHECK("wgrad3_tile");
    return VAE_OK;
  }
  VAE_CHECK(a.xf == VAE_XF_NONE || xf_wgrad_ok(a.g, a.npix, a.nsplit, a.N),
            "wgrad: fused GroupNorm needs the split's scale/shift rows to fit LDS (see vae_wgrad_plan)");
  if (vec) {  // 32-bit byte offsets inside one split's pixel range
    const int64_t hw = (int64_t)a.g.Ho * a.g.Wo;
    int64_t chunk = (a.npix + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + 31) / 32 * 32;
    const int64_t span = std::min<int64_t>(a.g.B, chunk / hw + 2);
    VAE_CHECK((size_t)chunk * a.ldy * 4u < BUF_MAX && (size_t)span * a.g.Hs * a.g.Ws * a.g.Cs * 4u < BUF_MAX,
              "wgrad: operand too large for 32-bit byte offsets (raise nsplit)");
  }
  int rc;
  if (a.prec == VAE_PREC_BF16 && vec) rc = launch_wgrad_bf16(a, st);
  else if (a.M <= 32) rc = launch_wgrad<32, 128, 1, 4>(a, vec, st);
  else if (a.N <= 32) rc = launch_wgrad<128, 32, 4, 1>(a, vec, st);
  else rc = launch_wgrad<128, 128, 4, 2>(a, vec, st);
  if (rc) return rc;
  VAE_LAUNCH_CHECK("wgrad");
  return VAE_OK;
}

static int reduce_splits_impl(const float* partial, int32_t nsplit, int64_t n, float* out, const float* partial2, int32_t n2, float* out2,
                              hipStream_t st) {
  const int extra = partial2 ? (n2 + 255) / 256 : 0;
  if (n % 4 == 0 && aligned16(partial) && aligned16(out)) {
    const int64_t n4 = n / 4;
    if (nsplit >= 32) {  // few elements, many splits: 4 threads per column quad
      const int64_t blocks = (n4 + 63) / 64;
      VAE_CHECK(blocks + extra < (1ll << 31), "reduce_splits: too many elements");
      hipLaunchKernelGGL(reduce_splits_kernel<4>, dim3((unsigned)(blocks + extra)), dim3(256), 0, st, partial, nsplit, n, out, (int)blocks, partial2, n2, out2);
    } else {
      const int64_t blocks = std::min<int64_t>((n4 + 255) / 256, 8192);
      hipLaunchKernelGGL(reduce_splits_kernel<1>, dim3((unsigned)(blocks + extra)), dim3(256), 0, st, partial, nsplit, n, out, (int)blocks, partial2, n2, out2);
    }
  } else {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reduce_splits_scalar_kernel, dim3(blocks), dim3(256), 0, st, partial, nsplit, n, out);
    if (partial2) hipLaunchKernelGGL(reduce_splits_scalar_kernel, dim3(extra), dim3(256), 0, st, partial2, nsplit, (int64_t)n2, out2);
  }
  return 0;
}
extern "C" int vae_reduce_splits(const float* partial, int32_t nsplit, int64_t n, float* out, void* stream) {
  VAE_CHECK(partial && out && nsplit > 0 && n > 0, "reduce_splits: bad args");
  if (int rc = reduce_splits_impl(partial, nsplit, n, out, nullptr, 0, nullptr, (hipStream_t)stream)) return rc;
  VAE_LAUNCH_CHECK("reduce_splits");
  return VAE_OK;
}
extern "C" int vae_reduce_splits2(const float* partial, int32_t nsplit, int64_t n, float* out, const float* partial2, int32_t n2, float* out2,
                                  void* stream) {
  VAE_CHECK(partial && out && partial2 && out2 && nsplit > 0 && n > 0 && n2 > 0, "reduce_splits2: bad args");
  if (int rc = reduce_splits_impl(partial, nsplit, n, out, partial2, n2, out2, (hipStream_t)stream)) return rc;
  VAE_LAUNCH_CHECK("reduce_splits2");
  return VAE_OK;
}
