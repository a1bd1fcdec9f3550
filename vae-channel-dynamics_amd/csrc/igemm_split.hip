// fp32 contractions of the flat implicit-GEMM geometries (stride-2 convolutions, 1x1, parity-class dgrad, attention
// linears and batched GEMMs) on the bf16 matrix pipe WITHOUT giving up fp32 accuracy: every fp32 operand is split while it is
// staged into LDS into three bf16 terms  v = h + m + l  (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 24 significand
// bits), and a product a*b is accumulated in fp32 from six partial products  ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh;
// the three dropped terms are <= 2^-24 |a*b|, i.e. below the rounding of an fp32 product.  v_mfma_f32_32x32x16_bf16 runs at 16x
// the rate of v_mfma_f32_32x32x2_f32, so six of them per 16 channels cost 6/16 of the eight fp32 MFMAs they replace
// (tools/mfma_split_bench.hip, registers only: 409 against 123 fp32-equivalent TFLOP/s).  Derived from igemm_bf16.hip: same
// loaders, tile geometry 128 x 128 with 64 x 64 wave tiles (4 waves, two workgroups per CU), k-group of 16 per step, three
// operand planes per stage in LDS (72 KB per workgroup, STATIC: with a dynamic allocation of the same size the results of
// this kernel were not repeatable when two processes shared the GPU -- tools/det_check.py, 3 of 27 runs -- and are with this one).  Measured against torch fp32 in tests/test_kernels_gpu.py at the same
// tolerances as the fp32 kernels.  OPT-IN (VAEHIP_SPLIT=1; see rows_use_split in igemm.hip for why it is not the default yet).
#include "bf16_frag.h"

namespace {

constexpr int NP = 3;  // operand planes

// v -> (h, m, l) per element, each packed for one ds_write_b64
__device__ __forceinline__ void split3(f32x4 v, uint2 (&pl)[NP]) {
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (__bf16)v[e];
    const float r1 = v[e] - (float)h[e];
    m[e] = (__bf16)r1;
    const float r2 = r1 - (float)m[e];
    l[e] = (__bf16)r2;
  }
  pl[0] = __builtin_bit_cast(uint2, h);
  pl[1] = __builtin_bit_cast(uint2, m);
  pl[2] = __builtin_bit_cast(uint2, l);
}

template <int BM, int BN, int WM, int WN, bool BKM, int XF>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_rows_split_kernel(vae_igemm_args p) {
  constexpr int BK = 16;                      // one MFMA k-group per step
  constexpr int KQ = BK / 4;                  // float4 per tile row
  constexpr int NT = 64 * WM * WN;
  constexpr int RP = NT / KQ;                 // tile rows per loader pass
  constexpr int LDA = BK + 8;                 // 48 B rows: conflict-free ds_read_b128
  constexpr int LDB = BKM ? (BN + 32) : (BK + 8);
  constexpr int SA = BM * LDA;
  constexpr int SB = BKM ? BK * LDB : BN * LDB;
  constexpr int STAGE = NP * (SA + SB);       // u16 elements: the three planes of A, then the three of B
  constexpr int SSB = (XF != VAE_XF_NONE) ? 2 * SS_HALF * 2 : 0;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AR = BM / RP;
  constexpr int BR = BKM ? (BK / (NT / (BN / 4))) : (BN / RP);
  static_assert(AR >= 1 && BR >= 1 && TM % 32 == 0 && TN % 32 == 0, "tile/wave layout");
  __shared__ __attribute__((aligned(16))) u16 smem[2 * STAGE + SSB];
  float* sS = reinterpret_cast<float*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const float* __restrict__ A = p.A + (int64_t)z * p.sAb;
  const float* __restrict__ W = p.W + (int64_t)z * p.sWb;
  const int hw = g.Ho * g.Wo;

  // stride-2 dgrad, parity-class-major rows (see igemm.hip)
  const bool s2c = (g.mode == VAE_MODE_DGRAD_S2);
  const int hh = g.Ho >> 1, wh = g.Wo >> 1;
  const int cls_rows = s2c ? p.M >> 2 : 1;
  const int cls = s2c ? m0 / cls_rows : 0;
  const int cpy = cls >> 1, cpx = cls & 1;
  const int nkw = s2c ? (cpx ? 1 : 2) : 3;
  const int ntaps = s2c ? (cpy ? 1 : 2) * nkw : g.taps;
  auto row_pixel = [&](int m, int& b, int& y, int& x) {
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

  // operands through buffer descriptors (common.h; same conventions as the fp32 flat kernel)
  const int b_base = s2c ? (m0 - cls * cls_rows) / (hh * wh) : m0 / hw;
  const size_t img = (size_t)g.Hs * g.Ws * g.Cs;
  const size_t abytes = (size_t)(g.B - b_base) * img * 4u, wbytes = (size_t)(BKM ? (int64_t)p.K * p.sk : (int64_t)p.N * p.sn) * 4u;
  const auto rsA = VAE_BUF_RSRC(A + (int64_t)b_base * img, abytes < BUF_MAX ? abytes : BUF_MAX);
  const auto rsW = VAE_BUF_RSRC(W, wbytes < BUF_MAX ? wbytes : BUF_MAX);

  const int k4 = tid % KQ, r0 = tid / KQ;
  int rb[AR], ry[AR], rx[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + r0 + RP * i;
    if (m < p.M) {
      row_pixel(m, rb[i], ry[i], rx[i]);
    } else {
      rb[i] = -1; ry[i] = 0; rx[i] = 0;
    }
  }

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
  f32x4 ra[AR], rw[BR];
  int a_b[AR];
  int reg_c0 = 0;

  auto load_regs = [&](int s) {
    const int ord = s / kchunks;
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
      const bool ok = src_pixel(g, ry[i], rx[i], kh, kw, sy, sx) && (rb[i] >= 0);
      ra[i] = VAE_BUF_LOAD4(rsA, (ok && c < p.K) ? ((unsigned)(((rb[i] - b_base) * g.Hs + sy) * g.Ws + sx) * (unsigned)g.Cs + (unsigned)c) * 4u : BUF_OOB);
      a_b[i] = ok ? rb[i] : -1;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + RP * i;
        rw[i] = VAE_BUF_LOAD4(rsW, (n < p.N && c < p.K) ? ((unsigned)n * (unsigned)p.sn + (unsigned)tap * (unsigned)p.st + (unsigned)c) * 4u : BUF_OOB);
      }
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int k = c0 + kq + KR * i;
        const int n = n0 + n4 * 4;
        rw[i] = VAE_BUF_LOAD4(rsW, (k < p.K && n < p.N) ? ((unsigned)k * (unsigned)p.sk + (unsigned)tap * (unsigned)p.st + (unsigned)n) * 4u : BUF_OOB);
      }
    }
  };
  auto store_lds = [&](u16* sA, u16* sB) {
    const int c = reg_c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = ra[i];
      if (XF != VAE_XF_NONE) {
        const bool ok = (a_b[i] >= 0) && (c < p.K);
        const int o = ok ? (a_b[i] - b_lo) * p.K + c : 0;
        v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
      }
      uint2 pl[NP];
      split3(v, pl);
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(&sA[q * SA + (r0 + RP * i) * LDA + k4 * 4]) = pl[q];
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        uint2 pl[NP];
        split3(rw[i], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(&sB[q * SB + (r0 + RP * i) * LDB + k4 * 4]) = pl[q];
      }
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        uint2 pl[NP];
        split3(rw[i], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(&sB[q * SB + (kq + KR * i) * LDB + n4 * 4]) = pl[q];
      }
    }
  };
  // a step = one k-group of 16: per wave MI + NI fragments x 3 planes feed MI * NI * 6 MFMAs (hi*hi, hi*mid, mid*hi, mid*mid,
  // hi*lo, lo*hi: the partial products down to 2^-24 of the product; what is dropped is below one fp32 ulp)
  bf16x8 fa[MI][NP], fb[NI][NP];
  auto fetch = [&](const u16* sA, const u16* sB) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi][q] = frag_direct(sA + q * SA + (wm * TM + mi * 32 + lr) * LDA + lh * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (!BKM) fb[ni][q] = frag_direct(sB + q * SB + (wn * TN + ni * 32 + lr) * LDB + lh * 8);
        else fb[ni][q] = frag_tr(sB + q * SB + (lh * 8 + trq) * LDB + wn * TN + ni * 32 + trh * 16 + trp * 4, LDB);
      }
    }
  };
  // partial product `pr` of every accumulator block: consecutive MFMAs go to different accumulators (a dependent pair is
  // four instructions apart, as in the bf16 kernels)
  auto compute = [&](int pr) {
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][PA[pr]], fb[ni][PB[pr]], acc[mi][ni], 0, 0, 0);
  };

  load_regs(0);
  __syncthreads();  // scale/shift table visible
  store_lds(smem, smem + NP * SA);
  if (steps > 1) load_regs(1);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const u16* cA = smem + (s & 1) * STAGE;
    const u16* cB = cA + NP * SA;
    fetch(cA, cB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      compute(pr);
      if (pr == 2 && s + 1 < steps) {  // staged in the shadow of the MFMAs already issued
        __builtin_amdgcn_sched_barrier(0);
        u16* nA = smem + ((s + 1) & 1) * STAGE;
        store_lds(nA, nA + NP * SA);
        if (s + 2 < steps) load_regs(s + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue (fp32) ----------------
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
    float* red = reinterpret_cast<float*>(smem);  // [WM][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float s2 = tsum[ni] + __shfl_xor(tsum[ni], 32, 64);
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

// out[m][tap][n] = sum_pix dY[pix][m] * XF(X[src(pix,tap)][n]) : both tiles pixel-major, transposing reads

template <bool BKM, int XF>
int launch_split_t(const vae_igemm_args& a, dim3 grid, hipStream_t st) {
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
  hipLaunchKernelGGL((igemm_rows_split_kernel<BM, BN, WM, WN, BKM, XF>), grid, dim3(64 * WM * WN), 0, st, a);
  return 0;
}

}  // namespace

// vectorised shapes with N > 32 only (the caller checked); 128 x 128 tiles
int launch_rows_split(const vae_igemm_args& a, bool bkm, hipStream_t st) {
  dim3 grid((unsigned)(((a.M + 127) / 128) * ((a.N + 127) / 128)), 1, (unsigned)a.batch);
  if (bkm) return launch_split_t<true, VAE_XF_NONE>(a, grid, st);
  switch (a.xf) {
    case VAE_XF_NONE: return launch_split_t<false, VAE_XF_NONE>(a, grid, st);
    case VAE_XF_AFFINE: return launch_split_t<false, VAE_XF_AFFINE>(a, grid, st);
    default: return launch_split_t<false, VAE_XF_AFFINE_SILU>(a, grid, st);
  }
}
