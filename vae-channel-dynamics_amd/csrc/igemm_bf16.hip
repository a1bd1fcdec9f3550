// bf16-compute variants of the flat implicit-GEMM kernels of igemm.hip (VAE_PREC_BF16): every geometry the
// fp32 flat kernels serve (stride-2 convs, 1x1, parity-class dgrad, tiny spatial sizes, attention linears and
// batched GEMMs, skinny layers): operands rounded to bf16 while staged in LDS, products on v_mfma_f32_32x32x16_bf16, fp32
// accumulation.  Every activation operand is stored as fp32 or as bf16, per tensor (a_bf16 / out_bf16 / res_bf16, x_bf16 /
// y_bf16: wave-uniform switches on the load / store instructions; the weights W are always the fp32 master copy).  Only the vectorised (16-B aligned, channel counts % 4 == 0)
// shapes exist here; the rest stays on the fp32 kernels.
//   rows : A tile [BM rows][BK k] k-contiguous (one ds_read_b128 per fragment);
//          weight tile k-contiguous ([BN][BK], ds_read_b128) or n-contiguous ([BK][BN], transposing read)
//   wgrad: both tiles pixel-major ([BK px][BM], [BK px][BN]), both fragments through the transposing read
#include "bf16_frag.h"

namespace {

template <int BM, int BN, int WM, int WN, bool BKM, int XF>
__global__ __launch_bounds__(64 * WM * WN) void igemm_rows_bf16_kernel(vae_igemm_args p) {
  constexpr int BK = 64;
  constexpr int NT = 64 * WM * WN;
  constexpr int RP = NT / 16;                 // tile rows per loader pass (16 float4 per 64-channel row)
  constexpr int LDA = BK + 8;                 // 144 B rows: conflict-free ds_read_b128
  constexpr int LDB = BKM ? (BN + 32) : (BK + 8);
  constexpr int SA = BM * LDA;
  constexpr int SB = BKM ? BK * LDB : BN * LDB;
  constexpr int STAGE = SA + SB;              // u16 elements
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
  const SrcMap smap = make_srcmap(g);
  const bool abf = p.a_bf16 != 0, cbf = p.out_bf16 != 0, rbf = p.res_bf16 != 0;  // storage of A / C / res (uniform)
  const unsigned esA = abf ? 2u : 4u;
  const char* __restrict__ A = reinterpret_cast<const char*>(p.A) + (int64_t)z * p.sAb * esA;
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
  const size_t abytes = (size_t)(g.B - b_base) * img * esA, wbytes = (size_t)(BKM ? (int64_t)p.K * p.sk : (int64_t)p.N * p.sn) * 4u;
  const auto rsA = VAE_BUF_RSRC(A + (int64_t)b_base * img * esA, abytes < BUF_MAX ? abytes : BUF_MAX);
  const auto rsW = VAE_BUF_RSRC(W, wbytes < BUF_MAX ? wbytes : BUF_MAX);

  const int k4 = tid & 15, r0 = tid >> 4;
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
  uint4 ra[AR];  // A as loaded (fp32 quad, or 4 bf16 in the low half): converted at the LDS write
  f32x4 rw[BR];
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
      const bool ok = src_pixel(smap, ry[i], rx[i], kh, kw, sy, sx) && (rb[i] >= 0);
      ra[i] = buf_load4_raw(rsA, esA, neg_unless(ok && c < p.K, (((rb[i] - b_base) * g.Hs + sy) * g.Ws + sx) * g.Cs + c));
      a_b[i] = ok ? rb[i] : -1;
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + RP * i;
        rw[i] = VAE_BUF_LOAD4(rsW, oob_unless(n < p.N && c < p.K, ((unsigned)n * (unsigned)p.sn + (unsigned)tap * (unsigned)p.st + (unsigned)c) * 4u));
      }
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int k = c0 + kq + KR * i;
        const int n = n0 + n4 * 4;
        rw[i] = VAE_BUF_LOAD4(rsW, oob_unless(k < p.K && n < p.N, ((unsigned)k * (unsigned)p.sk + (unsigned)tap * (unsigned)p.st + (unsigned)n) * 4u));
      }
    }
  };
  auto store_lds = [&](u16* sA, u16* sB) {
    const int c = reg_c0 + k4 * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = raw4_to_f32(ra[i], abf);
      if (XF != VAE_XF_NONE) {
        const bool ok = (a_b[i] >= 0) && (c < p.K);
        const int o = ok ? (a_b[i] - b_lo) * p.K + c : 0;
        v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
      }
      *reinterpret_cast<uint2*>(&sA[(r0 + RP * i) * LDA + k4 * 4]) = pack4(v);
    }
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<uint2*>(&sB[(r0 + RP * i) * LDB + k4 * 4]) = pack4(rw[i]);
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<uint2*>(&sB[(kq + KR * i) * LDB + n4 * 4]) = pack4(rw[i]);
    }
  };
  // fragments of k-group kg+1 are requested before the MFMAs of kg (pinned with sched_barrier; see conv3_tile_bf16.hip)
  bf16x8 fa[2][MI], fb[2][NI];
  auto fetch = [&](const u16* sA, const u16* sB, int kg, bf16x8* a, bf16x8* b) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = frag_direct(sA + (wm * TM + mi * 32 + lr) * LDA + kg * 16 + lh * 8);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (!BKM) b[ni] = frag_direct(sB + (wn * TN + ni * 32 + lr) * LDB + kg * 16 + lh * 8);
      else b[ni] = frag_tr(sB + (kg * 16 + lh * 8 + trq) * LDB + wn * TN + ni * 32 + trh * 16 + trp * 4, LDB);
    }
  };
  auto compute = [&](const u16* sA, const u16* sB, int kg) {  // fragments of kg already requested
    if (kg + 1 < BK / 16) fetch(sA, sB, kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kg & 1][mi], fb[kg & 1][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  load_regs(0);
  __syncthreads();  // scale/shift table visible
  store_lds(smem, smem + SA);
  if (steps > 1) load_regs(1);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const u16* cA = smem + (s & 1) * STAGE;
    const u16* cB = cA + SA;
    fetch(cA, cB, 0, fa[0], fb[0]);
    compute(cA, cB, 0);
    compute(cA, cB, 1);
    if (s + 1 < steps) {
      u16* nA = smem + ((s + 1) & 1) * STAGE;
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
  const unsigned esC = cbf ? 2u : 4u, esR = rbf ? 2u : 4u;
  char* __restrict__ C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.sCb * esC;
  const auto rsC = VAE_BUF_RSRC(C, (size_t)p.M * p.ldc * esC);
  const auto rsR = VAE_BUF_RSRC(p.res ? reinterpret_cast<const char*>(p.res) + (int64_t)z * p.sCb * esR : C, (size_t)p.M * p.ldc * (p.res ? esR : esC));
  float tsum[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) tsum[ni] = 0.f;
  // bf16 output with an even column count (and a bf16 residual or none): adjacent lanes hold adjacent columns of the same 16
  // rows; they swap every other register, so a lane stores BOTH columns of its pair for 8 rows -- 4-byte stores and 4-byte
  // residual loads instead of 2-byte ones (the 2-byte form cost the flat kernels 20 % when bf16 storage came in)
  const bool pair16 = cbf && (p.N % 2 == 0) && (p.ldc % 2 == 0) && (p.res == nullptr || rbf) && p.track == nullptr &&
                      ((reinterpret_cast<uintptr_t>(C) & 3u) == 0);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * TN + ni * 32 + lr;
    const bool colok = col < p.N;
    const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
    if (pair16) {  // uniform
      const bool odd = lr & 1;
      const float b0 = (p.bias && colok) ? p.bias[col & ~1] : 0.f, b1 = (p.bias && colok) ? p.bias[col | 1] : 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        unsigned o16[8], rr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 2 * j + (odd ? 1 : 0);
          const int row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          int orow = row;
          if (s2c) {
            int b, y, x;
            row_pixel(row < p.M ? row : m0, b, y, x);
            orow = (b * g.Ho + y) * g.Wo + x;
          }
          o16[j] = (colok && row < p.M) ? (unsigned)(orow * p.ldc + (col & ~1)) * 2u : BUF_OOB;
          rr[j] = 0u;
        }
        if (p.res) {  // uniform
#pragma unroll
          for (int j = 0; j < 8; ++j) rr[j] = __builtin_amdgcn_raw_buffer_load_b32(rsR, o16[j], 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a0 = p.alpha * acc[mi][ni][2 * j], a1 = p.alpha * acc[mi][ni][2 * j + 1];
          const float recv = lane_xor1(odd ? a0 : a1);
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          bf16x2_t h;
          h[0] = (__bf16)((odd ? recv : a0) + b0 + __builtin_bit_cast(float, rr[j] << 16));
          h[1] = (__bf16)((odd ? a1 : recv) + b1 + __builtin_bit_cast(float, rr[j] & 0xffff0000u));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rsC, o16[j], 0, 0);
        }
      }
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int off[16];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        int orow = row;
        if (s2c) {  // class-major row -> pixel-major output row
          int b, y, x;
          row_pixel(row < p.M ? row : m0, b, y, x);
          orow = (b * g.Ho + y) * g.Wo + x;
        }
        off[r] = (colok && row < p.M) ? orow * p.ldc + col : -1;  // element offset
        rv[r] = 0.f;
      }
      if (p.res) {  // uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = buf_load1_elem(rsR, rbf, off[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = p.alpha * acc[mi][ni][r] + bv + rv[r];
        buf_store1_elem(rsC, cbf, off[r], v);
        tsum[ni] += (off[r] >= 0) ? fabsf(v) : 0.f;
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
template <int BM, int BN, int WM, int WN, int XF>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_bf16_kernel(vae_wgrad_args p) {
  constexpr int BK = 32;                       // pixels per step
  constexpr int NT = 64 * WM * WN;
  constexpr int LDA = BM + 32, LDB = BN + 32;  // 64 B mod 256 B row strides: conflict-free transposing reads
  constexpr int SA = BK * LDA, SB = BK * LDB;
  constexpr int STAGE = SA + SB;
  constexpr int SSB = (XF != VAE_XF_NONE) ? 2 * SS_HALF * 2 : 0;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AQ = BM / 4, AKR = NT / AQ, AI = BK / AKR;
  constexpr int BQ = BN / 4, BKR = NT / BQ, BI = BK / BKR;
  static_assert(AI >= 1 && BI >= 1 && TM % 32 == 0 && TN % 32 == 0, "tile/wave layout");
  __shared__ __attribute__((aligned(16))) u16 smem[2 * STAGE + SSB];
  float* sS = reinterpret_cast<float*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tap = blockIdx.y / p.nsplit, split = blockIdx.y % p.nsplit;
  const int z = blockIdx.z;
  const vae_conv_geom g = p.g;
  const SrcMap smap = make_srcmap(g);
  const int kh = (g.taps == 9) ? tap / 3 : 0, kw = (g.taps == 9) ? tap - kh * 3 : 0;
  const bool ybf = p.y_bf16 != 0, xbf = p.x_bf16 != 0;  // storage of dY / X (uniform)
  const unsigned esY = ybf ? 2u : 4u, esX = xbf ? 2u : 4u;
  const char* __restrict__ dY = reinterpret_cast<const char*>(p.dY) + (int64_t)z * p.sYb * esY;
  const char* __restrict__ X = reinterpret_cast<const char*>(p.X) + (int64_t)z * p.sXb * esX;

  int chunk = (p.npix + p.nsplit - 1) / p.nsplit;
  chunk = ((chunk + 31) / 32) * 32;  // same rounding as the fp32 kernel (the table-fit check assumes it)
  const int pbeg = split * chunk;
  const int pend = min(p.npix, pbeg + chunk);
  const int steps = (pend > pbeg) ? (pend - pbeg + BK - 1) / BK : 0;
  const int hw = g.Ho * g.Wo;
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0 && tap == 0 && z == 0;

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

  const size_t img = (size_t)g.Hs * g.Ws * g.Cs;
  const size_t ybytes = (size_t)(steps > 0 ? pend - pbeg : 0) * p.ldy * esY, xbytes = (size_t)(g.B - b_lo) * img * esX;
  const auto rsY = VAE_BUF_RSRC(dY + (int64_t)pbeg * p.ldy * esY, ybytes < BUF_MAX ? ybytes : BUF_MAX);
  const auto rsX = VAE_BUF_RSRC(X + (int64_t)b_lo * img * esX, xbytes < BUF_MAX ? xbytes : BUF_MAX);

  const int a4 = tid % AQ, akq = tid / AQ;
  const int b4 = tid % BQ, bkq = tid / BQ;
  uint4 ra[AI], rx[BI];  // as loaded: converted at the LDS write
  int xb[BI];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  auto load_regs = [&](int s) {
    const int pb = pbeg + s * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int pix = pb + akq + AKR * i;
      const int c = m0 + a4 * 4;
      ra[i] = buf_load4_raw(rsY, esY, neg_unless(pix < pend && c < p.M, (pix - pbeg) * p.ldy + c));
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int pix = pb + bkq + BKR * i;
      const int c = n0 + b4 * 4;
      const int b = pix / hw, rem = pix - b * hw;
      const int y = rem / g.Wo, x = rem - y * g.Wo;
      int sy = 0, sx = 0;
      const bool ok = src_pixel(smap, y, x, kh, kw, sy, sx) && (pix < pend);
      rx[i] = buf_load4_raw(rsX, esX, neg_unless(ok && c < p.N, (((b - b_lo) * g.Hs + sy) * g.Ws + sx) * g.Cs + c));
      xb[i] = ok ? b : -1;
    }
  };
  auto store_lds = [&](u16* sA, u16* sB) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const f32x4 v = raw4_to_f32(ra[i], ybf);
      *reinterpret_cast<uint2*>(&sA[(akq + AKR * i) * LDA + a4 * 4]) = ybf ? uint2{ra[i].x, ra[i].y} : pack4(v);
      if (do_bias) bsum += v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      f32x4 v = raw4_to_f32(rx[i], xbf);
      if (XF != VAE_XF_NONE) {
        const bool ok = xb[i] >= 0;
        const int o = ok ? (xb[i] - b_lo) * BN + b4 * 4 : 0;
        v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
      }
      *reinterpret_cast<uint2*>(&sB[(bkq + BKR * i) * LDB + b4 * 4]) = pack4(v);
    }
  };
  bf16x8 fa[2][MI], fb[2][NI];
  auto fetch = [&](const u16* sA, const u16* sB, int kg, bf16x8* a, bf16x8* b) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = frag_tr(sA + (kg * 16 + lh * 8 + trq) * LDA + wm * TM + mi * 32 + trh * 16 + trp * 4, LDA);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = frag_tr(sB + (kg * 16 + lh * 8 + trq) * LDB + wn * TN + ni * 32 + trh * 16 + trp * 4, LDB);
  };
  auto compute = [&](const u16* sA, const u16* sB, int kg) {  // fragments of kg already requested
    if (kg + 1 < BK / 16) fetch(sA, sB, kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kg & 1][mi], fb[kg & 1][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (steps > 0) {
    load_regs(0);
    __syncthreads();
    store_lds(smem, smem + SA);
    if (steps > 1) load_regs(1);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const u16* cA = smem + (s & 1) * STAGE;
      const u16* cB = cA + SA;
      fetch(cA, cB, 0, fa[0], fb[0]);
      compute(cA, cB, 0);
      if (s + 1 < steps) {
        u16* nA = smem + ((s + 1) & 1) * STAGE;
        store_lds(nA, nA + SA);
        if (s + 2 < steps) load_regs(s + 2);
      }
      compute(cA, cB, 1);
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
  if (do_bias) {
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

template <int BM, int BN, int WM, int WN, bool BKM>
void launch_rows_xf(const vae_igemm_args& a, dim3 grid, hipStream_t st) {
  if (BKM) {
    hipLaunchKernelGGL((igemm_rows_bf16_kernel<BM, BN, WM, WN, BKM, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a);
    return;
  }
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((igemm_rows_bf16_kernel<BM, BN, WM, WN, false, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((igemm_rows_bf16_kernel<BM, BN, WM, WN, false, VAE_XF_AFFINE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    default: hipLaunchKernelGGL((igemm_rows_bf16_kernel<BM, BN, WM, WN, false, VAE_XF_AFFINE_SILU>), grid, dim3(64 * WM * WN), 0, st, a); break;
  }
}
template <int BM, int BN, int WM, int WN>
void launch_wgrad_xf(const vae_wgrad_args& a, dim3 grid, hipStream_t st) {
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, WM, WN, VAE_XF_NONE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, WM, WN, VAE_XF_AFFINE>), grid, dim3(64 * WM * WN), 0, st, a); break;
    default: hipLaunchKernelGGL((wgrad_bf16_kernel<BM, BN, WM, WN, VAE_XF_AFFINE_SILU>), grid, dim3(64 * WM * WN), 0, st, a); break;
  }
}

}  // namespace

// vectorised shapes only (the caller checked `vec`); same tile selection as the fp32 flat kernels
int launch_rows_bf16(const vae_igemm_args& a, bool bkm, hipStream_t st) {
  if (a.N <= 32) {
    dim3 grid((unsigned)(((a.M + 127) / 128) * ((a.N + 31) / 32)), 1, (unsigned)a.batch);
    if (bkm) launch_rows_xf<128, 32, 4, 1, true>(a, grid, st); else launch_rows_xf<128, 32, 4, 1, false>(a, grid, st);
  } else {
    dim3 grid((unsigned)(((a.M + 127) / 128) * ((a.N + 127) / 128)), 1, (unsigned)a.batch);
    if (bkm) launch_rows_xf<128, 128, 4, 2, true>(a, grid, st); else launch_rows_xf<128, 128, 4, 2, false>(a, grid, st);
  }
  return 0;
}
int launch_wgrad_bf16(const vae_wgrad_args& a, hipStream_t st) {
  const unsigned gy = (unsigned)(a.g.taps * a.nsplit), gz = (unsigned)a.batch;
  if (a.M <= 32) launch_wgrad_xf<32, 128, 1, 4>(a, dim3((unsigned)(((a.M + 31) / 32) * ((a.N + 127) / 128)), gy, gz), st);
  else if (a.N <= 32) launch_wgrad_xf<128, 32, 4, 1>(a, dim3((unsigned)(((a.M + 127) / 128) * ((a.N + 31) / 32)), gy, gz), st);
  else launch_wgrad_xf<128, 128, 4, 2>(a, dim3((unsigned)(((a.M + 127) / 128) * ((a.N + 127) / 128)), gy, gz), st);
  return 0;
}
