// bf16-compute variant of conv3_tile.hip (3x3 stride-1 convolution forward / forward over a virtual
// nearest-2x upsample / dgrad) for `training.mixed_precision: bf16`:
//   * activations, weights and outputs stay fp32 in HBM (fp32 master weights, fp32 statistics);
//   * operands are rounded to bf16 while they are staged into LDS (after the fp32 GroupNorm+SiLU transform);
//   * products run on v_mfma_f32_32x32x16_bf16 (16x the fp32-input MFMA rate), accumulation is fp32.
// Same tiling as the fp32 kernel: 4x32-pixel output tile x 128 channels per 8-wave workgroup, the 9 taps share
// one staged halo; the channel chunk is 64 (8 MFMAs per wave per barrier).  Forward reads both operands as
// 16-byte k-contiguous fragments; dgrad keeps the weight tile in its memory order ([k = co][n = ci]) and reads
// it with the transposing LDS load (ds_read_b64_tr_b16).
#include "bf16_frag.h"

namespace {


constexpr int BK = 64, TH = 4, TW = 32, HW_ = TW + 2, HP = (TH + 2) * HW_;  // 204 halo pixels
constexpr int LDH = BK + 8;                 // halo row stride in bf16 (144 B: conflict-free ds_read_b128)
constexpr int BN = 128, NT = 512;
constexpr int SH = HP * LDH;                // halo stage (bf16 elements)
constexpr int HQ = HP * (BK / 4);           // float4 slots of one halo (3264)
constexpr int HI = (HQ + NT - 1) / NT;      // 7
constexpr int LDBK = BK + 8;                // weight tile [n][k] row stride (forward)
constexpr int LDBN = BN + 32;               // weight tile [k][n] row stride (dgrad): 320 B => tr reads conflict-free


template <bool BKM, bool DG, bool UP, int XF>
__global__ __launch_bounds__(NT, 4) void conv3_tile_bf16_kernel(vae_igemm_args p, int tiles_x, int tiles_y) {
  constexpr int LDB = BKM ? LDBN : LDBK;
  constexpr int SB = BKM ? BK * LDBN : BN * LDBK;
  constexpr int SSB = (XF != VAE_XF_NONE) ? 2 * SS_HALF * 2 : 0;  // fp32 table, counted in u16 units
  __shared__ __attribute__((aligned(16))) u16 smem[SH + 2 * SB + SSB];
  u16* sH = smem;
  u16* sBst = smem + SH;
  float* sS = reinterpret_cast<float*>(smem + SH + 2 * SB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = (p.N + BN - 1) / BN;
  int t = blockIdx.x;
  const int tn = t % tilesN; t /= tilesN;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int tile_lin = blockIdx.x / tilesN;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const float* __restrict__ A = p.A;
  const float* __restrict__ W = p.W;
  const int Hb = UP ? 2 * g.Hs : g.Hs, Wb = UP ? 2 * g.Ws : g.Ws;

  if (XF != VAE_XF_NONE) {
    for (int c = tid; c < p.K; c += NT) {
      sS[c] = p.scale[(int64_t)b * g.Cs + c];
      sS[SS_HALF + c] = p.shift[(int64_t)b * g.Cs + c];
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  const int kchunks = (p.K + BK - 1) / BK;
  const int steps = 9 * kchunks;

  // ---- halo staging ----
  f32x4 rh[HI];
  int hmask = 0, hc0 = 0;
  auto load_halo = [&](int c0) {
    hc0 = c0;
    hmask = 0;
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      const int pp = q >> 4, k4 = q & 15;
      const int ir = pp / HW_, jc = pp - ir * HW_;
      const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
      const int c = c0 + k4 * 4;
      const bool ok = (q < HQ) && ((unsigned)hy < (unsigned)Hb) && ((unsigned)hx < (unsigned)Wb) && (c < p.K);
      const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
      rh[i] = load4g<true>(A + (((int64_t)b * g.Hs + sy) * g.Ws + sx) * g.Cs + c, ok, A, c, p.K);
      hmask |= (ok ? 1 : 0) << i;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      if (q < HQ) {
        f32x4 v = rh[i];
        if (XF != VAE_XF_NONE) {
          const bool ok = (hmask >> i) & 1;
          const int o = ok ? hc0 + (q & 15) * 4 : 0;
          v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
        }
        *reinterpret_cast<uint2*>(&sH[(q >> 4) * LDH + (q & 15) * 4]) = pack4(v);
      }
    }
  };

  // ---- weight staging: 128 x 64 tile, 4 float4 per thread ----
  f32x4 rw[4];
  auto load_w = [&](int s) {
    const int cch = s / 9, tap = s - cch * 9;
    const int c0 = cch * BK;
    if (!BKM) {
      const int k4 = tid & 15, r0 = tid >> 4;  // rows n = r0 + 32 i
      const int c = c0 + k4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + r0 + 32 * i;
        rw[i] = load4g<true>(W + (int64_t)n * p.sn + (int64_t)tap * p.st + c, n < p.N, W, c, p.K);
      }
    } else {
      const int n4 = tid & 31, kq = tid >> 5;  // rows k = kq + 16 i
      const int n = n0 + n4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = c0 + kq + 16 * i;
        rw[i] = load4g<true>(W + (int64_t)k * p.sk + (int64_t)tap * p.st + n, k < p.K, W, n, p.N);
      }
    }
  };
  auto store_w = [&](u16* sB) {
    if (!BKM) {
      const int k4 = tid & 15, r0 = tid >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint2*>(&sB[(r0 + 32 * i) * LDB + k4 * 4]) = pack4(rw[i]);
    } else {
      const int n4 = tid & 31, kq = tid >> 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint2*>(&sB[(kq + 16 * i) * LDB + n4 * 4]) = pack4(rw[i]);
    }
  };

  // lane's address pattern for the transposing read: group row q, column quad pp
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  auto compute = [&](const u16* sA, const u16* sB) {
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      const bf16x8 a = frag_direct(sA + kg * 16 + lh * 8);
      bf16x8 bq[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (!BKM) {
          bq[ni] = frag_direct(sB + (wn * 64 + ni * 32 + lr) * LDB + kg * 16 + lh * 8);
        } else {
          bq[ni] = frag_tr(sB + (kg * 16 + lh * 8 + trq) * LDB + wn * 64 + ni * 32 + trh * 16 + trp * 4, LDB);
        }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[ni], acc[ni], 0, 0, 0);
    }
  };

  load_halo(0);
  load_w(0);
  __syncthreads();  // scale/shift table visible
  store_halo();
  store_w(sBst);
  if (steps > 1) load_w(1);
  __syncthreads();
  int cch = 0, tap = 0;
  for (int s = 0; s < steps; ++s) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dy = DG ? 2 - kh : kh, dx = DG ? 2 - kw : kw;
    const u16* cA = sH + ((wm + dy) * HW_ + lr + dx) * LDH;
    const u16* cB = sBst + (s & 1) * SB;
    if (tap == 0 && cch + 1 < kchunks) load_halo((cch + 1) * BK);
    if (s + 1 < steps) {
      store_w(sBst + ((s + 1) & 1) * SB);
      if (s + 2 < steps) load_w(s + 2);
    }
    compute(cA, cB);
    __syncthreads();
    if (++tap == 9) {
      tap = 0;
      if (++cch < kchunks) {
        store_halo();
        __syncthreads();
      }
    }
  }

  // ---------------- epilogue (fp32) ----------------
  const int oy = y0 + wm;
  float tsum[2] = {0.f, 0.f};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + lr;
    const bool colok = col < p.N;
    const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (colok && oy < g.Ho && ox < g.Wo) {
        const int64_t o = (((int64_t)b * g.Ho + oy) * g.Wo + ox) * p.ldc + col;
        float v = p.alpha * acc[ni][r] + bv;
        if (p.res) v += p.res[o];
        p.C[o] = v;
        tsum[ni] += fabsf(v);
      }
    }
  }
  if (p.track) {
    float* red = reinterpret_cast<float*>(smem);  // [4][BN]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const float s2 = tsum[ni] + __shfl_xor(tsum[ni], 32, 64);
      if (lh == 0) red[wm * BN + wn * 64 + ni * 32 + lr] = s2;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N)
      p.track[(int64_t)tile_lin * p.N + n0 + tid] = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
  }
}

template <bool BKM, bool DG, bool UP>
void launch_xf(const vae_igemm_args& a, dim3 grid, int tx, int ty, hipStream_t st) {
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((conv3_tile_bf16_kernel<BKM, DG, UP, VAE_XF_NONE>), grid, dim3(NT), 0, st, a, tx, ty); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((conv3_tile_bf16_kernel<BKM, DG, UP, VAE_XF_AFFINE>), grid, dim3(NT), 0, st, a, tx, ty); break;
    default: hipLaunchKernelGGL((conv3_tile_bf16_kernel<BKM, DG, UP, VAE_XF_AFFINE_SILU>), grid, dim3(NT), 0, st, a, tx, ty); break;
  }
}

}  // namespace

int launch_conv3_tile_bf16(const vae_igemm_args& a, bool bkm, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / TW, ty = g.Ho / TH;
  const int64_t nblk = (int64_t)((a.N + BN - 1) / BN) * tx * ty * g.B;
  if (nblk > 0x7fffffffLL) return VAE_EINVAL;
  dim3 grid((unsigned)nblk);
  if (g.mode == VAE_MODE_DGRAD) launch_xf<true, true, false>(a, grid, tx, ty, st);
  else if (g.mode == VAE_MODE_UP2X) launch_xf<false, false, true>(a, grid, tx, ty, st);
  else launch_xf<false, false, false>(a, grid, tx, ty, st);
  return 0;
}
