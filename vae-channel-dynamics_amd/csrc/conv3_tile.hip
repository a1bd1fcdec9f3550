// 3x3 stride-1 convolution (forward, forward over a virtual nearest-2x upsample, and dgrad) as a
// direct convolution over an LDS-staged input HALO tile -- no im2col, and the nine filter taps share one
// staged tile.
//
// Workgroup = 8 waves (512 threads) = one 4x32-pixel output tile x 128 output channels of one image.
// Per 32-channel input chunk the (4+2)x(32+2)-pixel halo is staged ONCE (GroupNorm+SiLU applied in that
// write pass, i.e. 1.6 transforms per input element instead of 9), then the 9 taps are 9 K-steps that only
// re-stage the 128x32 weight tile (double buffered, one barrier per step) and read the SAME halo at a
// shifted offset.  Wave (wm, wn) owns output row wm of the tile (32 consecutive pixels = one conflict-free
// ds_read_b128 fragment per k-group) and 64 channels.  Compared with the flat implicit GEMM
// (igemm.hip) this removes 8/9 of the activation loads, address arithmetic and transforms.
#include "common.h"
#include <algorithm>

namespace {

constexpr int BK = 32, TH = 4, TW = 32, HW_ = TW + 2, HP = (TH + 2) * HW_;  // 204 halo pixels
constexpr int LDA = BK + 4, BN = 128, NT = 512;
constexpr int SH = HP * LDA;            // halo stage (floats)
constexpr int HQ = HP * (BK / 4);       // float4 slots of the halo stage (1632)
constexpr int HI = (HQ + NT - 1) / NT;  // halo float4 per thread (4)

template <bool BKM, bool DG, bool UP, int XF>
__global__ __launch_bounds__(NT) void conv3_tile_kernel(vae_igemm_args p, int tiles_x, int tiles_y) {
  constexpr int LDB = BKM ? (BN + 4) : LDA;
  constexpr int SB = BKM ? BK * LDB : BN * LDB;
  constexpr int SS = (XF != VAE_XF_NONE) ? 2 * SS_HALF : 0;
  constexpr int RP = NT / 8;                               // 64 weight rows per loader pass
  constexpr int BR = BKM ? (BK / (NT / (BN / 4))) : (BN / RP);  // 2
  __shared__ __attribute__((aligned(16))) float smem[SH + 2 * SB + SS];
  float* sH = smem;
  float* sBst = smem + SH;
  float* sS = smem + SH + 2 * SB;

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
  // operand streams through buffer descriptors (common.h): the activation descriptor covers this tile's image
  // sub-sampled views (phase convolutions, vaehip.h): pixel (y,x) of A is (y*as+aoy, x*as+aox) of a tensor `as` times larger
  const int as = p.a_step > 1 ? p.a_step : 1, cs = p.c_step > 1 ? p.c_step : 1;
  const size_t abytes = (size_t)(g.Hs * as) * (g.Ws * as) * g.Cs * 4u;
  const auto rsA = VAE_BUF_RSRC(p.A + (int64_t)b * (g.Hs * as) * (g.Ws * as) * g.Cs, abytes);
  // taps to compute, packed 4 bits each (all 9 unless tapmask selects a subset)
  unsigned long long taplist = 0;
  int ntaps = 0;
  for (int tp = 0; tp < 9; ++tp)
    if (p.tapmask == 0 || ((p.tapmask >> tp) & 1)) taplist |= (unsigned long long)tp << (4 * ntaps++);
  const auto rsW = VAE_BUF_RSRC(p.W, (size_t)(BKM ? p.K * p.sk : p.N * p.sn) * 4u);
  const int Hb = UP ? 2 * g.Hs : g.Hs, Wb = UP ? 2 * g.Ws : g.Ws;  // bounds of the (virtual) source grid

  if (XF != VAE_XF_NONE) {
    for (int c = tid; c < p.K; c += NT) {  // K <= SS_HALF checked by the launcher; one image per tile
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
  const int steps = ntaps * kchunks;

  // ---- halo staging: thread owns float4 slots q = tid + NT*i ----
  f32x4 rh[HI];
  int hmask = 0, hc0 = 0;
  auto load_halo = [&](int c0) {
    hc0 = c0;
    hmask = 0;
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      const int pp = q >> 3, k4 = q & 7;
      const int ir = pp / HW_, jc = pp - ir * HW_;
      const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
      const int c = c0 + k4 * 4;
      const bool ok = (q < HQ) && ((unsigned)hy < (unsigned)Hb) && ((unsigned)hx < (unsigned)Wb) && (c < p.K);
      const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
      rh[i] = VAE_BUF_LOAD4(rsA, ok ? (unsigned)((((sy * as + p.a_oy) * (g.Ws * as) + sx * as + p.a_ox) * g.Cs + c) * 4) : BUF_OOB);
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
          const int o = ok ? hc0 + (q & 7) * 4 : 0;
          v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
        }
        *reinterpret_cast<f32x4*>(&sH[(q >> 3) * LDA + (q & 7) * 4]) = v;
      }
    }
  };

  // ---- weight staging (same tile formats as igemm.hip) ----
  f32x4 rw[BR];
  const int k4w = tid & 7, r0w = tid >> 3;
  auto load_w = [&](int s) {
    const int cch = s / ntaps, tap = (int)((taplist >> (4 * (s - cch * ntaps))) & 15);
    const int c0 = cch * BK;
    if (!BKM) {
      const int c = c0 + k4w * 4;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0w + RP * i;
        rw[i] = VAE_BUF_LOAD4(rsW, (n < p.N && c < p.K) ? (unsigned)((n * (int)p.sn + tap * (int)p.st + c) * 4) : BUF_OOB);
      }
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int k = c0 + kq + KR * i;
        const int n = n0 + n4 * 4;
        rw[i] = VAE_BUF_LOAD4(rsW, (k < p.K && n < p.N) ? (unsigned)((k * (int)p.sk + tap * (int)p.st + n) * 4) : BUF_OOB);
      }
    }
  };
  auto store_w = [&](float* sB) {
    if (!BKM) {
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(r0w + RP * i) * LDB + k4w * 4]) = rw[i];
    } else {
      constexpr int NQ = BN / 4, KR = NT / NQ;
      const int n4 = tid % NQ, kq = tid / NQ;
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&sB[(kq + KR * i) * LDB + n4 * 4]) = rw[i];
    }
  };

  // k-group kk = 8 channels: one halo fragment and two weight fragments feed 8 MFMAs.  The fragments of k-group kk+1
  // are requested before the MFMAs of kk are issued (pinned with sched_barrier; hipcc otherwise puts the reads right in
  // front of their MFMAs and waits for them there).
  f32x4 fa[2], fb[2][2];
  auto fetch = [&](const float* sA, const float* sB, int kk, f32x4& a, f32x4* bq) {
    a = *reinterpret_cast<const f32x4*>(&sA[kk * 8 + lh * 4]);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      if (!BKM) {
        bq[ni] = *reinterpret_cast<const f32x4*>(&sB[(wn * 64 + ni * 32 + lr) * LDB + kk * 8 + lh * 4]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[ni][j] = sB[(kk * 8 + lh * 4 + j) * LDB + wn * 64 + ni * 32 + lr];
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
      for (int ni = 0; ni < 2; ++ni)
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][j], fb[kk & 1][ni][j], acc[ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  load_halo(0);
  load_w(0);
  __syncthreads();  // scale/shift table visible
  store_halo();
  store_w(sBst);
  if (steps > 1) load_w(1);
  __syncthreads();
  int cch = 0, ti = 0;
  for (int s = 0; s < steps; ++s) {
    const int tap = (int)((taplist >> (4 * ti)) & 15);
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dy = DG ? 2 - kh : kh, dx = DG ? 2 - kw : kw;
    const float* cA = sH + ((wm + dy) * HW_ + lr + dx) * LDA;
    const float* cB = sBst + (s & 1) * SB;
    if (ti == 0 && cch + 1 < kchunks) load_halo((cch + 1) * BK);  // lands during this chunk's taps
    fetch(cA, cB, 0, fa[0], fb[0]);
    compute(cA, cB, 0);
    compute(cA, cB, 1);
    if (s + 1 < steps) {
      store_w(sBst + ((s + 1) & 1) * SB);
      if (s + 2 < steps) load_w(s + 2);
    }
    compute(cA, cB, 2);
    compute(cA, cB, 3);
    __syncthreads();
    if (++ti == ntaps) {
      ti = 0;
      if (++cch < kchunks) {  // every wave has left the old halo (barrier above): restage it for the next chunk
        store_halo();
        __syncthreads();
      }
    }
  }

  // ---------------- epilogue ----------------
  // outputs and the residual through buffer descriptors over this tile's image: out-of-range = load 0 / store dropped,
  // so the 16 residual loads of a block are issued back to back and there is no branch per element
  const int oy = y0 + wm;
  const size_t obytes = (size_t)(g.Ho * cs) * (g.Wo * cs) * p.ldc * 4u;
  const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
  const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
  float tsum[2] = {0.f, 0.f}, gs1[2] = {0.f, 0.f}, gs2[2] = {0.f, 0.f}, gpv[2] = {0.f, 0.f};  // shifted sums around the lane's first value
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + lr;
    const bool colok = col < p.N && oy < g.Ho;
    const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
    unsigned off[16];
    float rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      off[r] = (colok && ox < g.Wo) ? (unsigned)((((oy * cs + p.c_oy) * (g.Wo * cs) + ox * cs + p.c_ox) * p.ldc + col) * 4) : BUF_OOB;
      rv[r] = 0.f;
    }
    if (p.res) {  // uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off[r], 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = p.alpha * acc[ni][r] + bv + rv[r];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off[r], 0, 0);
      tsum[ni] += (off[r] != BUF_OOB) ? fabsf(v) : 0.f;
      if (r == 0) gpv[ni] = v;
      const float dv = v - gpv[ni];  // the statistics epilogue only runs on full tiles (every element valid)
      gs1[ni] += dv;
      gs2[ni] += dv * dv;
    }
  }
  if (p.gstat) {  // uniform: centred moments (mean, M2) of this tile's outputs per group (layout of vae_gn_stats_partial)
    const int cpg = p.N / p.gstat_groups, gpt = BN / cpg;  // channels per group (4, 8 or 16), groups per 128-channel tile
    float* red2 = smem + 4 * BN;                           // [4 rows][gpt][2], behind the tracker scratch
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const MeanM2 a = mm2_wave_group(mm2_from_shifted(gpv[ni], gs1[ni], gs2[ni], 16.f), cpg, 16.f);
      if (lh == 0 && (lr & (cpg - 1)) == 0) {
        const int gl = (wn * 64 + ni * 32 + lr) / cpg;
        red2[(wm * gpt + gl) * 2] = a.m;
        red2[(wm * gpt + gl) * 2 + 1] = a.M2;
      }
    }
    __syncthreads();
    if (tid < gpt) {  // the 4 rows of the tile, fixed order; each holds 32 pixels x cpg channels
      const float nrow = 32.f * (float)cpg;
      MeanM2 a{red2[tid * 2], red2[tid * 2 + 1]};
#pragma unroll
      for (int rr = 1; rr < 4; ++rr) a = mm2_merge(a, nrow * (float)rr, MeanM2{red2[(rr * gpt + tid) * 2], red2[(rr * gpt + tid) * 2 + 1]}, nrow);
      float* o = p.gstat + (((int64_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * p.gstat_groups + n0 / cpg + tid) * 2;
      o[0] = a.m;
      o[1] = a.M2;
    }
    __syncthreads();
  }
  if (p.track) {
    float* red = smem;  // [4][BN]; the loop's last barrier separates this from the MFMA-phase reads
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
    case VAE_XF_NONE: hipLaunchKernelGGL((conv3_tile_kernel<BKM, DG, UP, VAE_XF_NONE>), grid, dim3(NT), 0, st, a, tx, ty); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((conv3_tile_kernel<BKM, DG, UP, VAE_XF_AFFINE>), grid, dim3(NT), 0, st, a, tx, ty); break;
    default: hipLaunchKernelGGL((conv3_tile_kernel<BKM, DG, UP, VAE_XF_AFFINE_SILU>), grid, dim3(NT), 0, st, a, tx, ty); break;
  }
}

}  // namespace

// geometry this kernel covers; everything else stays on the flat implicit GEMM
bool conv3_tile_eligible(const vae_igemm_args& a, bool vec, bool bkm) {
  const vae_conv_geom& g = a.g;
  if (!vec || a.batch != 1 || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.N <= 32 || a.K % 4 != 0 || a.alpha != 1.0f) return false;
  const size_t as = a.a_step > 1 ? a.a_step : 1, cs = a.c_step > 1 ? a.c_step : 1;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u * as * as >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u * cs * cs >= BUF_MAX) return false;  // one image per descriptor
  if ((size_t)std::max((int64_t)a.K * a.sk, (int64_t)a.N * a.sn) * 4u >= BUF_MAX) return false;
  if (g.Wo % TW != 0 || g.Ho % TH != 0) return false;
  if (a.xf != VAE_XF_NONE && (a.K > SS_HALF || bkm)) return false;
  if (g.mode == VAE_MODE_FWD) return g.Ho == g.Hs && g.Wo == g.Ws && !bkm;
  if (g.mode == VAE_MODE_UP2X) return g.Ho == 2 * g.Hs && g.Wo == 2 * g.Ws && !bkm;
  if (g.mode == VAE_MODE_DGRAD) return g.Ho == g.Hs && g.Wo == g.Ws && bkm && a.xf == VAE_XF_NONE;
  return false;
}

// chunks per image of the statistics epilogue (0 = not available for these arguments)
int conv3_tile_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % BN != 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD || a.c_step > 1) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / TW) * (g.Ho / TH);
}

int launch_conv3_tile(const vae_igemm_args& a, bool bkm, hipStream_t st) {
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
