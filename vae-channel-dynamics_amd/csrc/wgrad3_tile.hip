// Weight gradient of a 3x3 stride-1 convolution (plain or over a virtual nearest-2x upsample) with the nine
// taps sharing one staged tile.
//
//   dW[co][tap][ci] = sum_{b,y,x} dY[b][y][x][co] * XF(X)[b][y+kh-1][x+kw-1][ci]
//
// Workgroup (12 waves) = 128 output channels (co) x 32 input channels (ci) x ALL 9 taps, over a range of
// "units" (one unit = a 2x32-pixel output tile).  Per unit it stages dY[64 px][128 co] and the 4x34-pixel
// halo of X (GroupNorm+SiLU applied while staging): 52 KB per 1152 MFMAs, against 32 KB per 256 MFMAs for
// the flat kernel (igemm.hip wgrad_kernel), and the activation is transformed 2.1x instead of 9x per co-tile.  Wave w owns co rows (w&3)*32.. and the three taps of filter row kh = w>>2: its dY
// fragment is read once per 8-pixel group and reused for its 3 taps (3 accumulators = 48 VGPRs).  Both LDS tiles are pixel-major, so a
// fragment read is one conflict-free ds_read_b32 per k.  LDS is double buffered, one barrier per unit.
// Split-K over unit ranges; partial slabs are summed in fixed order by vae_reduce_splits (deterministic).
#include "common.h"

namespace {

constexpr int TH = 2, TW = 32;          // unit = TH x TW output pixels
constexpr int UPX = TH * TW;            // pixels per unit (64)
constexpr int BMT = 128, BNT = 32;      // co tile, ci tile
constexpr int LDA = BMT + 4;            // dY stage row stride (floats)
constexpr int HWD = TW + 2;             // halo width (34)
constexpr int HPX = (TH + 2) * HWD;     // halo pixels (136)
constexpr int LDH = BNT + 4;            // halo stage row stride
constexpr int SA = UPX * LDA;           // 4224 floats
constexpr int SH = HPX * LDH;           // 3672 floats
constexpr int STAGE = SA + SH;
constexpr int NT = 768;                                    // 12 waves: 4 co sub-tiles x 3 filter rows
constexpr int AQ = UPX * (BMT / 4);                        // dY float4 slots (2048)
constexpr int AI = (AQ + NT - 1) / NT;                     // 2
constexpr int HQ = HPX * (BNT / 4);                        // halo float4 slots (1088)
constexpr int HI = (HQ + NT - 1) / NT;                     // 2

template <bool UP, int XF>
__global__ __launch_bounds__(NT) void wgrad3_tile_kernel(vae_wgrad_args p, int xblocks, int64_t nunits) {
  // unit u -> (image b, tile row ty = 0..Ho/TH-1, x block xb)
  constexpr int SS = (XF != VAE_XF_NONE) ? 2 * SS_HALF : 0;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + SS];
  float* sS = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int mt = wave & 3, tg = wave >> 2;  // co sub-tile, filter row kh
  const vae_conv_geom g = p.g;
  const int tilesN = p.N / BNT;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int m0 = tm * BMT, n0 = tn * BNT;
  const int split = blockIdx.y;
  const int64_t per = (nunits + p.nsplit - 1) / p.nsplit;
  const int64_t ubeg = split * per, uend = min(nunits, ubeg + per);
  const int nu = (int)max((int64_t)0, uend - ubeg);
  const int Hb = UP ? 2 * g.Hs : g.Hs, Wb = UP ? 2 * g.Ws : g.Ws;
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0;
  // phase convolutions (vaehip.h): dY is a sub-sampled view, and only the taps of tapmask are computed -- a wave whose kernel
  // row is masked out only helps with the staging, the others skip the masked columns
  const int ys = p.y_step > 1 ? p.y_step : 1;
  const int tmask = p.tapmask ? p.tapmask : 0x1ff;
  const int wmask = (tmask >> (3 * tg)) & 7;  // this wave's kernel row: bit t = tap (kh = tg, kw = t)
  const int rows_per_img = (g.Ho / TH) * xblocks;  // units per image

  const int b_lo = nu > 0 ? (int)(ubeg / rows_per_img) : 0;
  if (XF != VAE_XF_NONE && nu > 0) {
    const int nb = (int)((uend - 1) / rows_per_img) - b_lo + 1;
    const int nent = min(nb * BNT, SS_HALF);
    for (int i = tid; i < nent; i += NT) {
      const int j = i / BNT, c = i - j * BNT;
      sS[i] = p.scale[(int64_t)(b_lo + j) * g.Cs + n0 + c];
      sS[SS_HALF + i] = p.shift[(int64_t)(b_lo + j) * g.Cs + n0 + c];
    }
  }

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging registers
  f32x4 ra[AI], rh[HI];
  int hb = 0, hmask = 0;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const int a4 = tid & 31;  // dY: 32 float4 per pixel row; slot q = tid + NT*i keeps the same column quad (NT % 32 == 0)

  auto load_regs = [&](int64_t u) {
    const int b = (int)(u / rows_per_img);
    const int rem = (int)(u - (int64_t)b * rows_per_img);
    const int ty = rem / xblocks, xb = rem - ty * xblocks;
    const int y = ty * TH;  // first output row of the unit
    // buffer descriptors (common.h) over this unit's image of dY and of X: out-of-range offsets read zeros
    const auto rsY = VAE_BUF_RSRC(p.dY + (int64_t)b * (g.Ho * ys) * (g.Wo * ys) * p.ldy, (size_t)(g.Ho * ys) * (g.Wo * ys) * p.ldy * 4u);
    const auto rsX = VAE_BUF_RSRC(p.X + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 4u);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int q = tid + NT * i;
      const int k = q >> 5;  // pixel of the unit: (row k / TW, col k % TW)
      const int pix = ((y + k / TW) * ys + p.y_oy) * (g.Wo * ys) + (xb * TW + (k % TW)) * ys + p.y_ox;  // sub-sampled view of dY
      const int c = m0 + a4 * 4;
      ra[i] = VAE_BUF_LOAD4(rsY, (q < AQ && c < p.M) ? (unsigned)((pix * p.ldy + c) * 4) : BUF_OOB);
    }
    hb = b;
    hmask = 0;
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      const int pp = q >> 3, k4 = q & 7;
      const int ir = pp / HWD, jc = pp - ir * HWD;
      const int hy = y - 1 + ir, hx = xb * TW - 1 + jc;
      const bool ok = (q < HQ) && ((unsigned)hy < (unsigned)Hb) && ((unsigned)hx < (unsigned)Wb);
      const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
      const int c = n0 + k4 * 4;
      rh[i] = VAE_BUF_LOAD4(rsX, (ok && c < p.N) ? (unsigned)(((sy * g.Ws + sx) * g.Cs + c) * 4) : BUF_OOB);
      hmask |= (ok ? 1 : 0) << i;
    }
  };
  auto store_lds = [&](float* sA, float* sH) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int q = tid + NT * i;
      if (q < AQ) {
        *reinterpret_cast<f32x4*>(&sA[(q >> 5) * LDA + a4 * 4]) = ra[i];
        if (do_bias) bsum += ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      if (q < HQ) {
        f32x4 v = rh[i];
        if (XF != VAE_XF_NONE) {
          const bool ok = (hmask >> i) & 1;
          const int o = ok ? (hb - b_lo) * BNT + (q & 7) * 4 : 0;
          v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
        }
        *reinterpret_cast<f32x4*>(&sH[(q >> 3) * LDH + (q & 7) * 4]) = v;
      }
    }
  };
  // k-group kk = 8 pixels of unit row kk / 4: 4 dY values and 3 taps x 4 halo values per lane feed 12 MFMAs.  The
  // operands of k-group kk+1 are requested before the MFMAs of kk are issued (pinned with sched_barrier: hipcc otherwise
  // interleaves reads and MFMAs with a full lgkmcnt(0) wait in front of every few MFMAs).
  float fa[2][4], fb[2][3][4];
  auto fetch = [&](const float* sA, const float* sH, int kk, float* a, float (*bq)[4]) {
    const int r = kk / (TW / 8), c0 = (kk % (TW / 8)) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = sA[(kk * 8 + lh * 4 + j) * LDA + mt * 32 + lr];
#pragma unroll
    for (int t = 0; t < 3; ++t)  // tap (kh = tg, kw = t): halo pixel (r + kh, c + kw)
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[t][j] = sH[((r + tg) * HWD + c0 + lh * 4 + j + t) * LDH + lr];
  };
  auto compute = [&](const float* sA, const float* sH, int k0, int k1) {  // k-groups [k0, k1); k0 is already fetched
#pragma unroll
    for (int kk = k0; kk < k1; ++kk) {
      if (kk + 1 < UPX / 8) fetch(sA, sH, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if ((wmask >> t) & 1) {  // wave-uniform
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][j], fb[kk & 1][t][j], acc[t], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nu > 0) {
    load_regs(ubeg);
    __syncthreads();  // scale/shift table visible
    store_lds(smem, smem + SA);
    if (nu > 1) load_regs(ubeg + 1);
    __syncthreads();
    for (int s = 0; s < nu; ++s) {
      const float* cA = smem + (s & 1) * STAGE;
      fetch(cA, cA + SA, 0, fa[0], fb[0]);
      compute(cA, cA + SA, 0, UPX / 16);
      if (s + 1 < nu) {  // staged in the shadow of the MFMAs already issued
        float* nA = smem + ((s + 1) & 1) * STAGE;
        store_lds(nA, nA + SA);
        if (s + 2 < nu) load_regs(ubeg + s + 2);
      }
      compute(cA, cA + SA, UPX / 16, UPX / 8);
      __syncthreads();
    }
  }

  const int64_t ld = (int64_t)9 * p.N;
  float* __restrict__ O = (p.nsplit == 1 ? p.out : p.partial + (int64_t)split * p.M * ld);
  const int col = n0 + lr;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = tg * 3 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < p.M) O[(int64_t)row * ld + (int64_t)tap * p.N + col] = p.alpha * acc[t][r];
    }
  }
  if (do_bias) {  // workgroup-uniform
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [NT/32][32]
    red[tid] = bsum;                              // tid = rowgroup*32 + a4
    __syncthreads();
    if (tid < BMT / 4) {
      f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < NT / (BMT / 4); ++r) t4 += red[r * (BMT / 4) + tid];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + tid * 4 + e;
        if (m < p.M) p.bias_partial[(int64_t)split * p.M + m] = t4[e];
      }
    }
  }
}

}  // namespace

bool wgrad3_tile_eligible(const vae_wgrad_args& a, bool vec) {
  const vae_conv_geom& g = a.g;
  if (!vec || a.batch != 1 || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.M <= 32 || a.N % BNT != 0 || g.Wo % TW != 0 || g.Ho % TH != 0) return false;
  if (g.mode == VAE_MODE_FWD && !(g.Ho == g.Hs && g.Wo == g.Ws)) return false;
  if (g.mode == VAE_MODE_UP2X && !(g.Ho == 2 * g.Hs && g.Wo == 2 * g.Ws)) return false;
  if (g.mode == VAE_MODE_DGRAD) return false;
  const size_t ys = a.y_step > 1 ? a.y_step : 1;
  if ((size_t)g.Ho * g.Wo * a.ldy * 4u * ys * ys >= BUF_MAX || (size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX) return false;  // one image per descriptor
  return true;
}

int64_t wgrad3_tile_units(const vae_conv_geom& g) { return (int64_t)g.B * (g.Ho / TH) * (g.Wo / TW); }

int launch_wgrad3_tile(const vae_wgrad_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int xblocks = g.Wo / TW;
  const int64_t nunits = wgrad3_tile_units(g);
  dim3 grid((unsigned)(((a.M + BMT - 1) / BMT) * (a.N / BNT)), (unsigned)a.nsplit, 1);
  const bool up = g.mode == VAE_MODE_UP2X;
#define WG3(UPV, XFV) hipLaunchKernelGGL((wgrad3_tile_kernel<UPV, XFV>), grid, dim3(NT), 0, st, a, xblocks, nunits)
  switch (a.xf) {
    case VAE_XF_NONE: if (up) WG3(true, VAE_XF_NONE); else WG3(false, VAE_XF_NONE); break;
    case VAE_XF_AFFINE: if (up) WG3(true, VAE_XF_AFFINE); else WG3(false, VAE_XF_AFFINE); break;
    default: if (up) WG3(true, VAE_XF_AFFINE_SILU); else WG3(false, VAE_XF_AFFINE_SILU); break;
  }
#undef WG3
  return 0;
}
