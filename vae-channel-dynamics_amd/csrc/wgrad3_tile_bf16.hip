// bf16-compute weight gradient of a 3x3 stride-1 convolution (plain or over a virtual nearest-2x upsample):
//   dW[co][tap][ci] = sum_{b,y,x} dY[b][y][x][co] * XF(X)[b][y+kh-1][x+kw-1][ci]
// fp32 tensors in HBM, operands rounded to bf16 while staged in LDS (after the fp32 GroupNorm+SiLU), products
// on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 split-K partial slabs (vae_reduce_splits sums them).
//
// Workgroup (12 waves) = 128 co x 64 ci x all 9 taps over a range of units; one unit = a 2x32-pixel output tile.
// Per unit the dY tile [64 px][128 co] and the 4x34-pixel X halo [136 px][64 ci] are staged pixel-major (double
// buffered)
// (their memory order); since the contraction runs over pixels, both MFMA operands need 8 consecutive PIXELS
// per lane, which the transposing LDS load (ds_read_b64_tr_b16) delivers from the pixel-major images.
// Wave (mt, nt, kh) owns co rows mt*64.. (two 32-row blocks), ci columns nt*32.. and filter row kh: 3 taps x 2 co blocks = 6
// accumulators.  Per 16-pixel k-group it reads 2 dY fragments (shared by its 3 taps) and 3 halo fragments for 6 MFMAs (5 KB;
// the earlier 32 co x 64 ci split read 1 + 6 = 7 KB for the same 6 MFMAs).
#include "bf16_frag.h"

namespace {


constexpr int TH = 2, TW = 32, UPX = TH * TW;   // 64 pixels per unit
constexpr int HWD = TW + 2, HPX = (TH + 2) * HWD;  // 136 halo pixels
constexpr int BMT = 128, BNT = 64;
constexpr int LDA = BMT + 32;                   // dY image row stride (320 B: tr reads conflict-free)
constexpr int LDH = BNT + 32;                   // halo image row stride (192 B)
constexpr int SA = UPX * LDA, SH = HPX * LDH;   // u16 elements
constexpr int STAGE = SA + SH;
constexpr int NT = 768;
constexpr int AQ = UPX * (BMT / 4);             // dY float4 slots (2048)
constexpr int AI = (AQ + NT - 1) / NT;          // 3
constexpr int HQ = HPX * (BNT / 4);             // halo float4 slots (2176)
constexpr int HI = (HQ + NT - 1) / NT;          // 3


// Y16: dY comes as a bf16 image (vae_wgrad_args.dY16): 16-byte loads written to LDS as they are
template <bool UP, int XF, bool X16, bool Y16>
__global__ __launch_bounds__(NT) void wgrad3_tile_bf16_kernel(vae_wgrad_args p, int tiles_x, int tiles_y, int64_t nunits) {
  constexpr int SSB = (XF != VAE_XF_NONE) ? 2 * SS_HALF * 2 : 0;
  __shared__ __attribute__((aligned(16))) u16 smem[2 * STAGE + SSB];
  float* sS = reinterpret_cast<float*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int mt = wave & 1, nt = (wave >> 1) & 1, tg = wave >> 2;  // 64-row co block, 32-column ci block, filter row kh
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
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
  const int units_per_img = tiles_x * tiles_y;
  // phase convolutions of an upsampler (vaehip.h): dY is a sub-sampled view (pixel (y,x) at (y*ys+y_oy, x*ys+y_ox)) and only the
  // taps of tapmask are computed -- a wave whose kernel row is masked out only helps with the staging, the others skip the
  // masked columns (their accumulators stay zero and are written as zeros)
  const int ys = (!UP && p.y_step > 1) ? p.y_step : 1;
  const int wmask = ((p.tapmask ? p.tapmask : 0x1ff) >> (3 * tg)) & 7;

  const int b_lo = nu > 0 ? (int)(ubeg / units_per_img) : 0;
  if (XF != VAE_XF_NONE && nu > 0) {
    const int nb = (int)((uend - 1) / units_per_img) - b_lo + 1;
    const int nent = min(nb * BNT, SS_HALF);
    for (int i = tid; i < nent; i += NT) {
      const int j = i / BNT, c = i - j * BNT;
      sS[i] = p.scale[(int64_t)(b_lo + j) * g.Cs + n0 + c];
      sS[SS_HALF + i] = p.shift[(int64_t)(b_lo + j) * g.Cs + n0 + c];
    }
  }

  f32x16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mi][r] = 0.f;

  // with a bf16 activation image (X16) a halo slot is 8 channels = one 16-byte load, written to LDS as it is
  constexpr int HQ16 = HPX * (BNT / 8), HI16 = (HQ16 + NT - 1) / NT;  // 1088 slots, 2 per thread
  constexpr int AQ16 = UPX * (BMT / 8), AI16 = (AQ16 + NT - 1) / NT;  // dY image: 1024 slots of 8 channels, 2 per thread
  uint4 rh16[X16 ? HI16 : 1], ra16[Y16 ? AI16 : 1];
  f32x4 ra[Y16 ? 1 : AI], rh[HI];
  int hb = 0, hmask = 0;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, bsum2 = {0.f, 0.f, 0.f, 0.f};  // Y16: 8 columns per thread
  // Staging routines start from an opaque copy of the thread id (hipcc would otherwise keep every slot's address and
  // mask in VGPRs across the unit loop and spill), and EVERY step issues the same loads: a unit beyond the range is
  // requested with all lanes out of range (zeros, no traffic), so the waitcnt pass knows exactly what is in flight.
  // (the id is rebuilt from the wave number in an SGPR and the lane count of EXEC: no VGPR lives across the loop for it)
  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  auto fresh_tid = [&]() {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return wave_s * 64 + l;
  };

  // the unit to request next, decoded once and then advanced (image, tile row, tile column): no division per step
  int nb_ = 0, nty_ = 0, ntx_ = 0, nleft_ = nu;
  if (nu > 0) {
    nb_ = (int)(ubeg / units_per_img);
    const int rem = (int)(ubeg - (int64_t)nb_ * units_per_img);
    nty_ = rem / tiles_x;
    ntx_ = rem - nty_ * tiles_x;
  }
  auto load_regs = [&]() {
    const int tid = fresh_tid();
    const int a4 = tid & 31;  // dY column quad (NT % 32 == 0: the same for every slot of this thread)
    const bool valid = nleft_ > 0;
    const int AQv = valid ? AQ : 0;
    const unsigned Hv = valid ? (unsigned)Hb : 0u;
    const int b = nb_;
    const int y0 = nty_ * TH, x0 = ntx_ * TW;
    --nleft_;
    if (++ntx_ == tiles_x) {
      ntx_ = 0;
      if (++nty_ == tiles_y) { nty_ = 0; ++nb_; }
    }
    // buffer descriptors (common.h) over this unit's image of dY and of X: out-of-range offsets read zeros
    const auto rsX = VAE_BUF_RSRC(p.X + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 4u);
    if (Y16) {
      const auto rsY16 = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.dY16) + (int64_t)b * (g.Ho * ys) * (g.Wo * ys) * p.ldy,
                                      (size_t)(g.Ho * ys) * (g.Wo * ys) * p.ldy * 2u);
      const int AQ16v = valid ? AQ16 : 0;
#pragma unroll
      for (int i = 0; i < AI16; ++i) {
        const int q = tid + NT * i;
        const int px = q >> 4, c = m0 + (q & 15) * 8;
        const int pix = ((y0 + (px >> 5)) * ys + p.y_oy) * (g.Wo * ys) + (x0 + (px & 31)) * ys + p.y_ox;
        ra16[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsY16, (q < AQ16v && c < p.M) ? (unsigned)((pix * p.ldy + c) * 2) : BUF_OOB, 0, 0));
      }
    } else {
      const auto rsY = VAE_BUF_RSRC(p.dY + (int64_t)b * (g.Ho * ys) * (g.Wo * ys) * p.ldy, (size_t)(g.Ho * ys) * (g.Wo * ys) * p.ldy * 4u);
#pragma unroll
      for (int i = 0; i < (Y16 ? 1 : AI); ++i) {
        const int q = tid + NT * i;
        const int px = q >> 5;  // 0..63 : (row px>>5, col px&31)
        const int pix = ((y0 + (px >> 5)) * ys + p.y_oy) * (g.Wo * ys) + (x0 + (px & 31)) * ys + p.y_ox;
        const int c = m0 + a4 * 4;
        ra[i] = VAE_BUF_LOAD4(rsY, (q < AQv && c < p.M) ? (unsigned)((pix * p.ldy + c) * 4) : BUF_OOB);
      }
    }
    if (X16) {
      const auto rsX16 = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.X16) + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 2u);
#pragma unroll
      for (int i = 0; i < HI16; ++i) {
        const int q = tid + NT * i;
        const int pp = q >> 3, k8 = q & 7;
        const int ir = pp / HWD, jc = pp - ir * HWD;
        const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
        const bool ok = (q < HQ16) && ((unsigned)hy < Hv) && ((unsigned)hx < (unsigned)Wb);
        const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
        const int c = n0 + k8 * 8;
        rh16[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX16, (ok && c < p.N) ? (unsigned)(((sy * g.Ws + sx) * g.Cs + c) * 2) : BUF_OOB, 0, 0));
      }
      return;
    }
    hb = b;
    hmask = 0;
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      const int pp = q >> 4, k4 = q & 15;
      const int ir = pp / HWD, jc = pp - ir * HWD;
      const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
      const bool ok = (q < HQ) && ((unsigned)hy < Hv) && ((unsigned)hx < (unsigned)Wb);
      const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
      const int c = n0 + k4 * 4;
      rh[i] = VAE_BUF_LOAD4(rsX, (ok && c < p.N) ? (unsigned)(((sy * g.Ws + sx) * g.Cs + c) * 4) : BUF_OOB);
      hmask |= (ok ? 1 : 0) << i;
    }
  };
  auto store_lds = [&](u16* sA, u16* sH) {
    const int tid = fresh_tid();
    const int a4 = tid & 31;
    if (Y16) {
#pragma unroll
      for (int i = 0; i < AI16; ++i) {
        const int q = tid + NT * i;
        if (q < AQ16) {
          *reinterpret_cast<uint4*>(&sA[(q >> 4) * LDA + (q & 15) * 8]) = ra16[i];
          if (do_bias) {  // column sums of the bf16 values, in fp32 (NT % 16 == 0: a thread keeps its 8 columns)
            const uint4 r = ra16[i];
            bsum[0] += __builtin_bit_cast(float, r.x << 16); bsum[1] += __builtin_bit_cast(float, r.x & 0xffff0000u);
            bsum[2] += __builtin_bit_cast(float, r.y << 16); bsum[3] += __builtin_bit_cast(float, r.y & 0xffff0000u);
            bsum2[0] += __builtin_bit_cast(float, r.z << 16); bsum2[1] += __builtin_bit_cast(float, r.z & 0xffff0000u);
            bsum2[2] += __builtin_bit_cast(float, r.w << 16); bsum2[3] += __builtin_bit_cast(float, r.w & 0xffff0000u);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < (Y16 ? 1 : AI); ++i) {
        const int q = tid + NT * i;
        if (q < AQ) {
          *reinterpret_cast<uint2*>(&sA[(q >> 5) * LDA + a4 * 4]) = pack4(ra[i]);
          if (do_bias) bsum += ra[i];
        }
      }
    }
    if (X16) {
#pragma unroll
      for (int i = 0; i < HI16; ++i) {
        const int q = tid + NT * i;
        if (q < HQ16) *reinterpret_cast<uint4*>(&sH[(q >> 3) * LDH + (q & 7) * 8]) = rh16[i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = tid + NT * i;
      if (q < HQ) {
        f32x4 v = rh[i];
        if (XF != VAE_XF_NONE) {
          const bool ok = (hmask >> i) & 1;
          const int o = ok ? (hb - b_lo) * BNT + (q & 15) * 4 : 0;
          v = xform4_tab<XF>(v, sS + o, sS + SS_HALF + o, ok);
        }
        *reinterpret_cast<uint2*>(&sH[(q >> 4) * LDH + (q & 15) * 4]) = pack4(v);
      }
    }
  };
  // One unit = 12 MFMA groups (4 k-groups of 16 pixels x 3 taps of this wave's kernel row), each 2 MFMAs (the two
  // 32-row co blocks) on two dY fragments (shared by the 3 taps of a k-group) and one halo fragment.  The
  // fragments of group i+1 are requested before the MFMAs of group i are issued (pinned with sched_barrier: hipcc
  // otherwise sinks every read to just before its MFMA and waits for it there, 24 exposed LDS round trips per unit).
  // this lane's element offsets into the dY image and the halo image (everything else is a compile-time constant)
  const int aoff = (lh * 8 + trq) * LDA + mt * 64 + trh * 16 + trp * 4;
  const int boff = (tg * HWD + lh * 8 + trq) * LDH + nt * 32 + trh * 16 + trp * 4;
  auto fetch_a = [&](const u16* sA, int kg, bf16x8* a) {
    a[0] = frag_tr(sA + aoff + kg * 16 * LDA, LDA);
    a[1] = frag_tr(sA + aoff + kg * 16 * LDA + 32, LDA);
  };
  auto fetch_b = [&](const u16* sH, int grp) {
    const int kg = grp / 3, t = grp - kg * 3;  // tap (kh = tg, kw = t): halo pixel (r + kh, c + kw)
    const int r = kg >> 1, c0 = (kg & 1) * 16;
    return frag_tr(sH + boff + (r * HWD + c0 + t) * LDH, LDH);
  };
  bf16x8 fa[2][2], fb[2];
  auto compute = [&](const u16* sA, const u16* sH, int g0, int g1) {  // MFMA groups [g0, g1) of the unit; g0 is prefetched
#pragma unroll
    for (int grp = g0; grp < g1; ++grp) {
      const int kg = grp / 3, t = grp - kg * 3;
      if (grp + 1 < 12) {
        if (t == 2) fetch_a(sA, kg + 1, fa[(kg + 1) & 1]);
        fb[(grp + 1) & 1] = fetch_b(sH, grp + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((wmask >> t) & 1) {  // uniform per wave (always true without a tap mask)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[t][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kg & 1][mi], fb[grp & 1], acc[t][mi], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nu > 0) {
    load_regs();
    __syncthreads();  // scale/shift table visible
    store_lds(smem, smem + SA);
    load_regs();
    __syncthreads();
    for (int s = 0; s < nu; ++s) {
      const u16* cA = smem + (s & 1) * STAGE;
      fetch_a(cA, 0, fa[0]);
      fb[0] = fetch_b(cA + SA, 0);
      compute(cA, cA + SA, 0, 6);
      {  // staged in the shadow of the MFMAs already issued; unconditional (see load_regs)
        u16* nA = smem + ((s + 1) & 1) * STAGE;
        store_lds(nA, nA + SA);
        load_regs();
      }
      compute(cA, cA + SA, 6, 12);
      __syncthreads();
    }
  }

  const int64_t ld = (int64_t)9 * p.N;
  float* __restrict__ O = (p.nsplit == 1 ? p.out : p.partial + (int64_t)split * p.M * ld);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = tg * 3 + t;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int col = n0 + nt * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + mt * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M) O[(int64_t)row * ld + (int64_t)tap * p.N + col] = p.alpha * acc[t][mi][r];
      }
    }
  }
  if (do_bias && Y16) {
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [NT/16][16][2]: thread t holds columns (t & 15) * 8 .. + 7
    red[tid * 2] = bsum;
    red[tid * 2 + 1] = bsum2;
    __syncthreads();
    if (tid < BMT / 4) {  // quad `tid` of the 128 columns = half (tid & 1) of column group tid >> 1
      f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < NT / 16; ++r) t4 += red[(r * 16 + (tid >> 1)) * 2 + (tid & 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + tid * 4 + e;
        if (m < p.M) p.bias_partial[(int64_t)split * p.M + m] = t4[e];
      }
    }
  } else if (do_bias) {
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [NT/32][32]
    red[tid] = bsum;
    __syncthreads();
    if (tid < BMT / 4) {
      f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < NT / 32; ++r) t4 += red[r * 32 + tid];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + tid * 4 + e;
        if (m < p.M) p.bias_partial[(int64_t)split * p.M + m] = t4[e];
      }
    }
  }
}

}  // namespace

bool wgrad3_tile_bf16_eligible(const vae_wgrad_args& a, bool vec) {
  const vae_conv_geom& g = a.g;
  if (!vec || a.batch != 1 || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.M <= 32 || a.N % BNT != 0 || g.Wo % TW != 0 || g.Ho % TH != 0) return false;
  if (g.mode == VAE_MODE_FWD && !(g.Ho == g.Hs && g.Wo == g.Ws)) return false;
  if (g.mode == VAE_MODE_UP2X && !(g.Ho == 2 * g.Hs && g.Wo == 2 * g.Ws)) return false;
  if (g.mode == VAE_MODE_DGRAD) return false;
  const size_t ys = a.y_step > 1 ? a.y_step : 1;
  if ((ys > 1 || a.tapmask != 0) && g.mode != VAE_MODE_FWD) return false;
  if ((size_t)g.Ho * g.Wo * a.ldy * 4u * ys * ys >= BUF_MAX || (size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX) return false;  // one image per descriptor
  return true;
}
int64_t wgrad3_tile_bf16_units(const vae_conv_geom& g) { return (int64_t)g.B * (g.Ho / TH) * (g.Wo / TW); }
int wgrad3_tile_bf16_columns(const vae_wgrad_args& a) { return ((a.M + BMT - 1) / BMT) * (a.N / BNT); }

int launch_wgrad3_tile_bf16(const vae_wgrad_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / TW, ty = g.Ho / TH;
  const int64_t nunits = wgrad3_tile_bf16_units(g);
  dim3 grid((unsigned)wgrad3_tile_bf16_columns(a), (unsigned)a.nsplit, 1);
  const bool up = g.mode == VAE_MODE_UP2X;
  const bool y16 = a.dY16 != nullptr;
#define WG3(UPV, XFV, X16V) do { if (y16) hipLaunchKernelGGL((wgrad3_tile_bf16_kernel<UPV, XFV, X16V, true>), grid, dim3(NT), 0, st, a, tx, ty, nunits); \
                                else hipLaunchKernelGGL((wgrad3_tile_bf16_kernel<UPV, XFV, X16V, false>), grid, dim3(NT), 0, st, a, tx, ty, nunits); } while (0)
  if (a.X16 != nullptr) {  // transformed bf16 activation image (xf == NONE checked by the caller)
    if (up) WG3(true, VAE_XF_NONE, true); else WG3(false, VAE_XF_NONE, true);
    return 0;
  }
  switch (a.xf) {
    case VAE_XF_NONE: if (up) WG3(true, VAE_XF_NONE, false); else WG3(false, VAE_XF_NONE, false); break;
    case VAE_XF_AFFINE: if (up) WG3(true, VAE_XF_AFFINE, false); else WG3(false, VAE_XF_AFFINE, false); break;
    default: if (up) WG3(true, VAE_XF_AFFINE_SILU, false); else WG3(false, VAE_XF_AFFINE_SILU, false); break;
  }
#undef WG3
  return 0;
}
