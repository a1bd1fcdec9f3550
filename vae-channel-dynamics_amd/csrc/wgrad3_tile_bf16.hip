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
// Two kernels: wgrad3_tile_bf16_kernel stages through registers (any storage combination, fused GroupNorm / SiLU of X); wgrad3_dma_bf16_kernel
// (further down; both operands as bf16 images, stride 1 or 2) stages by LDS-DMA into swizzled images -- what a bf16 step runs.
#include "bf16_frag.h"
#ifndef VAE_ABLATE
#define VAE_ABLATE 0  // diagnostic builds (tools/ablation_builds.sh, wrong results): bit 0 no global loads / DMA pieces fetch nothing, 1 no LDS stores, 4 no barrier
#endif

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



// Workgroup id -> (column = (co tile, ci tile), split).  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (one L2 each).  The columns of one split stream through the SAME pixels at the same pace (every ci tile re-reads the dY rows,
// every co tile the X halo): with id = column + columns * split the 8 columns of a 256 -> 256 layer sat on 8 different XCDs and
// every L2 fetched the split's pixels for itself -- 1.08 GB per launch from the memory side, 3.9 TB/s, which is what the staging
// cost (the kernel ran 0.276 ms with its DMA pieces, 0.218 with the same instructions fetching nothing).  Here XCD x takes the
// splits congruent x mod 8, all columns of a split together: one L2 fetches a split's pixels once.
__device__ __forceinline__ void wg_column_split(int nsplit, int& column, int& split) {
  const int cols = gridDim.x, L = blockIdx.y * cols + blockIdx.x;
  if (nsplit % 8 == 0) {
    const int j = L >> 3;
    column = j % cols;
    split = (j / cols) * 8 + (L & 7);
  } else {
    column = blockIdx.x;
    split = blockIdx.y;
  }
}

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
  int colw, split;
  wg_column_split(p.nsplit, colw, split);
  const int tm = colw / tilesN, tn = colw % tilesN;
  const int m0 = tm * BMT, n0 = tn * BNT;
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
        if (!(VAE_ABLATE & 2)) store_lds(nA, nA + SA);
        if (!(VAE_ABLATE & 1)) load_regs();
      }
      compute(cA, cA + SA, 6, 12);
      if (!(VAE_ABLATE & 16)) __syncthreads();
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


// ------------------------------------------------------------------------------------------------------------------------
// Both operands as bf16 images (X16 and dY16: every 3x3 layer of a bf16 step): the same workgroup shape, wave roles, MFMA order
// and epilogue as the kernel above, but the unit's two images are staged by LDS-DMA (buffer_load_dwordx4 ... lds) instead of
// global load -> registers -> ds_write_b128.  Ablations of the register-staged kernel at 256 -> 256 channels (tools/
// microbench_bf16.py, diagnostic builds): 0.298 ms; without the global loads 0.264; also without the LDS stores 0.219; also
// without the barrier 0.211 -- a quarter of the kernel was staging, most of it the 16-byte LDS stores (13 cycles each through
// the VGPR -> LDS path that all twelve waves share).  The DMA needs no registers, no stores and no waits inside the step.
//   * A DMA wave-instruction writes 64 x 16 B lane-linear at M0: no row padding is possible inside a piece, so the images are
//     swizzled by WHICH 16 bytes a lane fetches.  The transposing read (ds_read_b64_tr_b16) of a 32-lane group takes 4 pixel rows
//     x 64 B; the four rows have to sit in four different 64-byte bank ranges of the 256-byte bank cycle:
//       dY image  (256 B per pixel = four 64-byte segments): piece = 4 pixels; segment g of pixel q at  1024 (px >> 2) + 256 q +
//                 64 ((g + q) & 3),  q = px & 3   -- the rows of a read are rotated against each other;
//       halo image (128 B per pixel = two segments):          piece = 8 pixels; segment s of halo pixel hp at  512 (hp >> 2) + 256 s
//                 + 64 (hp & 3)  -- four consecutive pixels are four bank ranges wherever the group starts (the tap shifts 0, 1, 2
//                 move the group off the piece grid); a lane keeps the addresses of its pixel + 0..3, the rest is an immediate.
//   * 33 pieces per unit (16 + 17) + 3 all-out-of-range ones: three per wave, so every wave waits for the same count.  Three
//     stages in LDS (108 KB): the pieces of unit s + 2 are requested at the top of step s (its buffer was last read in step s - 1,
//     behind that step's barrier), `s_waitcnt vmcnt(3)` in front of the barrier of step s retires those of unit s + 1.  Inline asm,
//     as in conv3_wino4.hip: with a DMA it can see in flight hipcc waits for vmcnt(0) at every other load.
//   * bias gradient (workgroups of the first ci block): the column sums of dY are read back from the LDS image, same thread ->
//     (8 columns, pixels) assignment and same order of additions as the register-staged kernel: bitwise the same sums.
constexpr int DPIECES = 36, DSTAGE = DPIECES * 1024;   // bytes per stage: dY image 16 KB, halo image 17 KB, 3 KB never read
constexpr int DMA_LDS = 3 * DSTAGE;

// STR = 2: the stride-2 downsampler convolutions (pad (0,1,0,1)): unit = 1 x 32 output pixels, halo = 3 rows x 65 columns -- again
// 8 + 25 = 33 pieces.  A lane's four rows of a transposing read are then TWO halo pixels apart, so the bank range of a halo pixel is
// (hp >> 1) & 3 instead of hp & 3:  segment s of pixel hp at  1024 (hp >> 3) + 64 (4 (2 (hp & 1) + s) + ((hp >> 1) & 3)).
template <bool UP, int STR>
__global__ __launch_bounds__(NT) void wgrad3_dma_bf16_kernel(vae_wgrad_args p, int tiles_x, int tiles_y, int64_t nunits) {
  static_assert(STR == 1 || (STR == 2 && !UP), "stride 1 (plain or over a virtual upsample) or stride 2");
  constexpr int UTH = STR == 2 ? 1 : TH, UPXS = UTH * TW;          // output rows / pixels of a unit
  constexpr int HR = STR == 2 ? 3 : TH + 2, HWS = STR == 2 ? 2 * TW + 1 : HWD, HPXS = HR * HWS;  // halo rows, columns, pixels
  constexpr int ND = UPXS / 4, NH = (HPXS + 7) / 8;                // dY pieces, halo pieces
  static_assert(ND + NH <= DPIECES, "three pieces per wave");
  constexpr int NKG = UPXS / 16;                                    // 16-pixel k-groups of a unit
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int mt = wave & 1, nt = (wave >> 1) & 1, tg = wave >> 2;  // 64-row co block, 32-column ci block, filter row kh
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const vae_conv_geom g = p.g;
  const int tilesN = p.N / BNT;
  int colw, split;
  wg_column_split(p.nsplit, colw, split);
  const int tm = colw / tilesN, tn = colw % tilesN;
  const int m0 = tm * BMT, n0 = tn * BNT;
  const int64_t per = (nunits + p.nsplit - 1) / p.nsplit;
  const int64_t ubeg = split * per, uend = min(nunits, ubeg + per);
  const int nu = (int)max((int64_t)0, uend - ubeg);
  const int Hb = UP ? 2 * g.Hs : g.Hs, Wb = UP ? 2 * g.Ws : g.Ws;
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0;
  const int units_per_img = tiles_x * tiles_y;
  const int ys = (!UP && p.y_step > 1) ? p.y_step : 1;
  const int wmask = ((p.tapmask ? p.tapmask : 0x1ff) >> (3 * tg)) & 7;

  f32x16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mi][r] = 0.f;

  // ---- this lane's share of the wave's three pieces: what it fetches (per-unit part added in request()) ----
  // piece id = 3 wave + j: 0..15 dY (pixels 4 id ..), 16..32 halo (halo pixels 8 (id - 16) ..), 33..35 nothing
  int kind[3];        // 0 = dY, 1 = halo, 2 = none (uniform per wave)
  unsigned yoff[3];   // dY: byte offset inside the unit's first pixel row / BUF_OOB (channel beyond M)
  int hir[3], hjc[3], hc[3];  // halo: row and column inside the 4 x 34 halo, first channel
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int id = wave * 3 + j;
    const int sl = lane >> 2, ci = lane & 3;
    kind[j] = (VAE_ABLATE & 1) ? 2 : (id < ND ? 0 : (id < ND + NH ? 1 : 2));  // (ablation: every piece out of range, same instructions)
    {  // dY: slot sl of the piece = pixel q = sl >> 2, bank range sl & 3 -> segment ((sl & 3) - q) & 3
      const int q = sl >> 2, sg = ((sl & 3) - q) & 3;
      const int px = 4 * (id % ND) + q, c = m0 + sg * 32 + ci * 8;
      yoff[j] = c < p.M ? (unsigned)((((px >> 5) * ys * (g.Wo * ys) + (px & 31) * ys) * p.ldy + c) * 2) : BUF_OOB;
    }
    {  // halo: slot sl = (quad, segment, pixel in quad); stride 2: (parity, segment, pixel pair)
      const int hp = 8 * (id - ND) + (STR == 2 ? 2 * (sl & 3) + (sl >> 3) : 4 * (sl >> 3) + (sl & 3)), sgm = (sl >> 2) & 1;
      hir[j] = hp / HWS;
      hjc[j] = hp - hir[j] * HWS;
      hc[j] = n0 + sgm * 32 + ci * 8;
      if (hp >= HPXS) hc[j] = 0x40000000;  // (beyond the halo: never in range)
    }
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)dsm);
  // the unit to request next (image, tile row, tile column), advanced without divisions
  int nb_ = 0, nty_ = 0, ntx_ = 0, nleft_ = nu;
  if (nu > 0) {
    nb_ = (int)(ubeg / units_per_img);
    const int rem = (int)(ubeg - (int64_t)nb_ * units_per_img);
    nty_ = rem / tiles_x;
    ntx_ = rem - nty_ * tiles_x;
  }
  auto request = [&](int buf) {  // the three pieces of the next unit into stage `buf`; beyond the range: descriptors of size 0 (zeros)
    const bool valid = nleft_ > 0;
    const int b = nb_, y0 = nty_ * UTH, x0 = ntx_ * TW;
    --nleft_;
    if (++ntx_ == tiles_x) {
      ntx_ = 0;
      if (++nty_ == tiles_y) { nty_ = 0; ++nb_; }
    }
    const size_t ybytes = (size_t)(g.Ho * ys) * (g.Wo * ys) * p.ldy * 2u, xbytes = (size_t)g.Hs * g.Ws * g.Cs * 2u;
    const auto rsY = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.dY16) + (int64_t)b * (g.Ho * ys) * (g.Wo * ys) * p.ldy, valid ? ybytes : (size_t)0);
    const auto rsX = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.X16) + (int64_t)b * g.Hs * g.Ws * g.Cs, valid ? xbytes : (size_t)0);
    const unsigned ybase = (unsigned)((((y0 * ys + p.y_oy) * (g.Wo * ys) + x0 * ys + p.y_ox) * p.ldy) * 2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * (unsigned)DSTAGE + (unsigned)(wave * 3 + j) * 1024u);
      unsigned keep;
      if (kind[j] == 0) {  // uniform
        const unsigned off = yoff[j] == BUF_OOB ? BUF_OOB : yoff[j] + ybase;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(off), "s"(rsY) : "memory");
      } else {
        const int hy = STR == 2 ? 2 * y0 + hir[j] : y0 - 1 + hir[j], hx = STR == 2 ? 2 * x0 + hjc[j] : x0 - 1 + hjc[j];
        const bool ok = kind[j] == 1 && ((unsigned)hy < (unsigned)Hb) && ((unsigned)hx < (unsigned)Wb) && hc[j] < p.N;
        const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
        const unsigned off = ok ? (unsigned)(((sy * g.Ws + sx) * g.Cs + hc[j]) * 2) : BUF_OOB;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(off), "s"(rsX) : "memory");
      }
    }
  };

  // ---- fragments: byte addresses inside a stage ----
  // dY: pixel P = 16 kg + 8 lh + trq (+ 4), columns mt*64 + mi*32 + trh*16 + trp*4: P & 3 = trq
  int aoffb[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) aoffb[mi] = lh * 2048 + trq * 256 + (((mt * 2 + mi) + trq) & 3) * 64 + trh * 32 + trp * 8;
  // halo: pixel hp = L + C,  L = tg*34 + 8 lh + trq,  C = r*34 + c0 + t (+ 4): address of L + m for m = 0..3, + 512 (C >> 2)
  // (stride 2: L = tg*65 + 2 (8 lh + trq), C = 32 kg + t (+ 8): address of L + m for m = 0..2, + 1024 (C >> 3))
  int boffb[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (STR == 2) {
      const int hp = tg * HWS + 2 * (lh * 8 + trq) + m;
      boffb[m] = ND * 1024 + (hp >> 3) * 1024 + (((hp & 1) * 2 + nt) * 4 + ((hp >> 1) & 3)) * 64 + trh * 32 + trp * 8;
    } else {
      const int hp = tg * HWD + lh * 8 + trq + m;
      boffb[m] = ND * 1024 + (hp >> 2) * 512 + nt * 256 + (hp & 3) * 64 + trh * 32 + trp * 8;
    }
  }
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto tr8 = [&](const unsigned char* lo, const unsigned char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  auto fetch_a = [&](const unsigned char* st, int kg, bf16x8* a) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi] = tr8(st + aoffb[mi] + kg * 4096, st + aoffb[mi] + kg * 4096 + 1024);
  };
  auto fetch_b = [&](const unsigned char* st, int grp) {
    const int kg = grp / 3, t = grp - kg * 3;
    if (STR == 2) return tr8(st + boffb[t] + kg * 4096, st + boffb[t] + kg * 4096 + 1024);
    const int C = (kg >> 1) * HWD + (kg & 1) * 16 + t;
    return tr8(st + boffb[C & 3] + (C >> 2) * 512, st + boffb[C & 3] + ((C >> 2) + 1) * 512);
  };
  bf16x8 fa[2][2], fb[2];
  auto compute = [&](const unsigned char* st) {  // the unit's 12 MFMA groups; the fragments of group i + 1 are requested before group i
    fetch_a(st, 0, fa[0]);
    fb[0] = fetch_b(st, 0);
#pragma unroll
    for (int grp = 0; grp < 3 * NKG; ++grp) {
      const int kg = grp / 3, t = grp - kg * 3;
      if (grp + 1 < 3 * NKG) {
        if (t == 2) fetch_a(st, kg + 1, fa[(kg + 1) & 1]);
        fb[(grp + 1) & 1] = fetch_b(st, grp + 1);
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
  // bias gradient: thread -> columns (tid & 15) * 8 .. + 7 at pixels (tid >> 4) and (tid >> 4) + 48 of the unit
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, bsum2 = {0.f, 0.f, 0.f, 0.f};
  int bofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + NT * i, px = q >> 4, cg = q & 15;  // 8-column group cg = segment cg >> 2, 16-byte chunk cg & 3
    bofs[i] = q < UPXS * 16 ? (px >> 2) * 1024 + (px & 3) * 256 + ((((cg >> 2) + px) & 3) * 64) + (cg & 3) * 16 : -1;
  }
  auto bias_sums = [&](const unsigned char* st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (bofs[i] >= 0) {
        const uint4 r = *reinterpret_cast<const uint4*>(st + bofs[i]);
        bsum[0] += __builtin_bit_cast(float, r.x << 16); bsum[1] += __builtin_bit_cast(float, r.x & 0xffff0000u);
        bsum[2] += __builtin_bit_cast(float, r.y << 16); bsum[3] += __builtin_bit_cast(float, r.y & 0xffff0000u);
        bsum2[0] += __builtin_bit_cast(float, r.z << 16); bsum2[1] += __builtin_bit_cast(float, r.z & 0xffff0000u);
        bsum2[2] += __builtin_bit_cast(float, r.w << 16); bsum2[3] += __builtin_bit_cast(float, r.w & 0xffff0000u);
      }
    }
  };

  if (nu > 0) {
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // unit 0 has landed
    __syncthreads();
    int cur = 0, nxt2 = 2;
    for (int s = 0; s < nu; ++s) {
      request(nxt2);  // unit s + 2 (beyond the range: zeros into a buffer nobody reads)
      const unsigned char* st = dsm + cur * DSTAGE;
      compute(st);
      if (do_bias) bias_sums(st);  // uniform
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // this wave's pieces of unit s + 1 have landed
      __syncthreads();
      cur = cur == 2 ? 0 : cur + 1;
      nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last two requests write LDS the bias reduction below reuses)
    __syncthreads();
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
  if (do_bias) {
    f32x4* red = reinterpret_cast<f32x4*>(dsm);  // [NT/16][16][2]: thread t holds columns (t & 15) * 8 .. + 7
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
  }
}

}  // namespace

bool wgrad3_tile_bf16_dma(const vae_wgrad_args& a);
// the stride-2 downsampler convolutions (3x3, padding (0,1,0,1)) on the LDS-DMA kernel: 1 x 32-pixel units
static bool wgrad3_s2_geom(const vae_wgrad_args& a) {
  const vae_conv_geom& g = a.g;
  return g.mode == VAE_MODE_FWD && g.taps == 9 && g.stride == 2 && g.pad_t == 0 && g.pad_l == 0 && g.Hs == 2 * g.Ho && g.Ws == 2 * g.Wo &&
         a.tapmask == 0 && a.y_step <= 1;
}
bool wgrad3_tile_bf16_eligible(const vae_wgrad_args& a, bool vec) {
  const vae_conv_geom& g = a.g;
  if (vec && a.batch == 1 && wgrad3_s2_geom(a)) {
    if (!wgrad3_tile_bf16_dma(a) || a.M <= 32 || a.N % BNT != 0 || g.Wo % TW != 0) return false;
    return (size_t)g.Ho * g.Wo * a.ldy * 4u < BUF_MAX && (size_t)g.Hs * g.Ws * g.Cs * 4u < BUF_MAX;
  }
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
// the LDS-DMA kernel: both operands as 16-byte-aligned bf16 images whose rows are whole 16-byte pieces
bool wgrad3_tile_bf16_dma(const vae_wgrad_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.X16 == nullptr || a.dY16 == nullptr || a.xf != VAE_XF_NONE || vae_opt().no_wgrad_dma) return false;
  if (a.M % 8 != 0 || a.ldy % 8 != 0 || g.Cs % 8 != 0 || !aligned16(a.X16) || !aligned16(a.dY16)) return false;
  return true;
}
int64_t wgrad3_tile_bf16_units(const vae_conv_geom& g) { return (int64_t)g.B * (g.stride == 2 ? g.Ho : g.Ho / TH) * (g.Wo / TW); }
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
  if (wgrad3_tile_bf16_dma(a)) {  // both operands as bf16 images: staged by LDS-DMA
    if (g.stride == 2) {
      VAE_RESERVE_LDS((wgrad3_dma_bf16_kernel<false, 2>), DMA_LDS, "wgrad3_dma_bf16");
      hipLaunchKernelGGL((wgrad3_dma_bf16_kernel<false, 2>), grid, dim3(NT), DMA_LDS, st, a, tx, g.Ho, nunits);
    } else if (up) {
      VAE_RESERVE_LDS((wgrad3_dma_bf16_kernel<true, 1>), DMA_LDS, "wgrad3_dma_bf16");
      hipLaunchKernelGGL((wgrad3_dma_bf16_kernel<true, 1>), grid, dim3(NT), DMA_LDS, st, a, tx, ty, nunits);
    } else {
      VAE_RESERVE_LDS((wgrad3_dma_bf16_kernel<false, 1>), DMA_LDS, "wgrad3_dma_bf16");
      hipLaunchKernelGGL((wgrad3_dma_bf16_kernel<false, 1>), grid, dim3(NT), DMA_LDS, st, a, tx, ty, nunits);
    }
    return 0;
  }
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
