// bf16 mode: weight gradient of a 3x3 stride-1 'same' convolution with a <= 4-channel side (encoder.conv_in: 3 input channels;
// decoder.conv_out: 3 output channels) on the matrix pipe.  wgrad_smallk_kernel (skinny.hip) runs these on the VALU, one lane
// per channel of the wide side with 36 FMAs and nine LDS broadcasts per pixel: 1.1 + 1.5 ms per step at 256x256, batch 32,
// for two launches whose HBM traffic (one 128-channel bf16 tensor read once: 537 MB) takes 0.1 ms.  Here the contraction over
// pixels is a GEMM
//     out[m = (tap, s)][n] = sum_q T[q][m] * V[q][n]          m < 36 (9 taps x <= 4 narrow channels), n = wide channel
// with T = the im2col of the NARROW tensor (gathered per 128-pixel tile into LDS planes [m][pixel], k-contiguous: the A operand
// is one ds_read_b128) and V = the WIDE tensor as stored, pixel-major, read as B operand through the transposing LDS load
// (ds_read_b64_tr_b16, bf16_frag.h).  v_mfma_f32_32x32x16_bf16, fp32 accumulation; 36 rows padded to 64 (two row blocks).
//   kind 1 (SMALL_X, conv_in) : V = dY (bf16) over output pixels, T = X (fp32, rounded to bf16 in LDS) at (y+kh-1, x+kw-1);
//                               out[co][tap][s]; the bias gradient sum_q dY[q][co] is row 36 of T set to ones
//   kind 2 (conv_out)         : V = XF(X) (bf16 storage; GroupNorm(+SiLU) applied in fp32 while staged, rounded once), T = dY
//                               (fp32) at (y+1-kh, x+1-kw); out[s][tap][ci]; bias gradient = sums of dY (from the centre tap)
// Same slab protocol as wgrad_smallk_kernel (grid = splits x 128-channel blocks, tiles of 128 linear pixels dealt in ranges,
// fixed order: deterministic), so vae_wgrad_plan / vae_reduce_splits are unchanged.  Workgroup = 4 waves, wave w owns wide
// channels 32w..32w+31 (two accumulators); two workgroups per CU cover each other's staging.
#include "bf16_frag.h"
#include <algorithm>

namespace {

constexpr int WT_TP = 128;            // pixels per tile (the contraction length of a tile)
constexpr int WT_NT = 256;
constexpr int WT_LDW = 128 + 32;      // wide image row stride (u16): 320 B, transposing reads conflict-free (bf16_frag.h)
constexpr int WT_LDT = WT_TP + 8;     // thin plane row stride (u16): 272 B
constexpr int WT_MR = 64;             // rows of T (36 used + the ones row, padded to two 32-row blocks)
constexpr int WT_ITEMS = (WT_TP * 9 + WT_NT - 1) / WT_NT;  // (pixel, tap) gathers per thread and tile: 5

template <bool SMALL_X, int XF>
__global__ __launch_bounds__(WT_NT, 2) void wgrad_thin_bf16_kernel(vae_wgrad_args p, vae_conv_geom gs, int ntiles) {
  __shared__ __attribute__((aligned(16))) u16 sW[WT_TP * WT_LDW];
  __shared__ __attribute__((aligned(16))) u16 sT[WT_MR * WT_LDT];
  __shared__ float sB[WT_NT * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const int split = blockIdx.x;
  const int c0 = blockIdx.y * 128;                   // first wide channel of this workgroup
  const int narrow = SMALL_X ? p.N : p.M;
  const u16* __restrict__ V = reinterpret_cast<const u16*>(SMALL_X ? (const void*)p.dY : (const void*)p.X);
  const float* __restrict__ S = SMALL_X ? p.X : p.dY;
  const int ldv = SMALL_X ? p.ldy : p.g.Cs, lds_ = SMALL_X ? p.g.Cs : p.ldy;
  const int npix = gs.B * gs.Ho * gs.Wo, hw = gs.Ho * gs.Wo;
  const auto rsV = VAE_BUF_RSRC(V, (size_t)npix * ldv * 2u);
  const size_t sbytes = (size_t)gs.B * gs.Hs * gs.Ws * lds_ * 4u;
  const auto rsS = VAE_BUF_RSRC(S, sbytes);

  // rows 36..63 of T never change: zeros, and (SMALL_X) row 36 = ones -> out row 36 = column sums of dY = the bias gradient
  for (int i = tid; i < (WT_MR - 36) * WT_LDT; i += WT_NT) {
    const int r = 36 + i / WT_LDT;
    sT[36 * WT_LDT + i] = (SMALL_X && r == 36) ? (u16)0x3F80 : (u16)0;
  }
  if (narrow < 4) {  // rows (tap, s >= narrow) stay zero as well
    for (int i = tid; i < 36 * WT_LDT; i += WT_NT) sT[i] = 0;
  }

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};  // kind 2: this thread's share of sum_q dY[q][s]

  // wide operand: chunk q = tid + 256 i of the tile's 2048 16-byte chunks: pixel q >> 4, channel octet tid & 15 (fixed per thread)
  const int oct = tid & 15;
  const unsigned voff0 = (unsigned)(((tid >> 4) * ldv + c0 + oct * 8) * 2);  // + (16 i + tile * 128) * ldv * 2 bytes
  uint4 rw[8];
  f32x4 rt[WT_ITEMS];
  float sc[8], sh[8];
  int xb = -1;
  auto load_tile = [&](int tile, bool valid) {
    const unsigned tb = (unsigned)tile * (WT_TP * (unsigned)ldv * 2u);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsV, valid ? voff0 + (unsigned)(16 * i * ldv * 2) : BUF_OOB, tb, 0));
    const int m0 = tile * WT_TP;
#pragma unroll
    for (int i = 0; i < WT_ITEMS; ++i) {
      const int e = tid + WT_NT * i;
      const int row = e / 9, tap = e - row * 9;
      const int m = m0 + row;
      const int b = m / hw, rem = m - b * hw;
      const int y = rem / gs.Wo, x = rem - y * gs.Wo;
      const int kh = tap / 3, kw = tap - kh * 3;
      int sy = 0, sx = 0;
      const bool ok = valid && e < WT_TP * 9 && src_pixel(gs, y, x, kh, kw, sy, sx);
      const unsigned base = oob_unless(ok, (unsigned)(((b * gs.Hs + sy) * gs.Ws + sx) * lds_) * 4u);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < narrow) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, base, 4u * s, 0));
      rt[i] = v;
    }
  };
  auto store_tile = [&](int tile) {
    if (XF != VAE_XF_NONE) {
      const int b = (tile * WT_TP) / hw;  // uniform: a tile lies inside one image (hw % 128 == 0)
      if (b != xb) {
        xb = b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sc[e] = p.scale[(int64_t)b * p.g.Cs + c0 + oct * 8 + e];
          sh[e] = p.shift[(int64_t)b * p.g.Cs + c0 + oct * 8 + e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint4 r = rw[i];
      if (XF != VAE_XF_NONE) {
        const f32x4 lo = unpack4(uint2{r.x, r.y}), hi = unpack4(uint2{r.z, r.w});
        f32x4 a, bq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = lo[e] * sc[e] + sh[e], w = hi[e] * sc[4 + e] + sh[4 + e];
          if (XF == VAE_XF_AFFINE_SILU) {
            u = silu_f(u);
            w = silu_f(w);
          }
          a[e] = u;
          bq[e] = w;
        }
        const uint2 pa = pack4(a), pb = pack4(bq);
        r = uint4{pa.x, pa.y, pb.x, pb.y};
      }
      *reinterpret_cast<uint4*>(&sW[((tid >> 4) + 16 * i) * WT_LDW + oct * 8]) = r;
    }
#pragma unroll
    for (int i = 0; i < WT_ITEMS; ++i) {
      const int e = tid + WT_NT * i;
      if (e < WT_TP * 9) {
        const int row = e / 9, tap = e - row * 9;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < narrow) sT[(tap * 4 + s) * WT_LDT + row] = __builtin_bit_cast(u16, (__bf16)rt[i][s]);
        if (!SMALL_X && tap == 4) ssum += rt[i];  // centre tap: dY at this pixel itself (fp32, before the rounding)
      }
    }
  };

  const int per = (ntiles + p.nsplit - 1) / p.nsplit;
  const int tbeg = split * per, tend = min(ntiles, tbeg + per);
  const int aoff = lr * WT_LDT + 8 * lh;                                              // + 32 mt rows, + 16 ks pixels
  const int boff = (lh * 8 + trq) * WT_LDW + wave * 32 + trh * 16 + trp * 4;           // + 16 ks pixel rows
  if (tbeg < tend) load_tile(tbeg, true);
  for (int tile = tbeg; tile < tend; ++tile) {
    __syncthreads();  // the previous tile's fragment reads are done (and, first time round, the constant rows are in place)
    store_tile(tile);
    __syncthreads();
    load_tile(tile + 1, tile + 1 < tend);  // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < WT_TP / 16; ++ks) {
      const bf16x8 fb = frag_tr(sW + boff + ks * 16 * WT_LDW, WT_LDW);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8 fa = frag_direct(sT + aoff + mt * 32 * WT_LDT + ks * 16);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[mt], 0, 0, 0);
      }
    }
  }

  // slab of this split: row m = (tap, s) of the product, column = wide channel (lane lr of wave w: c0 + 32 w + lr)
  const int64_t ld = (int64_t)9 * p.N;
  float* __restrict__ O = (p.nsplit == 1 ? p.out : p.partial + (int64_t)split * p.M * ld);
  const int ch = c0 + wave * 32 + lr;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int t = m >> 2, s = m & 3;
      if (m < 36 && s < narrow) {
        if (SMALL_X) O[(int64_t)ch * ld + (int64_t)t * p.N + s] = p.alpha * acc[mt][r];
        else O[(int64_t)s * ld + (int64_t)t * p.N + ch] = p.alpha * acc[mt][r];
      }
      if (SMALL_X && m == 36 && p.bias_partial) p.bias_partial[(int64_t)split * p.M + ch] = acc[mt][r];
    }
  if (!SMALL_X && p.bias_partial && blockIdx.y == 0) {  // sums of dY: 256 threads' shares, fixed order
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) sB[s * WT_NT + tid] = ssum[s];
    __syncthreads();
    if (tid < p.M) {
      float t = 0.f;
      for (int i = 0; i < WT_NT; ++i) t += sB[tid * WT_NT + i];
      p.bias_partial[(int64_t)split * p.M + tid] = t;
    }
  }
}

}  // namespace

// after wgrad_canon; kind = wgrad_smallk_kind(a) (1: the narrow side is X, 2: the narrow side is dY)
bool wgrad_thin_bf16_eligible(const vae_wgrad_args& a, int kind) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || kind == 0 || a.batch != 1 || a.X16 != nullptr || a.dY16 != nullptr) return false;
  if (g.mode != VAE_MODE_FWD || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (((int64_t)g.Ho * g.Wo) % WT_TP != 0) return false;  // whole 128-pixel tiles, each inside one image
  if (kind == 1) {
    if (!a.y_bf16 || a.x_bf16 || a.xf != VAE_XF_NONE || a.M % 128 != 0 || a.ldy % 8 != 0 || !aligned16(a.dY)) return false;
    if ((size_t)a.npix * a.ldy * 2u >= BUF_MAX || (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX) return false;
  } else {
    if (!a.x_bf16 || a.y_bf16 || a.N % 128 != 0 || g.Cs % 8 != 0 || !aligned16(a.X)) return false;
    if ((size_t)a.npix * g.Cs * 2u >= BUF_MAX || (size_t)a.npix * a.ldy * 4u >= BUF_MAX) return false;
  }
  return true;
}

int launch_wgrad_thin_bf16(const vae_wgrad_args& a, int kind, const vae_conv_geom& gs, int ntiles, hipStream_t st) {
  const int wide = kind == 1 ? a.M : a.N;
  dim3 grid((unsigned)a.nsplit, (unsigned)(wide / 128));
#define WTK(SX, XFV) hipLaunchKernelGGL((wgrad_thin_bf16_kernel<SX, XFV>), grid, dim3(WT_NT), 0, st, a, gs, ntiles)
  if (kind == 1) WTK(true, VAE_XF_NONE);
  else if (a.xf == VAE_XF_NONE) WTK(false, VAE_XF_NONE);
  else if (a.xf == VAE_XF_AFFINE) WTK(false, VAE_XF_AFFINE);
  else WTK(false, VAE_XF_AFFINE_SILU);
#undef WTK
  return 0;
}
