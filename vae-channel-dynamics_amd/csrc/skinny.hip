// Convolutions with a <= 4-channel side (conv_in: 3 input channels; conv_out: 3 output channels; the 4-channel latent
// convs) on the VALU.  On the MFMA tiles those sides are padded to 32 and 7/8 of the matrix work multiplies zeros; at
// 256x256 the five launches cost 6.7 ms.  They are HBM-bound by nature (one 128-channel tensor is read or written
// once: 0.1-0.2 ms at the full rate), so here one lane owns one channel of the WIDE side and keeps its 9 x (<=4)
// weights / accumulators in registers, and the few values of the NARROW side that a pixel's 9 taps need are staged per
// 128-pixel tile in LDS ([128 px][9 taps][4], 18 KB) and read back as broadcast ds_read_b128.  Wide-side global
// accesses are 512-byte rows per pixel; every load goes through a buffer descriptor (out of range = 0).
//   conv_smallk : y[px][n]      = bias[n] + sum_{tap,s} S[src(px,tap)][s] * W[n][tap][s]      (rows form, K <= 4)
//   wgrad_smallk: out[..]       = sum_px V[px][lane] * S[src(px,tap)][s]                      (N <= 4 or M <= 4)
//   conv_smalln : y[px][s]      = bias[s] + sum_{tap,c} XF(x)[px+tap][c] * W[s][tap][c]       (<= 4 outputs; end of file)
// The WIDE side may be stored as bf16 (bf16 mode keeps the 128-channel activations / gradients as bf16): out_bf16 for
// conv_smallk's output, y_bf16 / x_bf16 for wgrad_smallk's V, a_bf16 for conv_smalln's input; the narrow side, the weights
// and the arithmetic are fp32.
#include "bf16_frag.h"
#include <algorithm>

namespace {

constexpr int TP = 128;   // pixels per tile
constexpr int NT = 256;   // 2 pixel streams x 128 channel lanes
constexpr int TAPS_MAX = 9;

// stage S[src(row, tap)][0..3] for the TP rows of tile `m0` (row = linear pixel of the row grid of `g`)
__device__ __forceinline__ void stage_small(const vae_conv_geom& g, const float* __restrict__ S, int Cs_small, int K, int m0,
                                            int mrows, f32x4* __restrict__ sI) {
  const int hw = g.Ho * g.Wo;
  const size_t sbytes = (size_t)g.B * g.Hs * g.Ws * Cs_small * 4u;
  const auto rsS = VAE_BUF_RSRC(S, sbytes < BUF_MAX ? sbytes : BUF_MAX);
  for (int e = threadIdx.x; e < TP * g.taps; e += NT) {
    const int row = e / g.taps, tap = e - row * g.taps;
    const int m = m0 + row;
    const int b = m / hw, rem = m - b * hw;
    const int y = rem / g.Wo, x = rem - y * g.Wo;
    const int kh = (g.taps == 9) ? tap / 3 : 0, kw = (g.taps == 9) ? tap - kh * 3 : 0;
    int sy = 0, sx = 0;
    const bool ok = (m < mrows) && src_pixel(g, y, x, kh, kw, sy, sx);
    const unsigned base = ok ? (unsigned)(((b * g.Hs + sy) * g.Ws + sx) * Cs_small) * 4u : BUF_OOB;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < K) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, ok ? base + 4u * s : BUF_OOB, 0, 0));
    sI[e] = v;
  }
}

__global__ __launch_bounds__(NT) void conv_smallk_kernel(vae_igemm_args p) {
  __shared__ f32x4 sI[TP * TAPS_MAX];
  const vae_conv_geom g = p.g;
  const int tid = threadIdx.x, lane = tid & 127, half = tid >> 7;
  const int m0 = blockIdx.x * TP;
  const int n = blockIdx.y * 128 + lane;
  const bool nok = n < p.N;
  stage_small(g, p.A, g.Cs, p.K, m0, p.M, sI);

  float w[TAPS_MAX][4];
#pragma unroll
  for (int t = 0; t < TAPS_MAX; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      w[t][s] = (nok && t < g.taps && s < p.K) ? p.W[(int64_t)n * p.sn + (int64_t)t * p.st + (int64_t)s * p.sk] : 0.f;
  const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
  const bool cbf = p.out_bf16 != 0;
  const size_t obytes = (size_t)p.M * p.ldc * (cbf ? 2u : 4u);
  const auto rsC = VAE_BUF_RSRC(p.C, obytes);
  __syncthreads();

  float tsum = 0.f;
#pragma unroll 4
  for (int i = 0; i < TP / 2; ++i) {
    const int row = half + 2 * i;
    float acc = bv;
#pragma unroll
    for (int t = 0; t < TAPS_MAX; ++t) {
      if (t < g.taps) {  // uniform
        const f32x4 v = sI[row * g.taps + t];  // same address in every lane: broadcast read
#pragma unroll
        for (int s = 0; s < 4; ++s) acc += v[s] * w[t][s];
      }
    }
    const int m = m0 + row;
    const bool ok = nok && m < p.M;
    buf_store1_elem(rsC, cbf, ok ? m * p.ldc + n : -1, acc);
    const float stored = cbf ? (float)(__bf16)acc : acc;  // the tracker describes the tensor as stored
    tsum += ok ? fabsf(stored) : 0.f;
  }
  if (p.track) {  // uniform: per-channel sum of |y| over this 128-row tile (same partial layout as the MFMA kernels)
    __syncthreads();
    float* red = reinterpret_cast<float*>(sI);
    red[tid] = tsum;
    __syncthreads();
    if (half == 0 && nok) p.track[(int64_t)blockIdx.x * p.N + n] = red[lane] + red[128 + lane];
  }
}

// SMALL_X: the narrow side is X (N <= 4): V = dY over output pixels, S = X gathered with g, out[m = lane][tap][s]
// else   : the narrow side is dY (M <= 4): V = XF(X) over INPUT pixels, S = dY gathered with the transposed geometry
//          (the output pixel whose tap lands on this input pixel), out[s][tap][n = lane]
template <bool SMALL_X, int XF>
__global__ __launch_bounds__(NT) void wgrad_smallk_kernel(vae_wgrad_args p, vae_conv_geom gs, int ntiles) {
  __shared__ f32x4 sI[TP * TAPS_MAX + 32];  // + 128 floats for the bias sums of the final reduction
  const int tid = threadIdx.x, lane = tid & 127, half = tid >> 7;
  const int split = blockIdx.x;
  const int ch = blockIdx.y * 128 + lane;            // channel of the wide side
  const int wide = SMALL_X ? p.M : p.N, narrow = SMALL_X ? p.N : p.M;
  const bool chok = ch < wide;
  const float* __restrict__ V = SMALL_X ? p.dY : p.X;
  const float* __restrict__ S = SMALL_X ? p.X : p.dY;
  const int ldv = SMALL_X ? p.ldy : p.g.Cs, lds_ = SMALL_X ? p.g.Cs : p.ldy;
  const int npixv = gs.B * gs.Ho * gs.Wo;            // pixels of V (= row grid of gs)
  const int hwv = gs.Ho * gs.Wo;
  const bool vbf = (SMALL_X ? p.y_bf16 : p.x_bf16) != 0;  // storage of the wide operand
  const size_t vbytes = (size_t)npixv * ldv * (vbf ? 2u : 4u);
  const auto rsV = VAE_BUF_RSRC(V, vbytes < BUF_MAX ? vbytes : BUF_MAX);

  float acc[TAPS_MAX][4];
#pragma unroll
  for (int t = 0; t < TAPS_MAX; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t][s] = 0.f;
  float vsum = 0.f;                                  // SMALL_X: bias gradient of this lane's channel
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f};                 // !SMALL_X: bias gradient = centre-tap sums of dY (same in every lane)
  int xb = -1;
  float xsc = 1.f, xsh = 0.f;

  const int per = (ntiles + p.nsplit - 1) / p.nsplit;
  const int tbeg = split * per, tend = min(ntiles, tbeg + per);
  for (int tile = tbeg; tile < tend; ++tile) {
    const int m0 = tile * TP;
    __syncthreads();  // the previous tile's reads of sI are done
    stage_small(gs, S, lds_, narrow, m0, npixv, sI);
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < TP / 2; ++i) {
      const int row = half + 2 * i;
      const int m = m0 + row;
      const bool ok = chok && m < npixv;
      float v = buf_load1_elem(rsV, vbf, ok ? m * ldv + ch : -1);
      if (XF != VAE_XF_NONE) {
        const int b = min(m, npixv - 1) / hwv;  // uniform per row
        if (b != xb) {
          xb = b;
          xsc = chok ? p.scale[(int64_t)b * p.g.Cs + ch] : 0.f;
          xsh = chok ? p.shift[(int64_t)b * p.g.Cs + ch] : 0.f;
        }
        float u = v * xsc + xsh;
        if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
        v = ok ? u : 0.f;
      }
      if (SMALL_X) vsum += v;
#pragma unroll
      for (int t = 0; t < TAPS_MAX; ++t) {
        if (t < gs.taps) {  // uniform
          const f32x4 sv = sI[row * gs.taps + t];
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[t][s] += v * sv[s];
          if (!SMALL_X && t == gs.taps / 2) ssum += sv;  // centre tap: the output pixel at this input pixel
        }
      }
    }
  }

  // the two pixel streams of a lane are added in a fixed order, then the slab is written
  __syncthreads();
  float* red = reinterpret_cast<float*>(sI);  // [128][36] + 4
  if (half == 1) {
#pragma unroll
    for (int t = 0; t < TAPS_MAX; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) red[lane * 36 + t * 4 + s] = acc[t][s];
    if (SMALL_X) red[128 * 36 + lane] = vsum;
    else if (lane == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) red[128 * 36 + s] = ssum[s];
    }
  }
  __syncthreads();
  if (half == 0) {
    const int64_t ld = (int64_t)gs.taps * p.N;
    float* __restrict__ O = (p.nsplit == 1 ? p.out : p.partial + (int64_t)split * p.M * ld);
    if (chok) {
#pragma unroll
      for (int t = 0; t < TAPS_MAX; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (t < gs.taps && s < narrow) {
            const float r = acc[t][s] + red[lane * 36 + t * 4 + s];
            if (SMALL_X) O[(int64_t)ch * ld + (int64_t)t * p.N + s] = p.alpha * r;
            else O[(int64_t)s * ld + (int64_t)t * p.N + ch] = p.alpha * r;
          }
    }
    if (p.bias_partial) {
      if (SMALL_X) {  // every channel block owns its channels' sums of dY
        if (chok) p.bias_partial[(int64_t)split * p.M + ch] = vsum + red[128 * 36 + lane];
      } else if (blockIdx.y == 0 && lane < p.M) {
        p.bias_partial[(int64_t)split * p.M + lane] = ssum[lane] + red[128 * 36 + lane];
      }
    }
  }
}

}  // namespace

bool conv_smallk_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  return a.K <= 4 && a.batch == 1 && a.xf == VAE_XF_NONE && a.res == nullptr && a.alpha == 1.0f && g.taps <= TAPS_MAX &&
         g.mode != VAE_MODE_DGRAD_S2 && (size_t)a.M * a.ldc * 4u < BUF_MAX && (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u < BUF_MAX;
}
int launch_conv_smallk(const vae_igemm_args& a, hipStream_t st) {
  dim3 grid((unsigned)((a.M + TP - 1) / TP), (unsigned)((a.N + 127) / 128));
  hipLaunchKernelGGL(conv_smallk_kernel, grid, dim3(NT), 0, st, a);
  return 0;
}

// 0 = not served here, 1 = the narrow side is X (N <= 4), 2 = the narrow side is dY (M <= 4)
int wgrad_smallk_kind(const vae_wgrad_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.batch != 1 || g.taps > TAPS_MAX) return 0;
  if ((size_t)a.npix * a.ldy * 4u >= BUF_MAX || (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX) return 0;
  if (a.N <= 4 && a.xf == VAE_XF_NONE && (g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_UP2X)) return 1;
  // transposed gather: plain stride-1 'same' geometry only (every input pixel is the centre tap of one output pixel)
  if (a.M <= 4 && g.mode == VAE_MODE_FWD && g.stride == 1 && g.Ho == g.Hs && g.Wo == g.Ws &&
      ((g.taps == 9 && g.pad_t == 1 && g.pad_l == 1) || (g.taps == 1 && g.pad_t == 0 && g.pad_l == 0)))
    return 2;
  return 0;
}
int wgrad_smallk_tiles(const vae_wgrad_args& a) { return (int)(((int64_t)a.g.B * a.g.Ho * a.g.Wo + TP - 1) / TP); }
bool wgrad_thin_bf16_eligible(const vae_wgrad_args& a, int kind);  // wgrad_thin_bf16.hip: the same launches on the bf16 matrix pipe
int launch_wgrad_thin_bf16(const vae_wgrad_args& a, int kind, const vae_conv_geom& gs, int ntiles, hipStream_t st);
bool wgrad_smallk_on_mfma(const vae_wgrad_args& a) { return wgrad_thin_bf16_eligible(a, wgrad_smallk_kind(a)) && !vae_opt().no_thin_mfma; }

int launch_wgrad_smallk(const vae_wgrad_args& a, hipStream_t st) {
  const int kind = wgrad_smallk_kind(a);
  const int ntiles = wgrad_smallk_tiles(a);
  vae_conv_geom gs = a.g;  // how a pixel of V and a tap map to a pixel of S
  if (kind == 2) {         // V = X over input pixels; S = dY at (y + pad - kh, x + pad - kw)
    gs.mode = VAE_MODE_DGRAD;
    gs.Cs = a.ldy;
  }
  if (wgrad_smallk_on_mfma(a)) return launch_wgrad_thin_bf16(a, kind, gs, ntiles, st);  // bf16 mode, wide side stored as bf16
  const int wide = kind == 1 ? a.M : a.N;
  dim3 grid((unsigned)a.nsplit, (unsigned)((wide + 127) / 128));
#define WSK(SX, XFV) hipLaunchKernelGGL((wgrad_smallk_kernel<SX, XFV>), grid, dim3(NT), 0, st, a, gs, ntiles)
  if (kind == 1) WSK(true, VAE_XF_NONE);
  else if (a.xf == VAE_XF_NONE) WSK(false, VAE_XF_NONE);
  else if (a.xf == VAE_XF_AFFINE) WSK(false, VAE_XF_AFFINE);
  else WSK(false, VAE_XF_AFFINE_SILU);
#undef WSK
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// conv_smalln: 3x3 stride-1 'same' convolution with <= 4 OUTPUT channels (decoder.conv_out: 128 -> 3), optionally
// with GroupNorm(+SiLU) fused on the input.  One lane owns one output pixel of a 4x32-pixel tile and walks the
// 9 x C input values of its window: activations come from an LDS halo (the staging applies the transform once per
// halo element), weights are wave-uniform and are read through the scalar cache into SGPRs (v_fma with an SGPR
// operand), so the VALU does nothing but the 9*C*N useful FMAs per pixel.
// ---------------------------------------------------------------------------------------------------------
namespace {

constexpr int NTH = 4, NTW = 32, NHW = NTW + 2, NHP = (NTH + 2) * NHW;  // 4x32 tile, 204 halo pixels
constexpr int NBK = 32, NLD = NBK + 4, NNT = 128;
constexpr int NHQ = NHP * (NBK / 4), NHI = (NHQ + NNT - 1) / NNT;       // 1632 float4 slots, 13 per thread

template <int XF>
__global__ __launch_bounds__(NNT) void conv_smalln_kernel(vae_igemm_args p, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float sH[NHP * NLD];
  const vae_conv_geom g = p.g;
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * NTH, x0 = tx * NTW;
  const int pr = tid >> 5, px = tid & 31;  // this lane's pixel of the tile
  const bool abf = p.a_bf16 != 0;
  const unsigned esA = abf ? 2u : 4u;
  const auto rsA = VAE_BUF_RSRC(reinterpret_cast<const char*>(p.A) + (int64_t)b * g.Hs * g.Ws * g.Cs * esA, (size_t)g.Hs * g.Ws * g.Cs * esA);
  const float* __restrict__ W = p.W;

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int hk4 = tid & 7;  // the thread's 4 channels are the same for all of its halo slots
  uint4 rh[NHI];  // as loaded (fp32 quad or 4 bf16): converted when the halo is stored
  int hmask = 0;
  f32x4 rsc = {0.f, 0.f, 0.f, 0.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  auto load_halo = [&](int c0) {
    hmask = 0;
    const int c = c0 + hk4 * 4;
    if (XF != VAE_XF_NONE) {
      const int cs = min(c, p.K - 4);
      rsc = *reinterpret_cast<const f32x4*>(p.scale + (int64_t)b * g.Cs + cs);
      rsh = *reinterpret_cast<const f32x4*>(p.shift + (int64_t)b * g.Cs + cs);
    }
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = tid + NNT * i;
      const int pp = q >> 3;
      const int ir = pp / NHW, jc = pp - ir * NHW;
      const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
      const bool ok = (q < NHQ) && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws) && (c < p.K);
      rh[i] = buf_load4_raw(rsA, esA, ok ? ((hy * g.Ws + hx) * g.Cs + c) : -1);
      hmask |= (ok ? 1 : 0) << i;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = tid + NNT * i;
      if (q < NHQ) {
        f32x4 v = raw4_to_f32(rh[i], abf);
        if (XF != VAE_XF_NONE) {  // padding must stay zero after the transform
          const bool ok = (hmask >> i) & 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = v[e] * rsc[e] + rsh[e];
            if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
            v[e] = ok ? u : 0.f;
          }
        }
        *reinterpret_cast<f32x4*>(&sH[(q >> 3) * NLD + hk4 * 4]) = v;
      }
    }
  };

  const int kchunks = (p.K + NBK - 1) / NBK;
  load_halo(0);
  for (int cch = 0; cch < kchunks; ++cch) {
    const int c0 = cch * NBK;
    __syncthreads();  // every lane has left the previous chunk's halo
    store_halo();
    __syncthreads();
    if (cch + 1 < kchunks) load_halo(c0 + NBK);  // in flight during this chunk's FMAs
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const float* hrow = &sH[((pr + kh) * NHW + px + kw) * NLD];
#pragma unroll
      for (int q = 0; q < NBK / 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(hrow + q * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s < p.N) {  // uniform
            // wave-uniform address: the compiler reads these through the scalar cache
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)s * p.sn + (int64_t)tap * p.st + c0 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[s] = __builtin_fmaf(a[e], w4[e], acc[s]);
          }
        }
      }
    }
  }

  const int oy = y0 + pr, ox = x0 + px;
  const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)b * g.Ho * g.Wo * p.ldc, (size_t)g.Ho * g.Wo * p.ldc * 4u);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < p.N) {
      const float v = acc[s] + (p.bias ? p.bias[s] : 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC,
                                            (oy < g.Ho && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + s) * 4) : BUF_OOB, 0, 0);
    }
}

}  // namespace

bool conv_smalln_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  return a.N <= 4 && a.K >= NBK && a.K % NBK == 0 && a.sn % 4 == 0 && a.st % 4 == 0 && aligned16(a.W) && a.batch == 1 && a.res == nullptr && a.track == nullptr && a.alpha == 1.0f &&
         a.A16 == nullptr && g.mode == VAE_MODE_FWD && g.taps == 9 && g.stride == 1 && g.pad_t == 1 && g.pad_l == 1 &&
         g.Ho == g.Hs && g.Wo == g.Ws && g.Wo % NTW == 0 && g.Ho % NTH == 0 && a.sk == 1 && g.Cs % 4 == 0 && aligned16(a.A) &&
         (a.xf == VAE_XF_NONE || (aligned16(a.scale) && aligned16(a.shift))) &&
         (size_t)g.Hs * g.Ws * g.Cs * 4u < BUF_MAX && (size_t)g.Ho * g.Wo * a.ldc * 4u < BUF_MAX;
}
int launch_conv_smalln(const vae_igemm_args& a, hipStream_t st) {
  const int tx = a.g.Wo / NTW, ty = a.g.Ho / NTH;
  dim3 grid((unsigned)((int64_t)tx * ty * a.g.B));
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((conv_smalln_kernel<VAE_XF_NONE>), grid, dim3(NNT), 0, st, a, tx, ty); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((conv_smalln_kernel<VAE_XF_AFFINE>), grid, dim3(NNT), 0, st, a, tx, ty); break;
    default: hipLaunchKernelGGL((conv_smalln_kernel<VAE_XF_AFFINE_SILU>), grid, dim3(NNT), 0, st, a, tx, ty); break;
  }
  return 0;
}
