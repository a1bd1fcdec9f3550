// bf16 mode: 3x3 convolutions whose CONTRACTION side has <= 4 channels (encoder.conv_in forward: 3 -> 128; the dgrad of
// decoder.conv_out: 3 -> 128) on the matrix pipe.  conv_smallk_kernel (skinny.hip) runs them on the VALU, one lane per output
// channel and 2-byte stores of the bf16 output: 0.5 ms per launch at 256x256, batch 32, for a 537 MB write that takes 0.1 ms.
//     out[px][n] = bias[n] + sum_{tap,s} S[src(px,tap)][s] * W[n][tap][s]       k = (tap, s): 36 values, padded to 48
// Workgroup = 4 waves over tiles of 128 linear output pixels x 128 output channels; the im2col of the narrow tensor is gathered
// per tile into LDS ([pixel][48 k], bf16, k-contiguous: the A operand is one ds_read_b128), the weights are rounded to bf16 once
// per workgroup and stay in registers as B operands (12 fragments), v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The output
// tile goes through LDS and leaves as whole 256-byte rows (16 bytes per lane).  The tracker sums (|value as stored| per channel
// and 128-row tile, the layout of conv_smallk_kernel) come from the accumulators.
#include "bf16_frag.h"
#include <algorithm>

namespace {

constexpr int CT_TP = 128, CT_NT = 256;
constexpr int CT_LDK = 48 + 8;         // im2col row stride (u16): 112 B
constexpr int CT_LDO = 128 + 8;        // output tile row stride (u16): 272 B
constexpr int CT_ITEMS = (CT_TP * 9 + CT_NT - 1) / CT_NT;  // (pixel, tap) gathers per thread and tile: 5

__global__ __launch_bounds__(CT_NT, 2) void conv_thin_bf16_kernel(vae_igemm_args p, int ntiles, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) u16 sA[CT_TP * CT_LDK];   // im2col of the tile (14,336 B)
  __shared__ __attribute__((aligned(16))) u16 sO[CT_TP * CT_LDO];   // output tile; first the weights [128 n][48 k] (34,816 B)
  __shared__ float sTr[4 * 128];
  const vae_conv_geom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 128;
  const int hw = g.Ho * g.Wo;
  const size_t sbytes = (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u;
  const auto rsS = VAE_BUF_RSRC(p.A, sbytes);
  const auto rsC = VAE_BUF_RSRC(p.C, (size_t)p.M * p.ldc * 2u);

  // weights -> bf16 [n][k = tap * 4 + s] in LDS (over sO), then into registers: B fragment of (k-step ks, channel block nb) =
  // 8 consecutive k of column n0 + 32 nb + lr
  for (int i = tid; i < 128 * 48; i += CT_NT) {
    const int n = i / 48, k = i - n * 48, t = k >> 2, s = k & 3;
    const float w = (k < 36 && s < p.K) ? p.W[(int64_t)(n0 + n) * p.sn + (int64_t)t * p.st + (int64_t)s * p.sk] : 0.f;
    sO[n * CT_LDK + k] = __builtin_bit_cast(u16, (__bf16)w);
  }
  // the k columns (tap, s >= K) and 36..47 of the im2col rows stay zero
  for (int i = tid; i < CT_TP * CT_LDK; i += CT_NT) sA[i] = 0;
  __syncthreads();
  bf16x8 fb[3][4];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) fb[ks][nb] = frag_direct(sO + (nb * 32 + lr) * CT_LDK + ks * 16 + 8 * lh);
  float bv[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bv[nb] = p.bias ? p.bias[n0 + nb * 32 + lr] : 0.f;

  f32x4 rt[CT_ITEMS];
  auto gather = [&](int tile, bool valid) {
    const int m0 = tile * CT_TP;
#pragma unroll
    for (int i = 0; i < CT_ITEMS; ++i) {
      const int e = tid + CT_NT * i;
      const int row = e / 9, tap = e - row * 9;
      const int m = m0 + row;
      const int b = m / hw, rem = m - b * hw;
      const int y = rem / g.Wo, x = rem - y * g.Wo;
      const int kh = tap / 3, kw = tap - kh * 3;
      int sy = 0, sx = 0;
      const bool ok = valid && e < CT_TP * 9 && src_pixel(g, y, x, kh, kw, sy, sx);
      const unsigned base = oob_unless(ok, (unsigned)(((b * g.Hs + sy) * g.Ws + sx) * g.Cs) * 4u);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < p.K) v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, base, 4u * s, 0));
      rt[i] = v;
    }
  };

  const int tbeg = blockIdx.x * tiles_per_wg, tend = min(ntiles, tbeg + tiles_per_wg);
  if (tbeg < tend) gather(tbeg, true);
  for (int tile = tbeg; tile < tend; ++tile) {
    __syncthreads();  // the previous tile's reads of sA / sO are done (first time: the weight fragments are in registers)
#pragma unroll
    for (int i = 0; i < CT_ITEMS; ++i) {
      const int e = tid + CT_NT * i;
      if (e < CT_TP * 9) {
        const int row = e / 9, tap = e - row * 9;
        *reinterpret_cast<uint2*>(&sA[row * CT_LDK + tap * 4]) = pack4(rt[i]);
      }
    }
    __syncthreads();
    gather(tile + 1, tile + 1 < tend);  // in flight under the MFMAs and the output pass below

    // wave w: pixels 32 w .. 32 w + 31 of the tile x 128 channels
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = bv[nb];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const bf16x8 fa = frag_direct(sA + (wave * 32 + lr) * CT_LDK + ks * 16 + 8 * lh);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[ks][nb], acc[nb], 0, 0, 0);
    }
    // accumulator (row = pixel 4 lh + 8 (r >> 2) + (r & 3), column = channel lr) -> bf16 tile in LDS; tracker sums on the way
    float tsum[4] = {0.f, 0.f, 0.f, 0.f};
    u16* const orow = sO + (wave * 32 + 4 * lh) * CT_LDO + lr;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const __bf16 h = (__bf16)acc[nb][r];
        orow[(8 * (r >> 2) + (r & 3)) * CT_LDO + nb * 32] = __builtin_bit_cast(u16, h);
        tsum[nb] += fabsf((float)h);  // the tracker describes the tensor as stored
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (p.track) {  // uniform
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float t2 = tsum[nb] + lane_xor32(tsum[nb]);
        if (lh == 0) sTr[wave * 128 + nb * 32 + lr] = t2;
      }
    }
    __syncthreads();
    // 128 rows x 256 B: 16 bytes per lane, 16 lanes per row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = tid + CT_NT * i, row = q >> 4, c8 = q & 15;
      const uint4 v = *reinterpret_cast<const uint4*>(&sO[row * CT_LDO + c8 * 8]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (unsigned)((row * p.ldc + n0 + c8 * 8) * 2),
                                             (unsigned)tile * (unsigned)(CT_TP * p.ldc * 2), 0);
    }
    if (p.track && tid < 128)
      p.track[(int64_t)tile * p.N + n0 + tid] = ((sTr[tid] + sTr[128 + tid]) + sTr[256 + tid]) + sTr[384 + tid];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// <= 4 OUTPUT channels (decoder.conv_out: 128 -> 3 behind GroupNorm + SiLU) on the matrix pipe.  conv_smalln_kernel (skinny.hip)
// walks the 9 x 128 window of every output pixel on the VALU: 0.83 ms at 256x256, batch 32 for a 537 MB read.  Here a workgroup
// (4 waves) takes a 4 x 32-pixel tile: the 6 x 34 halo of the bf16 input is transformed ONCE per element (fp32 GroupNorm(+SiLU),
// rounded to bf16) into LDS, pixel-major with the channels contiguous, so the A operand of v_mfma_f32_16x16x32_bf16 (16 pixels of
// a tile row x 32 channels of one tap) is one ds_read_b128; the weights W[s][tap][c] are k-contiguous as stored and sit in LDS
// as bf16 rows (the 13 unused output columns read a zero row).  Wave w = tile row w: 2 x 36 MFMAs.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CN_TH = 4, CN_TW = 32, CN_HW = CN_TW + 2, CN_HP = (CN_TH + 2) * CN_HW;  // 204 halo pixels
constexpr int CN_KMAX = 128, CN_LDH = CN_KMAX + 8;                                      // halo row stride (u16): 272 B
constexpr int CN_LDWT = 9 * CN_KMAX + 8;                                               // weight row stride (u16)

template <int XF>
__global__ __launch_bounds__(256, 2) void conv_thinn_bf16_kernel(vae_igemm_args p, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) u16 sH[CN_HP * CN_LDH];     // 55,488 B
  __shared__ __attribute__((aligned(16))) u16 sWt[4 * CN_LDWT];       // 9,280 B: rows 0..2 = W[s], row 3 = zeros
  const vae_conv_geom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * CN_TH, x0 = tx * CN_TW;
  const int K = p.K, nq = K / 8;  // channel octets per pixel
  const auto rsA = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.A) + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 2u);

  for (int i = tid; i < 4 * 9 * K; i += 256) {
    const int s = i / (9 * K), r = i - s * 9 * K, tap = r / K, c = r - tap * K;
    const float w = s < p.N ? p.W[(int64_t)s * p.sn + (int64_t)tap * p.st + c] : 0.f;
    sWt[s * CN_LDWT + tap * K + c] = __builtin_bit_cast(u16, (__bf16)w);
  }
  // halo: chunk q = (pixel, octet); a thread's octet is fixed when nq divides 256 (K = 128: 16 octets), so its 8 scale / shift
  // values stay in registers
  const int oct = tid % nq;
  float sc[8], sh[8];
  if (XF != VAE_XF_NONE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = p.scale[(int64_t)b * g.Cs + oct * 8 + e];
      sh[e] = p.shift[(int64_t)b * g.Cs + oct * 8 + e];
    }
  }
  const int nchunks = CN_HP * nq;
  for (int q0 = 0; q0 < nchunks; q0 += 256 * 4) {
    uint4 r[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + tid + 256 * i, pp = q / nq;
      const int ir = pp / CN_HW, jc = pp - ir * CN_HW;
      const int hy = y0 - 1 + ir, hx = x0 - 1 + jc;
      ok[i] = q < nchunks && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws);
      r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oob_unless(ok[i], (unsigned)(((hy * g.Ws + hx) * g.Cs + oct * 8) * 2)), 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + tid + 256 * i;
      if (q < nchunks) {
        uint4 v = r[i];
        if (XF != VAE_XF_NONE) {
          const f32x4 lo = unpack4(uint2{v.x, v.y}), hi = unpack4(uint2{v.z, v.w});
          f32x4 a, c;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = lo[e] * sc[e] + sh[e], w = hi[e] * sc[4 + e] + sh[4 + e];
            if (XF == VAE_XF_AFFINE_SILU) {
              u = silu_f(u);
              w = silu_f(w);
            }
            a[e] = ok[i] ? u : 0.f;  // padding stays zero AFTER the transform
            c[e] = ok[i] ? w : 0.f;
          }
          const uint2 pa = pack4(a), pc = pack4(c);
          v = uint4{pa.x, pa.y, pc.x, pc.y};
        }
        *reinterpret_cast<uint4*>(&sH[(q / nq) * CN_LDH + oct * 8]) = v;
      }
    }
  }
  __syncthreads();

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const u16* wrow = sWt + min(l16, 3) * CN_LDWT + 8 * kq;
  const int ksteps = K / 32;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const u16* hrow = sH + ((wave + kh) * CN_HW + l16 + kw) * CN_LDH + 8 * kq;
    for (int cq = 0; cq < ksteps; ++cq) {
      const bf16x8 fb = frag_direct(wrow + tap * K + cq * 32);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8 fa = frag_direct(hrow + mt * 16 * CN_LDH + cq * 32);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[mt], 0, 0, 0);
      }
    }
  }
  // D: column = output channel l16, rows = pixels 4 kq + r of the 16-pixel block
  if (l16 < p.N) {
    const float bv = p.bias ? p.bias[l16] : 0.f;
    float* __restrict__ C = p.C + ((int64_t)(b * g.Ho + y0 + wave) * g.Wo + x0) * p.ldc + l16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(int64_t)(mt * 16 + 4 * kq + r) * p.ldc] = acc[mt][r] + bv;
  }
}

}  // namespace

// after rows_canon: a launch conv_smallk_kernel would serve, with the output stored as bf16
bool conv_thin_bf16_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || !a.out_bf16 || a.a_bf16 || a.A16 != nullptr || a.res != nullptr || a.xf != VAE_XF_NONE) return false;
  if (a.K > 4 || a.batch != 1 || a.alpha != 1.0f || g.taps != 9 || g.stride != 1 || a.gstat || a.gnb_ws) return false;
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (a.N % 128 != 0 || a.ldc % 8 != 0 || a.M % CT_TP != 0 || !aligned16(a.C)) return false;
  if ((size_t)a.M * a.ldc * 2u >= BUF_MAX || (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX) return false;
  return true;
}

int launch_conv_thin_bf16(const vae_igemm_args& a, hipStream_t st) {
  const int ntiles = a.M / CT_TP;
  const int per = std::max(1, std::min(8, ntiles / 1024));  // a few tiles per workgroup: the weight fragments are built once
  dim3 grid((unsigned)((ntiles + per - 1) / per), (unsigned)(a.N / 128));
  hipLaunchKernelGGL(conv_thin_bf16_kernel, grid, dim3(CT_NT), 0, st, a, ntiles, per);
  return 0;
}

// after rows_canon: a launch conv_smalln_kernel would serve, with the input stored as bf16 (a_bf16) and at most 128 channels
bool conv_thinn_bf16_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || !a.a_bf16 || a.out_bf16 || a.A16 != nullptr || a.res != nullptr || a.track != nullptr) return false;
  if (a.N > 4 || a.K % 32 != 0 || a.K > CN_KMAX || (256 % (a.K / 8)) != 0 || a.batch != 1 || a.alpha != 1.0f || a.sk != 1) return false;
  if (g.mode != VAE_MODE_FWD || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.Wo % CN_TW != 0 || g.Ho % CN_TH != 0 || g.Cs % 8 != 0 || !aligned16(a.A) || a.gstat || a.gnb_ws) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 2u >= BUF_MAX) return false;
  return true;
}

int launch_conv_thinn_bf16(const vae_igemm_args& a, hipStream_t st) {
  const int tx = a.g.Wo / CN_TW, ty = a.g.Ho / CN_TH;
  dim3 grid((unsigned)((int64_t)tx * ty * a.g.B));
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((conv_thinn_bf16_kernel<VAE_XF_NONE>), grid, dim3(256), 0, st, a, tx, ty); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((conv_thinn_bf16_kernel<VAE_XF_AFFINE>), grid, dim3(256), 0, st, a, tx, ty); break;
    default: hipLaunchKernelGGL((conv_thinn_bf16_kernel<VAE_XF_AFFINE_SILU>), grid, dim3(256), 0, st, a, tx, ty); break;
  }
  return 0;
}
