// fp32 weight gradient of an Upsample2D block's convolution, conv3x3(nearest_upsample_2x(x)), with 9 multiplications per
// LOW-resolution pixel and (ci, co) pair (direct: 36; the four 2x2 phase weight gradients of round 1: 16).  Companion of
// conv3_upwino.hip; exact fp32 products on v_mfma_f32_32x32x2_f32.
//
// F(3x3, 2x2) on the 2x2 block of dY that belongs to low-resolution pixel (i, j) reads the 4x4 patch of the UPSAMPLED input with
// origin (2i-1, 2j-1), whose rows are x[i-1], x[i], x[i], x[i+1]: row 2 of B^T d vanishes, so only the positions {0, 1, 3}^2 of the
// transform domain carry anything.  With the factor 2 of position 1 moved into G nothing is halved:
//     dW = A''^T [ sum_pixels (G'' dy G''^T) (.) (L d3 L^T) ] A''     d3: 3x3 low-resolution patch of x, dy: 2x2 block of dY
//     L = [1 -1 0; 0 1 0; 0 1 -1]     G'' = [1 0; 1 1; 0 1]     A''^T = [1 1 0; 0 1 0; 0 1 -1]
// (row 0 of dW: dy0 (x[i-1] - x[i]) + (dy0 + dy1) x[i] = dy0 x[i-1] + dy1 x[i]: tap kh = 0 meets x[i-1] from the even output row
// and x[i] from the odd one.)  Per position a GEMM  M_p[ci][co] = sum_pixels V_p[pixel][ci] * D_p[pixel][co].
//
// Kernel: as wgrad3_wino.hip -- a unit = 8 low-resolution pixels of one row (a 2 x 16 strip of dY); per unit the 3 x 10-pixel halo
// of x and the strip of dY go to LDS untransformed, transposed to [row][channel][x]; a wave builds its MFMA operands while
// reading them; split-K slabs [split][9][Cin][Cout]; a reduction kernel sums the slabs in fixed order, applies A''^T . A'' and
// writes OHWI.  Workgroup = 8 waves = 32 ci x 128 co x 9 positions: wave w owns position w (all four 32-channel blocks of co), and
// the NINTH position is split by channel block over waves 0..3 -- one per SIMD (wave k of a workgroup runs on SIMD k % 4), so every
// SIMD carries 36 MFMAs per step.  (A first version with one position per wave and 9 waves put three waves of every workgroup
// on SIMD 0: 0.51 of the matrix peak.)  One workgroup per CU at 158 registers (a 128-register build spilled 27, with scratch reloads
// inside the loop); two staging register sets, the next-but-one unit requested behind the first channel block's MFMAs.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int GCI = 32, GCO = 128, NPOS = 9, GNT = 512, GNB = GCO / 32;
constexpr int XSX = 12;                 // x pitch of a staged halo row (floats): 10 columns + padding (16-byte reads at 0 / 4 / 8)
constexpr int XSY = 20;                 // x pitch of a staged dY row: 16 columns + padding
constexpr int SXF = 3 * GCI * XSX;      // halo stage: [3 rows][32 ci][XSX]
constexpr int SYF = 2 * GCO * XSY;      // dY stage:   [2 rows][128 co][XSY]
constexpr int GSTAGE = SXF + SYF;       // 6272 floats (25088 B)

__global__ __launch_bounds__(GNT, 2) void wgrad3_upwino_kernel(vae_wgrad_args p, int strips, int64_t nunits) {
  __shared__ __attribute__((aligned(16))) float smem[2 * GSTAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // wave = position (pr, pc) 0..7; position 8: block `wave` on waves 0..3
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;  // Hs x Ws: the low-resolution x; Ho x Wo = 2 Hs x 2 Ws: dY
  const int tilesN = p.N / GCI, ntile = tilesN * (p.M / GCO);
  int tile, split;
  {  // the tiles of one split walk through the same pixels: ids congruent mod 8 share an XCD's L2 (see wgrad3_wino.hip)
    const int L = blockIdx.x, ns = p.nsplit;
    if (ns % 8 == 0) {
      const int j = L >> 3;
      tile = j % ntile;
      split = (j / ntile) * 8 + (L & 7);
    } else {
      tile = L % ntile;
      split = L / ntile;
    }
  }
  const int tm = tile / tilesN, tn = tile % tilesN;
  const int m0 = tm * GCO, n0 = tn * GCI;
  const int64_t per = (nunits + p.nsplit - 1) / p.nsplit;
  const int64_t ubeg = split * per, uend = min(nunits, ubeg + per);
  const int nu = (int)max((int64_t)0, uend - ubeg);
  const int upi = g.Hs * strips;  // units per image
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0;

  // ---- staging roles.  dY strip (all 512 threads): column xx, channel quad yq, both rows.  x halo (threads 0..239): row xr,
  // column xc of the 3 x 10 halo, channel quad xq (fastest: 16-byte neighbours in memory) ----
  const int xx = tid & 15, yq = tid >> 4;
  const bool xrole = tid < 240;
  const int xq = tid & 7, xc = (tid >> 3) % 10, xr = tid / 80;
  const auto rsX = VAE_BUF_RSRC(p.X, (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u);
  const auto rsY = VAE_BUF_RSRC(p.dY, (size_t)g.B * g.Ho * g.Wo * p.ldy * 4u);
  // TWO sets of staging registers (the kernel runs at a 256-register budget): unit k+2 is requested during step k, behind its
  // first block's MFMAs, and stored at the end of step k+1 -- two steps (~3 us) of lead; with one set requested at the start
  // of a step and stored at its end the store sat on the HBM round trip
  struct Stg {
    f32x4 rx, ry[2];
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  Stg s0{z4, {z4, z4}}, s1{z4, {z4, z4}};
  f32x4 bsum = z4;
  // the unit the next load_unit call requests (calls go through ubeg, ubeg+1, ...: counters instead of divisions per step)
  int ub = (int)(ubeg / upi), urow = (int)((ubeg - (int64_t)ub * upi) / strips), ustrip = (int)((ubeg - (int64_t)ub * upi) % strips);
  const unsigned rowY = (unsigned)g.Wo * p.ldy * 4u;
  auto load_unit = [&](int k, Stg& r) {  // requests for unit ubeg + k (beyond the range: nothing is read, zeros)
    const bool ok = k < nu;
    const int b = ub, i = urow, x0 = ustrip * 8;
    if (++ustrip == strips) {
      ustrip = 0;
      if (++urow == g.Hs) {
        urow = 0;
        ++ub;
      }
    }
    const int hy = i - 1 + xr, hx = x0 - 1 + xc;
    const bool xin = ok && xrole && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws);
    r.rx = VAE_BUF_LOAD4(rsX, xin ? (unsigned)(((b * g.Hs + hy) * g.Ws + hx) * g.Cs + n0 + 4 * xq) * 4u : BUF_OOB);  // (unsigned before the * 4: descriptors reach 4 GiB)
    const unsigned baseY = (unsigned)((b * g.Ho + 2 * i) * g.Wo + 2 * x0 + xx) * (unsigned)p.ldy * 4u + (unsigned)(m0 + 4 * yq) * 4u;
#pragma unroll
    for (int a = 0; a < 2; ++a) r.ry[a] = VAE_BUF_LOAD4(rsY, ok ? baseY + a * rowY : BUF_OOB);
  };
  auto store_unit = [&](float* st, const Stg& r) {
    float* sx = st;
    float* sy = st + SXF;
    if (xrole) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sx[(xr * GCI + 4 * xq + e) * XSX + xc] = r.rx[e];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sy[(a * GCO + 4 * yq + e) * XSY + xx] = r.ry[a][e];
      if (do_bias) bsum += r.ry[a];
    }
  };

  f32x16 acc[GNB], accx;  // accx: position 8, channel block `wave` (waves 0..3)
#pragma unroll
  for (int nb = 0; nb < GNB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) accx[e] = 0.f;

  load_unit(0, s0);
  store_unit(smem, s0);
  load_unit(1, s1);
  __syncthreads();

  // operand builders for position (pr, pc).  V fragment: lane (ci = lr, half lh) takes tiles 4 lh + e: halo columns 4 lh .. 4 lh + 5
  // of the rows the position combines (rows 0, 1, 2 of the halo = x[i-1], x[i], x[i+1]: pr 0: r0 - r1, 1: r1, 2: r1 - r2).
  // D fragment of channel block nb: tiles 4 lh + e = dY columns 8 lh + 2 e, + 1 (pr 0: dy row 0, 1: row 0 + row 1, 2: row 1).
  auto build_a = [&](auto PR, auto PC, const float* cx, f32x4& a4) {
    constexpr int pr = decltype(PR)::value, pc = decltype(PC)::value;
    constexpr int vr1 = pr == 0 ? 0 : 1, vr2 = pr == 0 ? 1 : 2;
    const int voff1 = (vr1 * GCI + lr) * XSX + 4 * lh, voff2 = (vr2 * GCI + lr) * XSX + 4 * lh;
    float rc[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(&cx[voff1 + 4 * c]);
      if (pr == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rc[4 * c + e] = u1[e];
      } else {
        const f32x4 u2 = *reinterpret_cast<const f32x4*>(&cx[voff2 + 4 * c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) rc[4 * c + e] = u1[e] - u2[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = pc == 0 ? rc[e] - rc[e + 1] : (pc == 1 ? rc[e + 1] : rc[e + 1] - rc[e + 2]);
  };
  auto build_b = [&](auto PR, auto PC, const float* cy, int nb, f32x4& b4) {
    constexpr int pr = decltype(PR)::value, pc = decltype(PC)::value;
    constexpr int dr1 = pr == 2 ? 1 : 0;
    const int doff = (dr1 * GCO + nb * 32 + lr) * XSY + 8 * lh;
    float rc[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(&cy[doff + 4 * c]);
      if (pr == 1) {
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(&cy[doff + GCO * XSY + 4 * c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) rc[4 * c + e] = d0[e] + d1[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) rc[4 * c + e] = d0[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[e] = pc == 0 ? rc[2 * e] : (pc == 1 ? rc[2 * e] + rc[2 * e + 1] : rc[2 * e + 1]);
  };
  using I2 = std::integral_constant<int, 2>;

  // The loop body is instantiated per position: straight-line code, no per-step branches.
  auto run = [&](auto PR, auto PC, auto EXTRA) {
    constexpr bool extra = decltype(EXTRA)::value;  // this wave also owns block `wave` of position 8 = (2, 2)
    auto step = [&](int k, const Stg& cur, Stg& nxt) {  // cur: unit k+1 (requested during step k-1); nxt receives unit k+2
      const float* cx = smem + (k & 1) * GSTAGE;
      const float* cy = cx + SXF;
      float* nst = smem + ((k + 1) & 1) * GSTAGE;
      f32x4 a4, bb[2];
      build_a(PR, PC, cx, a4);
      build_b(PR, PC, cy, 0, bb[0]);
#pragma unroll
      for (int nb = 0; nb < GNB; ++nb) {
        __builtin_amdgcn_sched_barrier(0);
        if (nb == 1) {  // the requests for unit k+2 (index counters + 3 loads) behind the first block's MFMAs
          load_unit(k + 2, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (nb + 1 < GNB) build_b(PR, PC, cy, nb + 1, bb[(nb + 1) & 1]);  // the next block's operands, while this block's MFMAs issue
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bb[nb & 1][e], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (extra) {
        f32x4 ax, bx;
        build_a(I2{}, I2{}, cx, ax);
        build_b(I2{}, I2{}, cy, wave, bx);
#pragma unroll
        for (int e = 0; e < 4; ++e) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], accx, 0, 0, 0);
      }
      store_unit(nst, cur);
      __syncthreads();
    };
    int k = 0;
    for (; k + 1 < nu; k += 2) {
      step(k, s1, s0);
      step(k + 1, s0, s1);
    }
    if (k < nu) step(k, s1, s0);
  };
  {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F_ = std::false_type;
    switch (wave) {  // (wave-uniform: pr = wave / 3, pc = wave % 3; waves 0..3 carry the ninth position's blocks)
      case 0: run(I0{}, I0{}, T{}); break;
      case 1: run(I0{}, I1{}, T{}); break;
      case 2: run(I0{}, I2{}, T{}); break;
      case 3: run(I1{}, I0{}, T{}); break;
      case 4: run(I1{}, I1{}, F_{}); break;
      case 5: run(I1{}, I2{}, F_{}); break;
      case 6: run(I2{}, I0{}, F_{}); break;
      default: run(I2{}, I1{}, F_{}); break;
    }
  }

  // ---- epilogue: slab [split][9 positions][Cin][Cout]; lanes along co (128-byte rows) ----
  float* __restrict__ O = p.partial + (int64_t)split * NPOS * p.N * p.M;
#pragma unroll
  for (int nb = 0; nb < GNB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = n0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      O[((int64_t)wave * p.N + ci) * p.M + m0 + nb * 32 + lr] = acc[nb][e];
    }
  if (wave < 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = n0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      O[((int64_t)8 * p.N + ci) * p.M + m0 + wave * 32 + lr] = accx[e];
    }
  }
  if (do_bias) {  // workgroup-uniform: thread sums of its channel quad -> over the 16 columns of the strip
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = bsum[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
      if (xx == 0) p.bias_partial[(int64_t)split * p.M + m0 + 4 * yq + e] = sv;
    }
  }
}

// dW[co][a][b][ci] = sum_{p,q} At[a][p] At[b][q] sum_split slab[split][p*3+q][ci][co], At = A''^T = [1 1 0; 0 1 0; 0 1 -1]; 32 x 32
// (ci, co) tile per workgroup, read with lanes along co, written with lanes along ci.  Workgroups beyond the tiles reduce the
// bias-gradient slab.
__global__ __launch_bounds__(256) void upwino_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int N, int M, float* __restrict__ dW,
                                                                  int tiles, const float* __restrict__ bpart, float* __restrict__ db) {
  __shared__ float sT[9][32][33];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= tiles) {
    const int m = ((int)blockIdx.x - tiles) * 256 + tid;
    if (m < M) {
      float s = 0.f;
      for (int k = 0; k < nsplit; ++k) s += bpart[(int64_t)k * M + m];
      db[m] = s;
    }
    return;
  }
  const int tilesM = M / 32;
  const int c0 = ((int)blockIdx.x / tilesM) * 32, m0 = ((int)blockIdx.x % tilesM) * 32;
  const int64_t pstride = (int64_t)N * M, sstride = NPOS * pstride;
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int cl = (tid >> 5) + 8 * r, ml = tid & 31;
    const float* src = slab + (int64_t)(c0 + cl) * M + m0 + ml;
    float mm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float s = 0.f;
      for (int k = 0; k < nsplit; ++k) s += src[(int64_t)k * sstride + q * pstride];
      mm[q] = s;
    }
    float h[3][3];  // A''^T M
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h[0][j] = mm[0 * 3 + j] + mm[1 * 3 + j];
      h[1][j] = mm[1 * 3 + j];
      h[2][j] = mm[1 * 3 + j] - mm[2 * 3 + j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sT[a * 3 + 0][ml][cl] = h[a][0] + h[a][1];
      sT[a * 3 + 1][ml][cl] = h[a][1];
      sT[a * 3 + 2][ml][cl] = h[a][1] - h[a][2];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int ml = (tid >> 5) + 8 * r, cl = tid & 31;
#pragma unroll
    for (int t = 0; t < 9; ++t) dW[((int64_t)(m0 + ml) * 9 + t) * N + c0 + cl] = sT[t][ml][cl];
  }
}

}  // namespace

// conv3x3(nearest_upsample_2x(x)) in fp32: geometry mode UP2X (source = the low-resolution x, row grid = dY at twice the size),
// 8 | low-resolution width, 32 | Cin, 128 | Cout
bool wgrad3_upwino_eligible(const vae_wgrad_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_F32 || a.X16 != nullptr || a.dY16 != nullptr || a.dY == nullptr || a.batch != 1 || a.alpha != 1.0f) return false;
  if (a.x_bf16 || a.y_bf16 || a.xf != VAE_XF_NONE) return false;
  if (g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.mode != VAE_MODE_UP2X) return false;
  if (a.tapmask != 0 || a.y_step > 1 || g.Ho != 2 * g.Hs || g.Wo != 2 * g.Ws) return false;
  if (g.Ws % 8 != 0 || a.N % GCI != 0 || a.M % GCO != 0 || g.Cs % 4 != 0 || a.ldy % 4 != 0 || g.Cs < a.N) return false;
  if (!aligned16(a.X) || !aligned16(a.dY)) return false;
  if ((size_t)g.B * g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.B * g.Ho * g.Wo * a.ldy * 4u >= BUF_MAX) return false;
  return true;
}

int64_t wgrad3_upwino_units(const vae_conv_geom& g) { return (int64_t)g.B * g.Hs * (g.Ws / 8); }

int launch_wgrad3_upwino(const vae_wgrad_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int64_t nunits = wgrad3_upwino_units(g);
  dim3 grid((unsigned)((a.M / GCO) * (a.N / GCI) * a.nsplit), 1, 1);
  hipLaunchKernelGGL(wgrad3_upwino_kernel, grid, dim3(GNT), 0, st, a, g.Ws / 8, nunits);
  return 0;
}

int launch_upwino_wgrad_reduce(const float* slab, int nsplit, int N, int M, float* dW, const float* bpart, float* db, hipStream_t st) {
  const int tiles = (N / 32) * (M / 32);
  const int extra = bpart ? (M + 255) / 256 : 0;
  hipLaunchKernelGGL(upwino_wgrad_reduce_kernel, dim3((unsigned)(tiles + extra)), dim3(256), 0, st, slab, nsplit, N, M, dW, tiles, bpart, db);
  return 0;
}
