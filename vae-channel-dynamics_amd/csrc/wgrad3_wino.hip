// Winograd F(3x3, 2x2) for the fp32 weight gradient of the 3x3 stride-1 layers: the 3x3 gradient of a (ci, co) pair over a 2x2
// block of output pixels needs 16 multiplications instead of 36 (2.25x fewer MFMA passes); exact fp32 products on
// v_mfma_f32_32x32x2_f32, fp32 accumulation over all tiles in the transform domain.
//     dW = A^T [ sum_tiles (G dy G^T) (.) (B^T d B) ] A      d: 4x4 patch of XF(X) (origin 2t-1), dy: 2x2 block of dY, dW: 3x3
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0; .5 .5; .5 -.5; 0 1]   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1]
// Per position p = (i, j) of the 4x4 transform domain this is a GEMM  M_p[ci][co] = sum_tile V_p[tile][ci] * D_p[tile][co].
//   * Workgroup (8 waves) = 32 ci x 128 co x all 16 positions over a range of units; unit = a 2 x 16-pixel strip of the output
//     (8 tiles = the K-step).  Per unit the 4 x 18-pixel halo of X (GroupNorm + SiLU applied once per element) and the 2 x 16 x 128
//     strip of dY go to LDS *untransformed* but transposed to [row][channel][x]: 30 KB per step, double buffered, one barrier.
//   * Wave w owns positions (i = w/2, j = 2(w%2), 2(w%2)+1) and builds its MFMA operands while reading them: lane (channel, half)
//     reads the 12 / 8 contiguous x values its four tiles need from the 2 (or 1) rows that position row i combines
//     (conflict-free 16-byte reads at a row pitch of 20 floats) and forms the row / column combinations in registers
//     (18 additions for the V fragment pair, <= 12 per channel block for D).  No transformed image ever exists in LDS.
//     The halves of G are left out (D' = G' dy G'^T with G' = [1 0; 1 1; 1 -1; 0 1]) and applied as exact power-of-two
//     factors c_i c_j, c = (1, .5, .5, 1), by the reduction.
//   * Split-K over unit ranges: a workgroup writes its 16 x 32 x 128 accumulators into a slab [split][16][Cin][Cout];
//     wino_wgrad_reduce_kernel sums the slabs in fixed order (deterministic), applies A^T . A and writes OHWI through an LDS
//     transpose; the bias gradient (column sums of dY) rides along as in the direct kernel.
// Numerics: measured against the direct kernel in tests/test_kernels_gpu.py (the transform-domain sums cancel in the output
// transform, so the error is a few 1e-6 of the gradient scale instead of 1e-7; the parity bar is 1e-4).
#include "common.h"
#ifndef VAE_ABLATE
#define VAE_ABLATE 0  // diagnostic builds (tools/ablation_builds.sh, wrong results): bit 0 no global loads, 1 no LDS stores, 4 no barrier
#endif
#include <algorithm>
#include <type_traits>

namespace {

constexpr int GCI = 32, GCO = 128, GNT = 512;
constexpr int XS = 20;                  // x pitch of a staged row (floats): 18 halo / 16 strip columns + padding
constexpr int SXF = 4 * GCI * XS;       // halo stage: [4 rows][32 ci][XS]
constexpr int SYF = 2 * GCO * XS;       // dY stage:   [2 rows][128 co][XS]
constexpr int GSTAGE = SXF + SYF;       // 7680 floats (30720 B)

template <int XF>
__global__ __launch_bounds__(GNT, 1) void wgrad3_wino_kernel(vae_wgrad_args p, int strips, int64_t nunits) {
  __shared__ __attribute__((aligned(16))) float smem[2 * GSTAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = p.N / GCI, ntile = tilesN * (p.M / GCO);
  // Workgroup id -> (tile, split).  Hardware deals consecutive ids round-robin over the 8 XCDs (one L2 each); the tiles of one
  // split walk through the SAME pixels (every ci block re-reads the dY strip, every co block the X halo), so they are given
  // ids congruent mod 8: one L2 (or 8 / nsplit of them) fetches a split's rows once.
  int tile, split;
  {
    const int L = blockIdx.x, ns = p.nsplit;
    if (ns % 8 == 0) {
      const int j = L >> 3;
      tile = j % ntile;
      split = (j / ntile) * 8 + (L & 7);
    } else if ((ns == 2 || ns == 4) && ntile % (8 / ns) == 0) {
      split = (L & 7) % ns;
      tile = (L >> 3) * (8 / ns) + (L & 7) / ns;
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
  const int upi = (g.Ho / 2) * strips;  // units per image
  const bool do_bias = (p.bias_partial != nullptr) && tn == 0;

  // ---- staging roles.  X halo: thread -> (row xr, channel quad xq, column xx) of the 4 x 16 main block, threads with
  // (tid & 8) == 0 && tid < 128 also one element of the two extra columns (same channel quad: shared GroupNorm rows).
  // Lanes run along x (16) and 4 channel quads: the four 4-byte LDS writes of a loaded float4 are conflict-free ----
  const int xx = tid & 15, xq = (tid >> 4) & 7, xr = tid >> 7;
  const bool xe_role = tid < 128 && (tid & 8) == 0;
  const int ex = 16 + (tid & 1), er = (tid >> 1) & 3;
  // dY strip: thread -> (column xx, channel quad yq) for both rows
  const int yq = tid >> 4;
  const auto rsX = VAE_BUF_RSRC(p.X, (size_t)g.B * g.Hs * g.Ws * g.Cs * 4u);
  const auto rsY = VAE_BUF_RSRC(p.dY, (size_t)g.B * g.Ho * g.Wo * p.ldy * 4u);
  // Two sets of staging registers: the requests for unit k+2 go out at the start of step k (into the set unit k used) and are
  // stored during step k+1, so they have a whole step in flight wherever in the step a wave does its stores.
  struct Stg {
    f32x4 rx, rxe, ry[2];
    int b;  // image of the unit (workgroup-uniform)
    bool xin, xein;
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  Stg s0{z4, z4, {z4, z4}, -1, false, false}, s1 = s0;
  f32x4 bsum = z4;
  // GroupNorm rows of the thread's channel quad for image sc_b: one set, re-read when a stored unit belongs to the next image
  // (a workgroup's unit range crosses few image boundaries; the re-read waits for L2 once per boundary)
  f32x4 rsc = {1.f, 1.f, 1.f, 1.f}, rsh = z4;
  int sc_b = -1;
  // the unit the next load_unit call requests (calls go through ubeg, ubeg+1, ...: counters instead of divisions per step)
  int ub = (int)(ubeg / upi), uty = (int)((ubeg - (int64_t)ub * upi) / strips), ustrip = (int)((ubeg - (int64_t)ub * upi) % strips);
  // Addresses = a workgroup-uniform part (scalar registers) + a per-thread constant; an element is padding when one of the
  // thread's position flags meets the unit's border flag (bit 0: first halo row / top border, 1: last halo row / bottom border,
  // 2: first halo column / left border, 3: last halo column / right border, 4: always / unit beyond the range).
  const unsigned thrX = ((xr * g.Ws + xx) * g.Cs + n0 + 4 * xq) * 4u, thrE = ((er * g.Ws + ex) * g.Cs + n0 + 4 * xq) * 4u;
  const unsigned thrY = (xx * p.ldy + m0 + 4 * yq) * 4u, rowY = (unsigned)g.Wo * p.ldy * 4u;
  const int tfX = (xr == 0 ? 1 : 0) | (xr == 3 ? 2 : 0) | (xx == 0 ? 4 : 0) | 16;
  const int tfE = xe_role ? ((er == 0 ? 1 : 0) | (er == 3 ? 2 : 0) | (ex == 17 ? 8 : 0) | 16) : 31;
  auto load_unit = [&](int k, Stg& r) {  // requests for unit ubeg + k (beyond the range: nothing is read, zeros)
    const bool ok = k < nu;
    const int b = ub, ty = uty, x0 = ustrip * 16;
    if (++ustrip == strips) {
      ustrip = 0;
      if (++uty == g.Ho / 2) {
        uty = 0;
        ++ub;
      }
    }
    r.b = ok ? b : -1;
    const int uf = (ty == 0 ? 1 : 0) | (2 * ty + 2 >= g.Hs ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + 16 >= g.Ws ? 8 : 0) | (ok ? 0 : 16);
    // (the halo origin may lie before the tensor: unsigned arithmetic wraps, the sum with the thread's part is exact)
    const unsigned baseX = (unsigned)((b * g.Hs + 2 * ty - 1) * g.Ws + x0 - 1) * (unsigned)g.Cs * 4u;
    const unsigned baseY = (unsigned)((b * g.Ho + 2 * ty) * g.Wo + x0) * (unsigned)p.ldy * 4u;
    r.xin = (tfX & uf) == 0;
    r.rx = VAE_BUF_LOAD4(rsX, r.xin ? baseX + thrX : BUF_OOB);
    r.xein = (tfE & uf) == 0;
    r.rxe = VAE_BUF_LOAD4(rsX, r.xein ? baseX + thrE : BUF_OOB);
#pragma unroll
    for (int a = 0; a < 2; ++a) r.ry[a] = VAE_BUF_LOAD4(rsY, ok ? baseY + a * rowY + thrY : BUF_OOB);
  };
  auto xform = [&](f32x4 v, bool in) {
    if (XF != VAE_XF_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * rsc[e] + rsh[e];
        if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
        v[e] = in ? u : 0.f;  // padding stays zero AFTER the transform
      }
    }
    return v;
  };
  auto store_unit = [&](float* st, const Stg& r) {
    float* sx = st;
    float* sy = st + SXF;
    if (XF != VAE_XF_NONE && r.b >= 0 && r.b != sc_b) {  // workgroup-uniform
      sc_b = r.b;
      rsc = *reinterpret_cast<const f32x4*>(p.scale + (int64_t)sc_b * g.Cs + n0 + 4 * xq);
      rsh = *reinterpret_cast<const f32x4*>(p.shift + (int64_t)sc_b * g.Cs + n0 + 4 * xq);
    }
    const f32x4 v = xform(r.rx, r.xin);
#pragma unroll
    for (int e = 0; e < 4; ++e) sx[(xr * GCI + 4 * xq + e) * XS + xx] = v[e];
    if (xe_role) {
      const f32x4 ve = xform(r.rxe, r.xein);
#pragma unroll
      for (int e = 0; e < 4; ++e) sx[(er * GCI + 4 * xq + e) * XS + ex] = ve[e];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sy[(a * GCO + 4 * yq + e) * XS + xx] = r.ry[a][e];
      if (do_bias) bsum += r.ry[a];
    }
  };

  // ---- operand roles: wave -> row i of the transform domain and the column pair (2 jh, 2 jh + 1) ----
  const int wi = wave >> 1, jh = wave & 1;
  // V row combination: B^T row i = x[r1] + sg * x[r2]
  const int vr1 = wi == 0 ? 0 : (wi == 2 ? 2 : 1), vr2 = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
  const int voff1 = (vr1 * GCI + lr) * XS + 8 * lh, voff2 = (vr2 * GCI + lr) * XS + 8 * lh;
  // D' row combination: i = 0: dy row 0, 3: row 1, 1: row0 + row1, 2: row0 - row1
  const int dr1 = wi == 3 ? 1 : 0;

  f32x16 acc[2][4];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pi][nb][e] = 0.f;

  load_unit(0, s0);
  store_unit(smem, s0);
  load_unit(1, s1);
  __syncthreads();

  // The loop body is instantiated per (two-row D combination?, column pair): straight-line code, no per-step branches.
  auto run = [&](auto WI, auto JH) {
    constexpr int wic = decltype(WI)::value, jhc = decltype(JH)::value;
    constexpr bool two = wic == 1 || wic == 2;  // D' row i combines both dY rows
    constexpr bool vplus = wic == 1;            // sign of the second row in the combinations (V: B^T row i, D': G' row i)
    auto step = [&](int k, const Stg& cur, Stg& nxt) {  // cur: unit k+1 (requested during step k-1); nxt receives unit k+2
      const float* cx = smem + (k & 1) * GSTAGE;
      const float* cy = cx + SXF;
      float* nst = smem + ((k + 1) & 1) * GSTAGE;
      // D' operands: raw dY rows of channel block nb -> b4 of the two positions.  The operands of block nb+1 are formed and
      // the rows of block nb+2 requested WHILE the MFMAs of block nb issue (interleaved below): the two waves of a SIMD
      // take turns on the matrix pipe and so drift into the same phase -- operand building in a phase of its own would leave
      // the pipe idle in both at once.
      f32x4 rd[2][2];  // [row][x quad]
      auto read_d = [&](int nb, f32x4 (&d)[2][2]) {
        const int doff = (dr1 * GCO + nb * 32 + lr) * XS + 8 * lh;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          d[0][c] = *reinterpret_cast<const f32x4*>(&cy[doff + 4 * c]);
          if (two) d[1][c] = *reinterpret_cast<const f32x4*>(&cy[doff + GCO * XS + 4 * c]);
        }
      };
      auto build_b = [&](const f32x4 (&d)[2][2], f32x4 (&b4)[2]) {
        float rc[8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) rc[4 * c + e] = !two ? d[0][c][e] : (vplus ? d[0][c][e] + d[1][c][e] : d[0][c][e] - d[1][c][e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (jhc == 0) {
            b4[0][e] = rc[2 * e];
            b4[1][e] = rc[2 * e] + rc[2 * e + 1];
          } else {
            b4[0][e] = rc[2 * e] - rc[2 * e + 1];
            b4[1][e] = rc[2 * e + 1];
          }
        }
      };
      // V fragments of the two positions: 12 x values of the two rows, combined
      f32x4 a4[2], bb[2][2];
      {
        f32x4 u1[3], u2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          u1[c] = *reinterpret_cast<const f32x4*>(&cx[voff1 + 4 * c]);
          u2[c] = *reinterpret_cast<const f32x4*>(&cx[voff2 + 4 * c]);
        }
        float rc[12];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) rc[4 * c + e] = vplus ? u1[c][e] + u2[c][e] : u1[c][e] - u2[c][e];
        __builtin_amdgcn_sched_barrier(0);
        read_d(0, rd);  // (after the row combination: its 24 input registers are free again)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (jhc == 0) {
            a4[0][e] = rc[2 * e] - rc[2 * e + 2];
            a4[1][e] = rc[2 * e + 1] + rc[2 * e + 2];
          } else {
            a4[0][e] = rc[2 * e + 2] - rc[2 * e + 1];
            a4[1][e] = rc[2 * e + 1] - rc[2 * e + 3];
          }
        }
        build_b(rd, bb[0]);
        read_d(1, rd);
      }
      if (!(VAE_ABLATE & 2) && wave < 4) store_unit(nst, cur);  // the two waves of a SIMD store at opposite ends of the step
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        __builtin_amdgcn_sched_barrier(0);
        if (nb == 1) {  // the requests for unit k+2 (index counters + 4 loads) behind the first block's MFMAs, not in front of the
          // step, where both waves of a SIMD build operands and the matrix pipe has nothing to do (2.1-2.7 % of the kernel)
          if (!(VAE_ABLATE & 1)) load_unit(k + 2, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (nb == 3 && wave >= 4) {  // (waves 4..7: the LDS stores of unit k+1 behind the third block's MFMAs, not between the last MFMA and the barrier)
          if (!(VAE_ABLATE & 2)) store_unit(nst, cur);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (nb < 3) build_b(rd, bb[(nb + 1) & 1]);  // block nb+1's operands, during the first half of this block's MFMAs
        if (nb < 2) read_d(nb + 2, rd);             // block nb+2's rows into the same registers, during the second half
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[pi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[pi][e], bb[nb & 1][pi][e], acc[pi][nb], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(VAE_ABLATE & 16)) __syncthreads();
    };
    int k = 0;
    for (; k + 1 < nu; k += 2) {
      step(k, s1, s0);
      step(k + 1, s0, s1);
    }
    if (k < nu) step(k, s1, s0);
  };
  {
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    switch (wave) {  // (wave-uniform: wi = wave / 2, jh = wave % 2)
      case 0: run(std::integral_constant<int, 0>{}, J0{}); break;
      case 1: run(std::integral_constant<int, 0>{}, J1{}); break;
      case 2: run(std::integral_constant<int, 1>{}, J0{}); break;
      case 3: run(std::integral_constant<int, 1>{}, J1{}); break;
      case 4: run(std::integral_constant<int, 2>{}, J0{}); break;
      case 5: run(std::integral_constant<int, 2>{}, J1{}); break;
      case 6: run(std::integral_constant<int, 3>{}, J0{}); break;
      default: run(std::integral_constant<int, 3>{}, J1{}); break;
    }
  }

  // ---- epilogue: slab [split][16 positions][Cin][Cout]; lanes along co (128-byte rows) ----
  float* __restrict__ O = p.partial + (int64_t)split * 16 * p.N * p.M;
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    const int pos = wi * 4 + 2 * jh + pi;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = n0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        O[((int64_t)pos * p.N + ci) * p.M + m0 + nb * 32 + lr] = acc[pi][nb][e];
      }
  }
  if (do_bias) {  // workgroup-uniform: thread sums of its channel quad -> over the 16 columns of the strip
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = bsum[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      if (xx == 0) p.bias_partial[(int64_t)split * p.M + m0 + 4 * yq + e] = s;
    }
  }
}

// dW[co][a][b][ci] = sum_{i,j} At[a][i] At[b][j] c_i c_j sum_split slab[split][i*4+j][ci][co]; 32 x 32 (ci, co) tile per workgroup,
// read with lanes along co, written with lanes along ci.  Workgroups beyond the tiles reduce the bias-gradient slab.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int N, int M, float* __restrict__ dW,
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
  const int64_t pstride = (int64_t)N * M, sstride = 16 * pstride;
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int cl = (tid >> 5) + 8 * r, ml = tid & 31;
    const float* src = slab + (int64_t)(c0 + cl) * M + m0 + ml;
    float mm[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float s = 0.f;
      for (int k = 0; k < nsplit; ++k) s += src[(int64_t)k * sstride + q * pstride];
      const float ci = ((q >> 2) == 1 || (q >> 2) == 2) ? 0.5f : 1.f, cj = ((q & 3) == 1 || (q & 3) == 2) ? 0.5f : 1.f;
      mm[q] = s * (ci * cj);
    }
    float h[3][4];  // A^T M
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[0][j] = (mm[0 * 4 + j] + mm[1 * 4 + j]) + mm[2 * 4 + j];
      h[1][j] = mm[1 * 4 + j] - mm[2 * 4 + j];
      h[2][j] = (mm[1 * 4 + j] + mm[2 * 4 + j]) - mm[3 * 4 + j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sT[a * 3 + 0][ml][cl] = (h[a][0] + h[a][1]) + h[a][2];
      sT[a * 3 + 1][ml][cl] = h[a][1] - h[a][2];
      sT[a * 3 + 2][ml][cl] = (h[a][1] + h[a][2]) - h[a][3];
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

// plain 3x3 stride-1 pad-1 layer in fp32 with 16-pixel strips, 32 | Cin, 128 | Cout
bool wgrad3_wino_eligible(const vae_wgrad_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_F32 || a.X16 != nullptr || a.dY16 != nullptr || a.dY == nullptr || a.batch != 1 || a.alpha != 1.0f) return false;
  if (g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.mode != VAE_MODE_FWD) return false;
  if (a.tapmask != 0 || a.y_step > 1 || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.Ho % 2 != 0 || g.Wo % 16 != 0 || a.N % GCI != 0 || a.M % GCO != 0 || g.Cs % 4 != 0 || a.ldy % 4 != 0) return false;
  if (!aligned16(a.X) || !aligned16(a.dY)) return false;
  if ((size_t)g.B * g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.B * g.Ho * g.Wo * a.ldy * 4u >= BUF_MAX) return false;
  return true;
}

int64_t wgrad3_wino_units(const vae_conv_geom& g) { return (int64_t)g.B * (g.Ho / 2) * (g.Wo / 16); }

int launch_wgrad3_wino(const vae_wgrad_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int64_t nunits = wgrad3_wino_units(g);
  dim3 grid((unsigned)((a.M / GCO) * (a.N / GCI) * a.nsplit), 1, 1);
  const int strips = g.Wo / 16;
  if (a.xf == VAE_XF_NONE) hipLaunchKernelGGL(wgrad3_wino_kernel<VAE_XF_NONE>, grid, dim3(GNT), 0, st, a, strips, nunits);
  else if (a.xf == VAE_XF_AFFINE) hipLaunchKernelGGL(wgrad3_wino_kernel<VAE_XF_AFFINE>, grid, dim3(GNT), 0, st, a, strips, nunits);
  else hipLaunchKernelGGL(wgrad3_wino_kernel<VAE_XF_AFFINE_SILU>, grid, dim3(GNT), 0, st, a, strips, nunits);
  return 0;
}

int launch_wino_wgrad_reduce(const float* slab, int nsplit, int N, int M, float* dW, const float* bpart, float* db, hipStream_t st) {
  const int tiles = (N / 32) * (M / 32);
  const int extra = bpart ? (M + 255) / 256 : 0;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)(tiles + extra)), dim3(256), 0, st, slab, nsplit, N, M, dW, tiles, bpart, db);
  return 0;
}
