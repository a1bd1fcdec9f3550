// fp32 forward and dgrad of an Upsample2D block's convolution, conv3x3(nearest_upsample_2x(x)), as a Winograd-type minimal
// filtering scheme with 9 multiplications per LOW-resolution pixel and channel pair (direct: 36; the four 2x2 phase
// convolutions of round 1: 16).  Exact fp32 products on v_mfma_f32_32x32x2_f32 like every fp32 contraction here.
//
// Forward.  The 2x2 outputs of low-resolution pixel (i, j) see, through F(2x2, 3x3), the 4x4 patch of the upsampled image with
// origin (2i-1, 2j-1): its rows are x[i-1], x[i], x[i], x[i+1] -- rows (and columns) repeat in pairs, so row 2 of B^T d
// vanishes and only the 3 x 3 transform-domain positions {0, 1, 3}^2 carry anything.  With the factor 2 of position 1 folded
// into the weights nothing is halved at all:
//     Y = A'^T [ sum_ci (G' g G'^T) (.) (L d3 L^T) ] A'     d3: 3x3 low-resolution patch (origin (i-1, j-1)), g: 3x3 kernel
//     L = [1 -1 0; 0 1 0; 0 1 -1]     G' = [1 0 0; 1 1 1; 0 0 1]     A'^T = [1 1 0; 0 1 -1]
// (row 0 of the result: g0 (x[i-1] - x[i]) + (g0+g1+g2) x[i] = g0 x[i-1] + (g1+g2) x[i]: the phase kernel of an even output row).
// Dgrad (mode VAE_MODE_UP2X_DGRAD): the gradient wrt the low-resolution x is the 2x2 sum-pool of the 3x3 dgrad of the
// high-resolution dY; the pool folds into the output transform (sum of the rows of A^T = [1 2 0 -1]) and again only positions
// {0, 1, 3}^2 are needed:
//     dx = s^T [ sum_co (G' g~ G'^T) (.) (Bt3 d4 Bt3^T) ] s    d4: 4x4 patch of dY (origin (2i-1, 2j-1)), g~: rotated kernel
//     Bt3 = [1 0 -1 0; 0 1 1 0; 0 1 0 -1]     s = [1 1 -1]
// Neither the upsampled tensor nor the high-resolution input gradient ever exists.
//
// Kernel: as conv3_wino.hip (transformed weights U = G' g G'^T rebuilt from the live weights per launch, layout [K/8][9][N][8];
// chunk of 8 channels per step; the chunk's halo staged once, the V image [9][32 tiles][8] double buffered in LDS; U fragments
// from L2 straight into registers one step ahead; accumulators through LDS in the epilogue).  Workgroup = 8 waves = 32
// low-resolution pixels (4 x 8) x 64 channels: wave w owns position w (both 32-channel blocks) and the NINTH position is split
// by channel block over two waves on different SIMDs (waves 0, 1 in even workgroups, 2, 3 in odd ones: wave k of a workgroup runs on
// SIMD k % 4), so a workgroup loads the SIMDs 20 20 16 16 MFMAs per step and two co-resident workgroups 36 each.  (A first
// version with one position per wave and 9 waves put three waves of every workgroup on SIMD 0: 0.46 of the matrix peak.)
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr int UTH = 4, UTW = 8;            // low-resolution pixels (= Winograd tiles) per workgroup tile
constexpr int UTL = UTH * UTW;             // 32
constexpr int UBK = 8;                     // channels per step
constexpr int NPOS = 9;
constexpr int UNT = 512;                   // 8 waves
constexpr int UNB = 2;                     // 32-channel blocks per workgroup
constexpr int USV = NPOS * UTL * UBK;      // floats per V buffer (9216 B)
constexpr int UHF = (UTH + 2) * (UTW + 2); // forward halo pixels (6 x 10)
constexpr int UHD = (2 * UTH + 2) * (2 * UTW + 2);  // dgrad halo pixels (10 x 18)
constexpr int USH = UHD * UBK;             // floats per halo buffer (sized for the dgrad)
constexpr int UMLD = 33;                   // epilogue image row (floats)
constexpr int USM = NPOS * UTL * UMLD;     // epilogue image, overlays the V / halo buffers
constexpr int UP_LDS = (USM > 2 * USV + 2 * USH ? USM : 2 * USV + 2 * USH) * 4;  // 38016 B

// U[pos = a*3+b][n][k] = (G' g G'^T)[a][b] for g = W[n][.][.][k] (forward: N = Cout, K = Cin) or g = rot180(W[k][.][.][n]) (dgrad:
// N = Cin, K = Cout); G' = [1 0 0; 1 1 1; 0 0 1]; layout [K/8][9][N][8]
__global__ __launch_bounds__(256) void upwino_weights_kernel(const float* __restrict__ W, int N, int K, int dgrad, int64_t sn, int64_t sk, int64_t st,
                                                             float* __restrict__ U) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * K) return;
  const int k = (int)(i % K), n = (int)(i / K);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int tap = dgrad ? (2 - a) * 3 + (2 - b) : a * 3 + b;
      g[a][b] = W[(int64_t)n * sn + (int64_t)k * sk + (int64_t)tap * st];
    }
  float t[3][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = (g[0][b] + g[2][b]) + g[1][b];
    t[2][b] = g[2][b];
  }
  float* o = U + ((int64_t)(k >> 3) * NPOS * N + n) * 8 + (k & 7);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[(int64_t)(a * 3 + 0) * N * 8] = t[a][0];
    o[(int64_t)(a * 3 + 1) * N * 8] = (t[a][0] + t[a][2]) + t[a][1];
    o[(int64_t)(a * 3 + 2) * N * 8] = t[a][2];
  }
}

template <bool DG>
__global__ __launch_bounds__(UNT, 4) void conv3_upwino_kernel(vae_igemm_args p, const float* __restrict__ U, int tiles_x, int tiles_y, int xcd_sp) {
  constexpr int WBN = 32 * UNB;           // output channels per workgroup
  constexpr int HW_ = DG ? 2 * UTW + 2 : UTW + 2;  // halo width in pixels
  constexpr int HP = DG ? UHD : UHF;      // halo pixels
  __shared__ __attribute__((aligned(16))) float wsm[UP_LDS / 4];
  float* const sV = wsm;            // [2][USV]
  float* const sH = wsm + 2 * USV;  // [2][USH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // wave = position 0..7 of the 3 x 3 transform domain
  // position 8: channel block 0 on wave xw0, block 1 on wave xw0 + 1
  const int xw0 = (blockIdx.x & 1) * 2;
  const bool extra = wave == xw0 || wave == xw0 + 1;  // uniform per wave
  const int xnb = wave - xw0;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = (p.N + WBN - 1) / WBN;
  int t = blockIdx.x, tn;
  if (xcd_sp) {  // the channel blocks of a spatial tile on one XCD (see conv3_wino.hip)
    tn = (t >> 3) % tilesN;
    t = ((t >> 3) / tilesN) * 8 + (t & 7);
  } else {
    tn = t % tilesN;
    t /= tilesN;
  }
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * UTH, x0 = tx * UTW, n0 = tn * WBN;  // low-resolution origin of the tile
  const int nsteps = p.K / UBK;

  // ---- halo role: pixel hp of the chunk's halo, channel quad hq: loaded once per chunk, stored to sH two steps ahead ----
  const bool hrole = tid < 2 * HP;
  const int hp = tid >> 1, hq = tid & 1;
  const int hy = (DG ? 2 * y0 : y0) - 1 + hp / HW_, hx = (DG ? 2 * x0 : x0) - 1 + hp % HW_;
  const bool hin = hrole && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws);
  const auto rsA = VAE_BUF_RSRC(p.A + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 4u);
  const unsigned hbase = hin ? (unsigned)(((hy * g.Ws + hx) * g.Cs + hq * 4) * 4) : BUF_OOB;
  f32x4 rh = {0.f, 0.f, 0.f, 0.f};
  auto load_halo_into = [&](int step, f32x4& h) {
    const bool ok = hin && step < nsteps;
    h = VAE_BUF_LOAD4(rsA, ok ? hbase + (unsigned)(step * UBK * 4) : BUF_OOB);
  };
  auto store_halo_from = [&](float* dst, const f32x4& h) {
    if (hrole) *reinterpret_cast<f32x4*>(&dst[hp * UBK + hq * 4]) = h;
  };

  // ---- V role (threads 0..255): tile vt, channel vc of the chunk ----
  const bool vrole = tid < 256;
  const int vt = (tid >> 3) & 31, vc = tid & 7;
  const int vorg = DG ? ((2 * (vt >> 3)) * HW_ + 2 * (vt & 7)) * UBK + vc : ((vt >> 3) * HW_ + (vt & 7)) * UBK + vc;
  auto write_v = [&](const float* sHc, float* dst) {
    if (!vrole) return;
    float r[3][DG ? 4 : 3];
    if (!DG) {  // L d3 L^T
      float d[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) d[i][j] = sHc[vorg + (i * HW_ + j) * UBK];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        r[0][j] = d[0][j] - d[1][j];
        r[1][j] = d[1][j];
        r[2][j] = d[1][j] - d[2][j];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dst[((a * 3 + 0) * UTL + vt) * UBK + vc] = r[a][0] - r[a][1];
        dst[((a * 3 + 1) * UTL + vt) * UBK + vc] = r[a][1];
        dst[((a * 3 + 2) * UTL + vt) * UBK + vc] = r[a][1] - r[a][2];
      }
    } else {    // Bt3 d4 Bt3^T
      float d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = sHc[vorg + (i * HW_ + j) * UBK];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dst[((a * 3 + 0) * UTL + vt) * UBK + vc] = r[a][0] - r[a][2];
        dst[((a * 3 + 1) * UTL + vt) * UBK + vc] = r[a][1] + r[a][2];
        dst[((a * 3 + 2) * UTL + vt) * UBK + vc] = r[a][1] - r[a][3];
      }
    }
  };

  // ---- U fragments of this wave's position: [step][pos][n0 + 32 nb + lr][4 lh .. 4 lh + 3], L2 -> registers one step ahead ----
  const auto rsU = VAE_BUF_RSRC(U, (size_t)nsteps * NPOS * p.N * 8 * 4u);
  unsigned bvo[UNB];
#pragma unroll
  for (int q = 0; q < UNB; ++q) bvo[q] = (n0 + q * 32 + lr < p.N) ? (unsigned)(((n0 + q * 32 + lr) * 8 + lh * 4) * 4) : BUF_OOB;
  const unsigned bpos = (unsigned)p.N * 32u;  // bytes per position of the U image
  const unsigned bvx = extra ? bvo[xnb & 1] : BUF_OOB;  // the ninth position's block (waves without one request nothing)
  // ONE register set, refilled behind the MFMAs that consume it (conv3_wino.hip); every step issues the same UNB + 1 requests
  // in code every wave runs (beyond the last chunk the last one again, never used; waves without a ninth-position block request
  // out of range)
  f32x4 bq[UNB + 1];
  auto load_b1 = [&](int step, int i) {
    const int st = min(step, nsteps - 1);
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(st * NPOS + (i < UNB ? wave : 8)) * bpos);
    bq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, i < UNB ? bvo[i < UNB ? i : 0] : bvx, so, 0));
  };

  f32x16 acc[UNB], accx;
#pragma unroll
  for (int nb = 0; nb < UNB; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) accx[e] = 0.f;

  // prologue: halo(0), halo(1) in LDS, V(0) from halo(0); halo(2) and the U fragments of step 0 in registers
#pragma unroll
  for (int i = 0; i <= UNB; ++i) load_b1(0, i);
  {
    f32x4 h0, h1;
    load_halo_into(0, h0);
    load_halo_into(1, h1);
    load_halo_into(2, rh);
    store_halo_from(sH, h0);
    store_halo_from(sH + USH, h1);
  }
  __syncthreads();
  write_v(sH, sV);
  __syncthreads();
  auto multiply = [&](const float* cV, int s) {  // step s; requests step s+1's fragments as it goes
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&cV[(wave * UTL + lr) * UBK + 4 * lh]);
    f32x4 ax = a4;
    if (extra) ax = *reinterpret_cast<const f32x4*>(&cV[(8 * UTL + lr) * UBK + 4 * lh]);
#pragma unroll
    for (int nb = 0; nb < UNB; ++nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bq[nb][e], acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_b1(s + 1, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (extra) {
#pragma unroll
      for (int e = 0; e < 4; ++e) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bq[UNB][e], accx, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_b1(s + 1, UNB);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stage_next = [&](int s, int par) {
    if (s + 1 < nsteps) write_v(sH + (par ^ 1) * USH, sV + (par ^ 1) * USV);  // V(s+1): nobody reads that buffer now
    store_halo_from(sH + par * USH, rh);                                       // halo(s+2)
    load_halo_into(s + 3, rh);
  };
  // waves 0..3 (which own the V transform) stage first, waves 4..7 multiply first: two copies of the loop, so that the order of
  // the memory requests is fixed inside each and hipcc's wait counts are exact (conv3_wino.hip)
  auto step = [&](int s, int par, auto first_c) {
    const float* cV = sV + par * USV;
    if (decltype(first_c)::value) {
      stage_next(s, par);
      __builtin_amdgcn_sched_barrier(0);
      multiply(cV, s);
    } else {
      multiply(cV, s);
      __builtin_amdgcn_sched_barrier(0);
      stage_next(s, par);
    }
    __syncthreads();
  };
  auto run = [&](auto first_c) {
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
      step(s, 0, first_c);
      step(s + 1, 1, first_c);
    }
    if (s < nsteps) step(s, 0, first_c);
  };
  if (wave < 4) run(std::true_type{});  // uniform per wave
  else run(std::false_type{});

  // ---- epilogue: per 32-channel block, M through LDS, then the output transform ----
  float* const sM = wsm;  // [9][32 tiles][UMLD], over the V / halo buffers (the last step's barrier has passed)
  const size_t obytes = (size_t)g.Ho * g.Wo * p.ldc * 4u;
  const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)b * g.Ho * g.Wo * p.ldc, obytes);
#pragma unroll
  for (int nb = 0; nb < UNB; ++nb) {
    if (nb > 0) __syncthreads();  // the previous block's reads are done
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int tile = (e & 3) + 8 * (e >> 2) + 4 * lh;
      sM[(wave * UTL + tile) * UMLD + lr] = acc[nb][e];
    }
    if (wave == xw0 + nb) {  // the ninth position's block nb
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tile = (e & 3) + 8 * (e >> 2) + 4 * lh;
        sM[(8 * UTL + tile) * UMLD + lr] = accx[e];
      }
    }
    __syncthreads();
    {
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int co = tid & 31, tile = (tid >> 5) + 16 * rnd;
        const int col = n0 + nb * 32 + co;
        float m[9];
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) m[pos] = sM[(pos * UTL + tile) * UMLD + co];
        const int oy = y0 + (tile >> 3), ox = x0 + (tile & 7);  // low-resolution pixel of the tile
        if (!DG) {
          const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
          float h[2][3];  // A'^T M
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            h[0][j] = m[0 * 3 + j] + m[1 * 3 + j];
            h[1][j] = m[1 * 3 + j] - m[2 * 3 + j];
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const float yv[2] = {h[a][0] + h[a][1], h[a][1] - h[a][2]};
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              const int py = 2 * oy + a, px = 2 * ox + bb;
              const unsigned off = (col < p.N && py < g.Ho && px < g.Wo) ? (unsigned)(((py * g.Wo + px) * p.ldc + col) * 4) : BUF_OOB;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv[bb] + bv), rsC, off, 0, 0);
            }
          }
        } else {
          // s^T M s with s = [1 1 -1]
          const float h0 = (m[0] + m[3]) - m[6], h1 = (m[1] + m[4]) - m[7], h2 = (m[2] + m[5]) - m[8];
          const float v = (h0 + h1) - h2;
          const unsigned off = (col < p.N && oy < g.Ho && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + col) * 4) : BUF_OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off, 0, 0);
        }
      }
    }
  }
}

}  // namespace

// forward (mode UP2X: source = the low-resolution x, row grid = the high-resolution output) or dgrad (mode UP2X_DGRAD: source =
// the high-resolution dY, row grid = the low-resolution input gradient) of conv3x3(nearest_upsample_2x(x)) in fp32
bool conv3_upwino_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_F32 || a.A16 != nullptr || a.batch != 1 || a.alpha != 1.0f || a.xf != VAE_XF_NONE) return false;
  if (g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.tapmask != 0 || a.a_step > 1 || a.c_step > 1 || a.track != nullptr || a.res != nullptr || a.gstat != nullptr) return false;
  if (a.out_bf16 || a.a_bf16 || a.res_bf16) return false;
  int Hl, Wl;  // low-resolution size
  if (g.mode == VAE_MODE_UP2X) {
    if (g.Ho != 2 * g.Hs || g.Wo != 2 * g.Ws || a.sk != 1) return false;
    Hl = g.Hs; Wl = g.Ws;
  } else if (g.mode == VAE_MODE_UP2X_DGRAD) {
    if (g.Hs != 2 * g.Ho || g.Ws != 2 * g.Wo || a.sn != 1 || a.bias != nullptr) return false;
    Hl = g.Ho; Wl = g.Wo;
  } else {
    return false;
  }
  if (Hl % UTH != 0 || Wl % UTW != 0 || a.K % UBK != 0 || a.K < 64 || a.K > 1024 || a.N < 32 || a.N % 4 != 0 || g.Cs < a.K) return false;
  if (!aligned16(a.A) || !aligned16(a.C)) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return false;
  if ((size_t)a.K * NPOS * a.N * 4u >= BUF_MAX) return false;
  return true;
}

int launch_upwino_weights(const vae_igemm_args& a, float* U, hipStream_t st) {
  const int64_t n = (int64_t)a.N * a.K;
  hipLaunchKernelGGL(upwino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.W, a.N, a.K,
                     a.g.mode == VAE_MODE_UP2X_DGRAD ? 1 : 0, a.sn, a.sk, a.st, U);
  return 0;
}

int launch_conv3_upwino(const vae_igemm_args& a, const float* U, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const bool dg = g.mode == VAE_MODE_UP2X_DGRAD;
  const int Hl = dg ? g.Ho : g.Hs, Wl = dg ? g.Wo : g.Ws;
  const int tx = Wl / UTW, ty = Hl / UTH;
  const int tilesN = (a.N + 32 * UNB - 1) / (32 * UNB);
  const int64_t nt = (int64_t)tilesN * tx * ty * g.B;
  if (nt > 0x7fffffffLL) return VAE_EINVAL;
  constexpr size_t xcd_u = (size_t)4 << 20;
  const int xcd_sp = (tilesN > 1 && (size_t)a.K * NPOS * a.N * 4u <= xcd_u && ((int64_t)tx * ty * g.B) % 8 == 0) ? 1 : 0;
  if (dg) hipLaunchKernelGGL((conv3_upwino_kernel<true>), dim3((unsigned)nt), dim3(UNT), 0, st, a, U, tx, ty, xcd_sp);
  else hipLaunchKernelGGL((conv3_upwino_kernel<false>), dim3((unsigned)nt), dim3(UNT), 0, st, a, U, tx, ty, xcd_sp);
  return 0;
}
