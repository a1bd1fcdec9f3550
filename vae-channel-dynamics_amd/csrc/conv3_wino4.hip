// Winograd F(4x4, 3x3) for the fp32 forward and dgrad of the 3x3 stride-1 layers: 36 multiplications per 4x4 output tile and
// channel pair instead of 144 (4x fewer MFMA passes than the direct convolution, 1.78x fewer than F(2x2,3x3) in conv3_wino.hip),
// exact fp32 products on v_mfma_f32_32x32x2_f32.
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          d: 6x6 input patch, g: 3x3 kernel, Y: 4x4 outputs
// with Lavin's matrices for the points (0, +-1, +-2, inf).  Accuracy: tools/wino_f4_error_study.py (CPU emulation in fp32 on
// the operands of the oracle's R = 64 step, against float64): a layer's output / input gradient 4e-6..1.3e-5 of its max (F(2x2):
// 5e-7), the whole step's losses unchanged (2e-7), worst gradient tensor 3.3e-5 of its own max (F(2x2): 1.9e-5, direct fp32:
// 1.6e-5), tracked statistics 9e-7 -- inside the bars of tests/test_engine_gpu.py (profiles/r04_wino_f4_error_study.json).
//
// Per position (xi, nu) of the 6x6 transform domain the product summed over input channels is a GEMM
//     M[pos][tile][co] = sum_ci V[pos][tile][ci] * U[pos][co][ci]
//   * Workgroup = 12 waves (3 per SIMD, <= 168 registers) = 16 x 32 output pixels (32 tiles of 4x4) x 64 output channels, channel
//     chunks of 8 per step; ONE workgroup per CU (the 36 x 32 x 64 accumulators are 295 KB of the CU's 512 KB register file).
//     Wave w owns positions 3w .. 3w+2 for both 32-channel blocks: 24 MFMAs per wave and step, 96 accumulator registers.
//   * U = G g G^T built per launch from the LIVE weights (vae_wino_weights), [K/8][36][N][8]: the B fragment of (position,
//     channel block) is 16 contiguous bytes per lane, requested by the one wave that owns the position straight from L2 into
//     registers (one register set, re-requested right behind the MFMAs that consumed it), never through LDS.
//   * The chunk's 18 x 34 input halo comes by LDS-DMA (buffer_load_dwordx4 ... lds, two wave-instructions per wave and step: no
//     registers, no stores, nobody waits for it; retired by a counted vmcnt in front of the step's barrier) into a pixel-major image
//     with 32 bytes of padding behind every 4 pixels: dword (pixel, channel) = 8 (pixel + pixel / 4) + channel, so the 4 tiles x 8
//     channels a 32-lane group of the transform reads are 32 different banks and every patch read is base + immediate.  (The
//     instantiations with GroupNorm(+SiLU) fused into the staging go through registers, two steps ahead, into the same image.)
//   * All 768 threads transform: thread = (tile, channel, row pair): rows {1,2}, {3,4} or {0,5} of B^T d (they share their
//     sub-expressions pairwise), then those two rows times B: 48 operations, 12 values written to the V image [36][2 k-halves]
//     [32 tiles][4] (a wave's A-fragment read of one position is 1 KB, conflict-free for ds_read_b128).  The transform is cut into
//     six slices INTERLEAVED with the wave's own groups of 4 MFMAs: measured alone the staging of a step was a chain of LDS round
//     trips as long as its MFMA work (4400 against 4608 cycles; one after the other 6700), so each slice's latency has to pass
//     behind the following group.  One barrier per step; one copy of the step loop per row pair (uniform per wave).
//   * What a step still costs (tools/wino4_timing.py, profiles/r04_pmc_wino4.txt): 6300 cycles.  An fp32 MFMA runs at the fp32
//     vector rate and vector instructions cost matrix-pipe time one for one: 64 x 72 MFMAs + 4 x 3 x 63 vector instructions per
//     SIMD = 5364, barrier and first-fragment latency the rest.  Prologue + epilogue (one workgroup per CU: nothing beside them)
//     are 43k cycles per tile: 30 % of a 128-channel layer's workgroup, 9 % of a 512-channel one.
//   * Epilogue per 32-channel block: accumulators through LDS ([36][32 tiles][32 channels]), each thread takes A^T M A of its (tile,
//     channel) pairs, adds bias / residual, writes the 4x4 outputs (lanes along channels: 128-byte rows) and leaves the
//     GroupNorm moments of the outputs / -- dgrad launches -- the first pass of the GroupNorm backward, as conv3_wino.hip does.
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr int FTH = 16, FTW = 32;        // output pixels per workgroup tile
constexpr int FNTL = 32;                 // tiles per workgroup (4 rows x 8 columns of 4x4 outputs)
constexpr int FBK = 8;                   // channels per step
constexpr int FNT = 768;                 // 12 waves
constexpr int FPOS = 36;
constexpr int FBN = 64;                  // output channels per workgroup
constexpr int FSV = FPOS * FNTL * FBK;   // floats per V buffer (36,864 B)
constexpr int FHW = 34, FHH = 18;        // halo of the tile
constexpr int FHP = FHW * FHH;           // 612 pixels
constexpr int FSH = 1536 * 4;            // floats per halo buffer: 1536 slots of 16 B (24,576 B), see the kernel
constexpr int FSS = 2 * 1024;            // GroupNorm scale / shift rows of the image (K <= 1024)
constexpr int FMLD = 32;                 // epilogue image row (floats): writes are lane-contiguous per tile, reads per (position, tile): no padding needed
constexpr int FSM = FPOS * FNTL * FMLD + 12 * 8 * 2 + 24 * 32 * 2;  // epilogue image + statistics scratch + GroupNorm-backward sums
constexpr int FMAIN = 2 * FSV + 2 * FSH + FSS;
constexpr int F_LDS = (FSM > FMAIN ? FSM : FMAIN) * 4;  // 154,368 B
static_assert(F_LDS <= 160 * 1024, "LDS");

// U[pos][n][k] = (G g G^T)[pos] for g = W[n][.][.][k] (forward) or g = rot180(W[k][.][.][n]) (dgrad); layout [K/8][36][N][8]
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ W, int N, int K, int dgrad, int64_t sn, int64_t sk, int64_t st,
                                                            float* __restrict__ U) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * K) return;
  // thread -> (k / 8, n, k % 8) with k % 8 fastest: for every position the workgroup's 256 threads write 1 KB of U contiguously
  // (the image is 36/9 = 4x the bytes of the weights it is made from: the writes decide; (n, k) with k fastest wrote 32-byte pieces)
  const int k = (int)((i / (8 * (int64_t)N)) * 8 + (i & 7)), n = (int)((i >> 3) % N);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int tap = dgrad ? (2 - a) * 3 + (2 - b) : a * 3 + b;
      g[a][b] = W[(int64_t)n * sn + (int64_t)k * sk + (int64_t)tap * st];
    }
  constexpr float c4 = 0.25f, c6 = 1.0f / 6.0f, c12 = 1.0f / 12.0f, c24 = 1.0f / 24.0f;
  float t[6][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {  // G g
    const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
    t[0][b] = c4 * g0;
    t[1][b] = -c6 * ((g0 + g2) + g1);
    t[2][b] = -c6 * ((g0 + g2) - g1);
    t[3][b] = (c24 * g0 + c6 * g2) + c12 * g1;
    t[4][b] = (c24 * g0 + c6 * g2) - c12 * g1;
    t[5][b] = g2;
  }
  float* o = U + ((int64_t)(k >> 3) * FPOS * N + n) * 8 + (k & 7);
#pragma unroll
  for (int a = 0; a < 6; ++a) {  // (G g) G^T
    const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
    float u[6];
    u[0] = c4 * g0;
    u[1] = -c6 * ((g0 + g2) + g1);
    u[2] = -c6 * ((g0 + g2) - g1);
    u[3] = (c24 * g0 + c6 * g2) + c12 * g1;
    u[4] = (c24 * g0 + c6 * g2) - c12 * g1;
    u[5] = g2;
#pragma unroll
    for (int b = 0; b < 6; ++b) o[(int64_t)(a * 6 + b) * N * 8] = u[b];
  }
}

// one 1-D input transform B^T r of a 6-vector (also the row pass: (B^T d) B = (B^T (B^T d)^T)^T)
__device__ __forceinline__ void bt6(const float r[6], float v[6]) {
  v[0] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
  const float a = fmaf(-4.f, r[2], r[4]), b = fmaf(-4.f, r[1], r[3]);
  v[1] = a + b;
  v[2] = a - b;
  const float c = r[4] - r[2], e = r[3] - r[1];
  v[3] = fmaf(2.f, e, c);
  v[4] = fmaf(-2.f, e, c);
  v[5] = fmaf(4.f, r[1], fmaf(-5.f, r[3], r[5]));
}
// one 1-D output transform A^T m of a 6-vector -> 4
__device__ __forceinline__ void at6(const float m[6], float y[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y[0] = (m[0] + s12) + s34;
  y[1] = fmaf(2.f, d34, d12);
  y[2] = fmaf(4.f, s34, s12);
  y[3] = fmaf(8.f, d34, d12) + m[5];
}

template <int XF>
__global__ __launch_bounds__(FNT, 3) void conv3_wino4_kernel(vae_igemm_args p, const float* __restrict__ U, int tiles_x, int tiles_y, int xcd_sp) {
  __shared__ __attribute__((aligned(16))) float wsm[F_LDS / 4];
  float* const sH = wsm;            // [2][FSH]: the chunk's input halo (first: the LDS-DMA base stays below 64 KB)
  float* const sV = wsm + 2 * FSH;  // [2][FSV]
#ifdef VAE_WINO4_TIMING  // debug build (tools/wino4_timing.py): shader-clock stamps of waves 0, 4, 8 (the three waves of SIMD 0) of workgroups 0..7 -> p.track
  unsigned long long tE[10];  // entry, loop entry, loop exit, block 0: accumulators in LDS / outputs stored, block 1: the same, end; [8], [9]: 100 MHz clock
  unsigned long long tS[8][4];  // steps 2..9: begin, after the first phase, after the second, after the barrier
  const bool tw = (threadIdx.x & 255) == 0;
#define WEDGE(k) do { if (tw) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tE[k] = __builtin_amdgcn_s_memtime(); } } while (0)
#define WSTAMP(s, k) do { if (tw && (s) >= 2 && (s) < 10) { tS[(s) - 2][k] = __builtin_amdgcn_s_memtime(); } } while (0)
  if (tw) tE[8] = __builtin_amdgcn_s_memrealtime();
#else
#define WEDGE(k) do { } while (0)
#define WSTAMP(s, k) do { } while (0)
#endif
  WEDGE(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = p.N / FBN;
  // workgroup id -> (spatial tile, channel block): the rule of conv3_wino.hip (small U images: the channel blocks of a spatial
  // tile get ids congruent mod 8, one XCD's L2 fetches that tile's halo once)
  int t = blockIdx.x, tn;
  if (xcd_sp) {
    tn = (t >> 3) % tilesN;
    t = ((t >> 3) / tilesN) * 8 + (t & 7);
  } else {
    tn = t % tilesN;
    t /= tilesN;
  }
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * FTH, x0 = tx * FTW, n0 = tn * FBN;
  const int nsteps = p.K / FBK;

  // ---- the chunk's 18 x 34 input halo in LDS: pixel-major, 8 channels (32 B) per pixel, 32 B of padding behind every 4 pixels:
  //          dword address of (pixel hp, channel c) = 8 (hp + (hp >> 2)) + c
  // (4 tiles x 8 channels -- what a 32-lane group of the transform reads -- are then 32 different banks; without the padding all
  // tiles would sit on the same 8).  In 16-byte slots: slot m = 2 (hp + (hp >> 2)) + quad; 1530 slots, rounded to 1536 = 24
  // wave-instructions of LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes x 16 B land lane-linear at M0), TWO PER WAVE and step: the
  // halo never passes through registers, no thread waits for it and nothing is stored by hand.  Lanes whose slot is padding or
  // lies outside the image request an out-of-range offset: the DMA writes zeros (= the convolution's zero padding).
  // XF != NONE (GroupNorm(+SiLU) fused into the staging) cannot use the DMA: those instantiations stage through registers (two
  // steps ahead, as conv3_wino.hip) into the same image.
  const auto rsA = VAE_BUF_RSRC(p.A + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 4u);
  unsigned hoff[2];   // XF == NONE: source offsets of this lane's two slots; else: of the thread's two (pixel, quad) items
  int hdst[2];        // XF != NONE: LDS dword of the item
  bool hin[2], hown[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int hp, hq;
    if (XF == VAE_XF_NONE) {
      const int m = (wave * 2 + j) * 64 + lane, grp = m / 10, r = m - grp * 10;
      hp = 4 * grp + (r >> 1);
      hq = r & 1;
      hown[j] = r < 8 && hp < FHP;
    } else {
      const int it = tid + j * FNT;
      hp = it >> 1;
      hq = it & 1;
      hown[j] = it < 2 * FHP;
    }
    const int hy = y0 - 1 + hp / FHW, hx = x0 - 1 + hp % FHW;
    hin[j] = hown[j] && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws);
    hoff[j] = hin[j] ? (unsigned)(((hy * g.Ws + hx) * g.Cs + hq * 4) * 4) : BUF_OOB;
    hdst[j] = 8 * (hp + (hp >> 2)) + 4 * hq;
  }
  float* const sS = wsm + 2 * FSH + 2 * FSV;  // GroupNorm scale / shift rows of image b ([2][K])
  if (XF != VAE_XF_NONE) {
    for (int i = tid; i < p.K; i += FNT) {
      sS[i] = p.scale[(int64_t)b * g.Cs + i];
      sS[p.K + i] = p.shift[(int64_t)b * g.Cs + i];
    }
    __syncthreads();
  }
  // LDS-DMA of halo(step) into halo buffer `par` (XF == NONE).  Inline asm: hipcc must not see these loads -- with a visible
  // LDS-DMA in flight it waits vmcnt(0) at the next use of ANY load, which would serialise the U-fragment pipeline below.  They
  // are the OLDEST vector-memory operations of a step (6 fragment re-requests follow), so `s_waitcnt vmcnt(6)` in front of the
  // step's barrier retires them; beyond the last chunk the last one is requested again (never read) so the count stays fixed.
  const unsigned lds_h = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)sH) + (unsigned)wave * 2048u;
  auto dma_halo = [&](int step, int par) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(min(step, nsteps - 1) * FBK * 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_h + (unsigned)par * (FSH * 4u) + (unsigned)j * 1024u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst), "v"(hoff[j]), "s"(rsA), "s"(so) : "memory");
    }
  };
  struct Halo {
    f32x4 v[2];
  };
  Halo rh;
  rh.v[0] = rh.v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_halo_into = [&](int step, Halo& h) {  // XF != NONE
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = hin[j] && step < nsteps;
      h.v[j] = VAE_BUF_LOAD4(rsA, ok ? hoff[j] + (unsigned)(step * FBK * 4) : BUF_OOB);
    }
  };
  auto store_halo_from = [&](float* dst, const Halo& h, int step) {  // XF != NONE
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!hown[j]) continue;
      f32x4 v = h.v[j];
      const int c = min(step, nsteps - 1) * FBK + ((tid + j * FNT) & 1) * 4;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(&sS[c]), sh = *reinterpret_cast<const f32x4*>(&sS[p.K + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * sc[e] + sh[e];
        if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
        v[e] = (hin[j] && step < nsteps) ? u : 0.f;  // padding stays zero AFTER the transform
      }
      *reinterpret_cast<f32x4*>(&dst[hdst[j]]) = v;
    }
  };

  // ---- V role (all threads): channel vc of the chunk (fastest: a 32-lane group = 8 channels x 4 tile columns), tile (vty, vtx),
  // row pair RG = tid >> 8 (uniform per wave: each wave group runs its own copy of the main loop).  The thread takes rows
  // {1,2} (RG 0), {3,4} (RG 1) or {0,5} (RG 2) of B^T d -- they share their sub-expressions pairwise -- two patch columns at a
  // time, then those two rows times B: 48 operations, 12 values written to the V image [36][2 k-halves][32 tiles][4].
  const int vc = tid & 7, vtx = (tid >> 3) & 7, vty = (tid >> 6) & 3;
  const int vsrc = 10 * ((4 * vty) * FHW + 4 * vtx) + vc;                   // dword of the patch origin (a multiple of 4 pixels)
  const int vdst = ((vc >> 2) * FNTL + vty * 8 + vtx) * 4 + (vc & 3);       // + pos * (2 * FNTL * 4)
#define W4_HOFF(i, j) (8 * (FHW * (i) + (j)) + 8 * ((FHW * (i) + (j)) >> 2))  /* (origin + x) >> 2 = origin / 4 + (x >> 2) */
  // the patch values of columns 2k, 2k+1 the row pair needs: d[jj][i]
  auto v_read = [&](const float* sHc, auto rg_c, auto k_c, float (&d)[2][6]) {
    constexpr int RG = decltype(rg_c)::value, K2 = decltype(k_c)::value;
    const float* src = sHc + vsrc;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      constexpr int IA = RG == 2 ? 0 : 1, IB = RG == 2 ? 6 : 5;
#pragma unroll
      for (int i = IA; i < IB; ++i) d[jj][i] = src[W4_HOFF(i, 2 * K2 + jj)];
    }
  };
  auto v_cols = [&](const float (&d)[2][6], auto rg_c, auto k_c, float (&tr)[2][6]) {
    constexpr int RG = decltype(rg_c)::value, K2 = decltype(k_c)::value;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * K2 + jj;
      if (RG == 0) {  // rows 1, 2 of B^T d
        const float a = fmaf(-4.f, d[jj][2], d[jj][4]), bb = fmaf(-4.f, d[jj][1], d[jj][3]);
        tr[0][j] = a + bb;
        tr[1][j] = a - bb;
      } else if (RG == 1) {  // rows 3, 4
        const float c = d[jj][4] - d[jj][2], e = d[jj][3] - d[jj][1];
        tr[0][j] = fmaf(2.f, e, c);
        tr[1][j] = fmaf(-2.f, e, c);
      } else {  // rows 0, 5
        tr[0][j] = fmaf(4.f, d[jj][0], fmaf(-5.f, d[jj][2], d[jj][4]));
        tr[1][j] = fmaf(4.f, d[jj][1], fmaf(-5.f, d[jj][3], d[jj][5]));
      }
    }
  };
  auto v_row = [&](float* dst, auto rg_c, auto q_c, const float (&tr)[2][6]) {  // row q of the pair: times B, into the V image
    constexpr int RG = decltype(rg_c)::value, Q = decltype(q_c)::value;
    constexpr int I = Q == 0 ? (RG == 0 ? 1 : (RG == 1 ? 3 : 0)) : (RG == 0 ? 2 : (RG == 1 ? 4 : 5));
    float v[6];
    bt6(tr[Q], v);
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[(I * 6 + j) * (2 * FNTL * 4) + vdst] = v[j];
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto write_v = [&](const float* sHc, float* dst, auto rg_c) {  // the whole transform in one go (prologue)
    float d[2][6], tr[2][6];
    v_read(sHc, rg_c, I0{}, d); v_cols(d, rg_c, I0{}, tr);
    v_read(sHc, rg_c, I1{}, d); v_cols(d, rg_c, I1{}, tr);
    v_read(sHc, rg_c, I2{}, d); v_cols(d, rg_c, I2{}, tr);
    v_row(dst, rg_c, I0{}, tr);
    v_row(dst, rg_c, I1{}, tr);
  };

  // ---- U fragments: B operand of (position, channel block nb) = U[step][pos][n0 + 32 nb + lr][4 lh .. 4 lh + 3] ----
  const auto rsU = VAE_BUF_RSRC(U, (size_t)nsteps * FPOS * p.N * 8 * 4u);
  const unsigned bvo = (unsigned)(((n0 + lr) * 8 + lh * 4) * 4);  // (+ 1024 B for the second channel block)
  const unsigned bpos = (unsigned)p.N * 32u;  // bytes per position of the U image
  f32x4 bq[6];
  auto load_b1 = [&](int step, int i) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(min(step, nsteps - 1) * FPOS + 3 * wave) * bpos);
    bq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, bvo + (i & 1) * 1024u, so + (i >> 1) * bpos, 0));
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int pi = 0; pi < 3; ++pi)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pi][nb][e] = 0.f;

  // prologue: halo(0), halo(1) in LDS, V(0) from halo(0); the U fragments of step 0 in registers (XF != NONE: halo(2) too)
  if (XF == VAE_XF_NONE) {
    dma_halo(0, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_b1(0, i);
    dma_halo(1, 1);  // lands under the first transform: waited for in front of the barrier that opens the loop
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // halo(0) has landed (6 fragment requests + 2 DMAs are younger)
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) load_b1(0, i);
    Halo h0, h1;
    load_halo_into(0, h0);
    load_halo_into(1, h1);
    load_halo_into(2, rh);
    store_halo_from(sH, h0, 0);
    store_halo_from(sH + FSH, h1, 1);
  }
  __syncthreads();
  if (wave < 4) write_v(sH, sV, I0{});
  else if (wave < 8) write_v(sH, sV, I1{});
  else write_v(sH, sV, I2{});
  if (XF == VAE_XF_NONE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // halo(1) (and the first fragments) have landed
  __syncthreads();

  // One copy of the main loop per wave group (uniform per wave).  A wave's staging (LDS reads -> 48 operations -> LDS writes)
  // measured alone is a chain of round trips as long as the step's MFMA work (tools/wino4_timing.py: 4400 cycles per step with
  // the MFMAs removed, 5070 with the staging removed, 6700 with one after the other), so it is INTERLEAVED with the wave's own
  // MFMAs: each group of 4 MFMAs (one position x one channel block, 256 cycles of the matrix pipe) is followed by the next
  // slice of the transform, whose LDS latency then passes behind the following group:
  //   DMA halo(s+2) | A(0), patch columns 0-1 requested | G(0,0) | columns 0-1 -> rows; columns 2-3 requested | G(0,1), A(1) |
  //   columns 2-3; 4-5 requested | G(1,0) | columns 4-5; first row x B -> V(s+1) | G(1,1), A(2) | second row -> V(s+1) |
  //   G(2,0) | G(2,1) | vmcnt(6): the DMA has landed | barrier
  auto run = [&](auto rg_c) {
    auto group = [&](const f32x4& a, int s, auto i_c) {  // 4 MFMAs of fragment i = 2 pi + nb, then its re-request for step s+1
      constexpr int I = decltype(i_c)::value;
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 1  // (diagnostic builds: 1 = no MFMAs, 2 = no staging)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[I >> 1][I & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[I][e], acc[I >> 1][I & 1], 0, 0, 0);
#else
      acc[I >> 1][I & 1][0] += a[0] * bq[I][0];
#endif
      __builtin_amdgcn_sched_barrier(0);
      load_b1(s + 1, I);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int s, int par) {
      const float* cV = sV + par * FSV;
      const float* cH = sH + (par ^ 1) * FSH;   // halo(s+1)
      float* nV = sV + (par ^ 1) * FSV;         // V(s+1): nobody reads that buffer now
      auto read_a = [&](int pi) { return *reinterpret_cast<const f32x4*>(&cV[(((3 * wave + pi) * 2 + lh) * FNTL + lr) * 4]); };
      f32x4 a0, a1, a2;
      float d[2][6], tr[2][6];
      WSTAMP(s, 0);
      if (XF == VAE_XF_NONE) dma_halo(s + 2, par);  // over halo(s), which V(s) was made from
      a0 = read_a(0);
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 2
      v_read(cH, rg_c, I0{}, d);
#endif
      __builtin_amdgcn_sched_barrier(0);
      group(a0, s, I0{});
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 2
      v_cols(d, rg_c, I0{}, tr);
      v_read(cH, rg_c, I1{}, d);
#endif
      __builtin_amdgcn_sched_barrier(0);
      group(a0, s, I1{});
      a1 = read_a(1);
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 2
      v_cols(d, rg_c, I1{}, tr);
      v_read(cH, rg_c, I2{}, d);
#endif
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(s, 1);
      group(a1, s, I2{});
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 2
      v_cols(d, rg_c, I2{}, tr);
      v_row(nV, rg_c, I0{}, tr);
#endif
      __builtin_amdgcn_sched_barrier(0);
      group(a1, s, std::integral_constant<int, 3>{});
      a2 = read_a(2);
#if !defined(VAE_W4_SKIP) || VAE_W4_SKIP != 2
      v_row(nV, rg_c, I1{}, tr);
#endif
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(s, 2);
      group(a2, s, std::integral_constant<int, 4>{});
      if (XF != VAE_XF_NONE) {
        store_halo_from(sH + par * FSH, rh, s + 2);  // halo(s+2) over halo(s)
        load_halo_into(s + 3, rh);
        __builtin_amdgcn_sched_barrier(0);
      }
      group(a2, s, std::integral_constant<int, 5>{});
      if (XF == VAE_XF_NONE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // this step's DMA (older than the 6 re-requests) has landed
      __syncthreads();
      WSTAMP(s, 3);
    };
    int s = 0;
#pragma unroll 1
    for (; s + 1 < nsteps; s += 2) {
      step(s, 0);
      step(s + 1, 1);
    }
    if (s < nsteps) step(s, 0);
  };
  WEDGE(1);
  if (wave < 4) run(I0{});
  else if (wave < 8) run(I1{});
  else run(I2{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a DMA of the last steps may still be writing the buffers the epilogue overlays)
  __syncthreads();

  WEDGE(2);
  // ---- epilogue: per 32-channel block, M through LDS, then Y = A^T M A ----
  float* const sM = wsm;  // [36][32 tiles][FMLD], over the V / halo buffers (the last step's barrier has passed)
  const size_t obytes = (size_t)g.Ho * g.Wo * p.ldc * 4u;
  const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)b * g.Ho * g.Wo * p.ldc, obytes);
  const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)b * g.Ho * g.Wo * p.ldc, obytes);
  const bool gnb = p.gnb_ws != nullptr;  // uniform
  const unsigned xes = p.gnb_x_bf16 ? 2u : 4u;
  const auto rsX = VAE_BUF_RSRC(reinterpret_cast<const char*>(gnb ? p.gnb_x : (const void*)p.C) + (int64_t)b * g.Ho * g.Wo * p.ldc * xes,
                                (size_t)g.Ho * g.Wo * p.ldc * xes);
  const int nrnd = tid < 1024 - FNT ? 2 : 1;  // (tile, channel) items 0..1023 of a block: thread tid takes tid and tid + 768
  float* const red = sM + FPOS * FNTL * FMLD;  // [12 waves][groups of the 32-channel block][2]
  float* const redb = red + 12 * 8 * 2;        // [24 slots][32 channels][2]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
      const int pos = 3 * wave + pi;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tile = (e & 3) + 8 * (e >> 2) + 4 * lh;
        sM[(pos * FNTL + tile) * FMLD + lr] = acc[pi][nb][e];
      }
    }
    __syncthreads();
    WEDGE(3 + 2 * nb);
    const int co = tid & 31, col = n0 + nb * 32 + co;
    const float bv = p.bias ? p.bias[col] : 0.f;
    float gpv = 0.f, gs1 = 0.f, gs2 = 0.f;  // GroupNorm statistics of this thread's outputs of channel `col`: shifted sums
    float bs1 = 0.f, bs2 = 0.f;             // GroupNorm backward: sum dz, sum dz * xhat over the same outputs
    float bmu = 0.f, brs = 0.f, bga = 0.f, bbe = 0.f;
    if (gnb) {
      const int grp = col / (p.N / p.gnb_groups);
      bmu = p.gnb_mean[b * p.gnb_groups + grp];
      brs = p.gnb_rstd[b * p.gnb_groups + grp];
      bga = p.gnb_gamma[col];
      bbe = p.gnb_beta[col];
    }
#pragma unroll 1
    for (int rnd = 0; rnd < nrnd; ++rnd) {
      const int tile = (tid >> 5) + 24 * rnd;
      const int oy = y0 + 4 * (tile >> 3), ox = x0 + 4 * (tile & 7);
      // per-lane offset of the tile's first output + a workgroup-uniform (row, column) part in a scalar register: no address
      // arithmetic on the vector pipe per access
      const unsigned obase = (unsigned)(((oy * g.Wo + ox) * p.ldc + col) * 4);
      const unsigned rstep = (unsigned)(g.Wo * p.ldc * 4), cstep = (unsigned)(p.ldc * 4);
      float rres[16];
      if (p.res) {  // uniform: requested before the LDS reads below
#pragma unroll
        for (int q = 0; q < 16; ++q)
          rres[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, obase, (q >> 2) * rstep + (q & 3) * cstep, 0));
      }
      float xin[16];
      if (gnb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned so = (q >> 2) * rstep + (q & 3) * cstep;
          xin[q] = p.gnb_x_bf16 ? __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsX, obase >> 1, so >> 1, 0) << 16)
                                : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, obase, so, 0));
        }
      }
      float h[4][6];  // A^T M: per column j of M the 4 output rows
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float m[6], y[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = sM[((i * 6 + j) * FNTL + tile) * FMLD + co];
        at6(m, y);
#pragma unroll
        for (int a = 0; a < 4; ++a) h[a][j] = y[a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float y[4];
        at6(h[a], y);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int q = a * 4 + bb;
          float v = y[bb] + bv;
          if (p.res) v += rres[q];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, obase, a * rstep + bb * cstep, 0);
          if (p.gstat) {  // uniform (forward launches whose output feeds a GroupNorm)
            if (rnd == 0 && q == 0) gpv = v;
            const float dv = v - gpv;
            gs1 += dv;
            gs2 = fmaf(dv, dv, gs2);
          }
          if (gnb) {  // uniform; same arithmetic per element as gn_bwd_partial_kernel (norm.hip)
            const float xh = (xin[q] - bmu) * brs;
            float du = v;
            if (p.gnb_silu) du *= silu_grad_f(xh * bga + bbe);
            bs1 += du;
            bs2 += du * xh;
          }
        }
      }
    }
    WEDGE(4 + 2 * nb);
    if (gnb) {
      redb[((tid >> 5) * 32 + co) * 2] = bs1;
      redb[((tid >> 5) * 32 + co) * 2 + 1] = bs2;
    }
    const int cpg = p.gstat ? p.N / p.gstat_groups : 4, ng = 32 / cpg;
    const float nlane = 16.f * (float)nrnd;  // outputs per lane (uniform per wave: waves 0..3 take two rounds)
    if (p.gstat) {  // uniform: centred moments of the block's groups; a wave holds 2 tile slots x 32 channels
      const MeanM2 a = mm2_wave_group(mm2_from_shifted(gpv, gs1, gs2, nlane), cpg, nlane);
      if (lh == 0 && (lr & (cpg - 1)) == 0) {
        red[(wave * ng + lr / cpg) * 2] = a.m;
        red[(wave * ng + lr / cpg) * 2 + 1] = a.M2;
      }
    }
    __syncthreads();
    if (gnb && tid < 32) {  // the 24 slots of a channel, fixed order
      float a1 = redb[tid * 2], a2 = redb[tid * 2 + 1];
#pragma unroll
      for (int w = 1; w < 24; ++w) {
        a1 += redb[(w * 32 + tid) * 2];
        a2 += redb[(w * 32 + tid) * 2 + 1];
      }
      float* o = p.gnb_ws + (((int64_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * p.N + n0 + nb * 32 + tid) * 2;
      o[0] = a1;
      o[1] = a2;
    }
    if (p.gstat && tid < ng) {  // the 12 waves in fixed order: waves 0..3 hold 64 outputs per channel, the others 32
      MeanM2 a{red[tid * 2], red[tid * 2 + 1]};
      float na = 64.f * (float)cpg;
#pragma unroll
      for (int w = 1; w < 12; ++w) {
        const float nw = (w < 4 ? 64.f : 32.f) * (float)cpg;
        a = mm2_merge(a, na, MeanM2{red[(w * ng + tid) * 2], red[(w * ng + tid) * 2 + 1]}, nw);
        na += nw;
      }
      float* o = p.gstat + (((int64_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * p.gstat_groups + (n0 + nb * 32) / cpg + tid) * 2;
      o[0] = a.m;
      o[1] = a.M2;
    }
    __syncthreads();  // (the next block's accumulators overwrite the image and the scratch)
  }
#ifdef VAE_WINO4_TIMING
  WEDGE(7);
  if (tw) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tE[9] = __builtin_amdgcn_s_memrealtime();
    if (p.track && blockIdx.x < 8) {  // [workgroup][wave group][10 + 32] uint64
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.track) + (blockIdx.x * 3 + (threadIdx.x >> 8)) * 42;
#pragma unroll
      for (int i = 0; i < 10; ++i) o[i] = tE[i];
#pragma unroll
      for (int i = 0; i < 32; ++i) o[10 + i] = tS[i >> 2][i & 3];
    }
  }
#endif
}

}  // namespace

// forward / dgrad of a plain 3x3 stride-1 pad-1 layer in fp32, 16x32-pixel tiles, channel chunks of 8, 64 output channels per workgroup
bool conv3_wino4_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_F32 || a.A16 != nullptr || a.batch != 1 || a.alpha != 1.0f) return false;
  if (g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.tapmask != 0 || a.a_step > 1 || a.c_step > 1 || a.out_bf16) return false;
#ifndef VAE_WINO4_TIMING  // (the instrumented build writes its stamps through `track`)
  if (a.track != nullptr) return false;
#endif
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.mode == VAE_MODE_DGRAD && a.xf != VAE_XF_NONE) return false;
  if (g.Ho % FTH != 0 || g.Wo % FTW != 0 || a.K % FBK != 0 || a.K < 64 || a.K > 1024 || a.N % FBN != 0 || g.Cs < a.K) return false;
  if (!aligned16(a.A) || !aligned16(a.C)) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return false;
  if ((size_t)a.K * FPOS * a.N * 4u >= BUF_MAX) return false;
  return true;
}

// chunks per image of the statistics epilogue (0 = not available for these arguments)
int conv3_wino4_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / FTW) * (g.Ho / FTH);
}

// chunks per image of the GroupNorm-backward epilogue (0 = not available for these arguments)
int conv3_wino4_gnb_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (g.mode != VAE_MODE_DGRAD || a.gnb_x == nullptr || a.gnb_groups <= 0 || a.N % a.gnb_groups != 0 || a.ldc != a.N) return 0;
  if (a.res != nullptr || a.bias != nullptr || a.out_bf16) return 0;
  if ((size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return 0;
  return (g.Wo / FTW) * (g.Ho / FTH);
}

int launch_wino4_weights(const vae_igemm_args& a, float* U, hipStream_t st) {
  const int64_t n = (int64_t)a.N * a.K;
  hipLaunchKernelGGL(wino4_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.W, a.N, a.K, a.g.mode == VAE_MODE_DGRAD ? 1 : 0,
                     a.sn, a.sk, a.st, U);
  return 0;
}

template <int XF>
static int launch_wino4_t(const vae_igemm_args& a, const float* U, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / FTW, ty = g.Ho / FTH;
  const int tilesN = a.N / FBN;
  const int64_t nt = (int64_t)tilesN * tx * ty * g.B;
  if (nt > 0x7fffffffLL) return VAE_EINVAL;
  constexpr size_t xcd_u = (size_t)9 << 20;  // U images up to 9 MB (36 positions; 128 and 256 channels): the channel blocks of a tile share an XCD
  const int xcd_sp = (tilesN > 1 && (size_t)a.K * FPOS * a.N * 4u <= xcd_u && ((int64_t)tx * ty * g.B) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((conv3_wino4_kernel<XF>), dim3((unsigned)nt), dim3(FNT), 0, st, a, U, tx, ty, xcd_sp);
  return 0;
}

int launch_conv3_wino4(const vae_igemm_args& a, const float* U, hipStream_t st) {
  if (a.xf == VAE_XF_NONE) return launch_wino4_t<VAE_XF_NONE>(a, U, st);
  if (a.xf == VAE_XF_AFFINE) return launch_wino4_t<VAE_XF_AFFINE>(a, U, st);
  return launch_wino4_t<VAE_XF_AFFINE_SILU>(a, U, st);
}
