// Winograd F(2x2, 3x3) for the fp32 forward and dgrad of the 3x3 stride-1 layers: 16 multiplications per 2x2 output tile and
// channel pair instead of 36 (2.25x fewer MFMA passes), exact fp32 products on v_mfma_f32_32x32x2_f32 as everywhere in fp32 mode.
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch, g: 3x3 kernel, Y: 2x2 outputs
// The element-wise product summed over input channels is, per position (xi, nu) of the 4x4 transform domain, a GEMM
//     M[pos][tile][co] = sum_ci V[pos][tile][ci] * U[pos][co][ci]
// so a workgroup runs 16 GEMMs of [32 tiles] x [128 co] over the channel chunks.
//   * U = G g G^T is built once per launch by vae_wino_weights from the LIVE weights (they change every optimizer step and
//     the nudger edits parameters in place), laid out [K/8][16][N][8]: a workgroup's slab of one channel chunk is 16
//     contiguous 4 KB pieces.  For dgrad the kernel is rotated and transposed there (N = Cin, K = Cout).
//   * Workgroup (8 waves) = 8 x 16 output pixels (32 Winograd tiles) x 64 output channels (NB = 2 channel blocks), channel
//     chunk of 8 per step; 64 accumulator registers per wave, 128 VGPRs, so TWO workgroups share a CU and fill each other's
//     barrier / load bubbles (NB = 4: 128 channels, 256 VGPRs, one workgroup per CU, was 4-10 % slower and is not built).
//     Threads 0..359 load the chunk's 10 x 18 input halo once (GroupNorm + SiLU applied once per element) into LDS two steps
//     ahead; threads 0..255 each own one (tile, channel) of the chunk, take B^T d B of the 4x4 patch (32 additions) and write
//     the 16 values into the V image [16][32][8] (double buffered).  Wave w multiplies positions 2w, 2w+1: per position ONE
//     16-byte A fragment row read covers the chunk (k = 4*half + s over 4 MFMAs) against NB B fragments (channel blocks):
//     8 NB MFMAs per wave and step.  The GroupNorm scale / shift rows of the image sit in LDS (read when a halo is stored).
//   * The B fragment of (position, channel block) is 16 contiguous bytes per lane of the U image, and only the wave that owns
//     the position needs it: U goes from L2 straight into registers, never through LDS -- ONE register set: the fragment for
//     step s+1 is requested right behind the four MFMAs that consumed step s's.  One barrier per step (it publishes the next
//     V); the two waves of a SIMD run the step in opposite order (stage then multiply / multiply then stage) so the matrix
//     pipe has work while V is being transformed -- as two separate copies of the loop, which keeps hipcc's wait counts exact.
//   * Epilogue: the 16 positions of an output tile sit in 8 different waves, so the accumulators go through LDS one
//     32-channel block at a time ([16][32 tiles][32 co], over the V buffers), then each thread takes A^T M A of its
//     (tile, channel) pairs, adds bias / residual and writes the 2x2 outputs (lanes along channels: 128-byte rows).
// Numerics: the transforms only add / subtract / halve; measured against the direct kernels 1e-6 of the output scale
// (tests/test_kernels_gpu.py), far inside the 1e-4 bar.  The epilogue also leaves the GroupNorm centred moments of the outputs
// (one chunk per workgroup tile, as the direct kernels do) and -- dgrad launches with gnb_* set -- the first pass of the GroupNorm
// backward over the gradient it has just computed; a tracked output stays on the direct kernel.
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr int WTH = 8, WTW = 16;          // output pixels per workgroup tile
constexpr int NTL = 32;                   // Winograd tiles per workgroup (4 rows x 8 columns of 2x2 outputs)
constexpr int WBK = 8;                    // channels per step
constexpr int WNT = 512;
constexpr int SV = 16 * NTL * WBK;        // floats per V buffer (16384 B): rows of 8 floats, a wave's 16-byte fragment reads
                                          // cover 1 KB contiguously and the transform's 4-byte writes are thread-contiguous
constexpr int SMLD = 33;                  // epilogue image row (floats)
constexpr int SHL = 180 * WBK;            // floats per halo buffer (10 x 18 pixels x 8 channels, 5760 B)
constexpr int SM = 16 * NTL * SMLD + 8 * 8 * 2 + 16 * 32 * 2;  // epilogue image + statistics scratch + GroupNorm-backward sums; overlays the V / halo buffers
constexpr int WSS = 2 * 1024;             // GroupNorm scale / shift rows of the image (K <= 1024)
constexpr int WINO_LDS = (SM > 2 * SV + 2 * SHL + WSS ? SM : 2 * SV + 2 * SHL + WSS) * 4;  // 72192 B

// U[pos][n][k] = (G g G^T)[pos] for g = W[n][.][.][k] (forward, N = Cout, K = Cin) or g = rot180(W[k][.][.][n]) (dgrad,
// N = Cin, K = Cout); output layout [K/8][16][N][8]
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ W, int N, int K, int dgrad, int64_t sn, int64_t sk, int64_t st,
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
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
    t[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
    t[3][b] = g[2][b];
  }
  float* o = U + ((int64_t)(k >> 3) * 16 * N + n) * 8 + (k & 7);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0], u3 = t[a][2];
    const float u1 = 0.5f * ((t[a][0] + t[a][2]) + t[a][1]);
    const float u2 = 0.5f * ((t[a][0] + t[a][2]) - t[a][1]);
    o[(int64_t)(a * 4 + 0) * N * 8] = u0;
    o[(int64_t)(a * 4 + 1) * N * 8] = u1;
    o[(int64_t)(a * 4 + 2) * N * 8] = u2;
    o[(int64_t)(a * 4 + 3) * N * 8] = u3;
  }
}

template <int XF, int NB>
__global__ __launch_bounds__(WNT, (NB == 4 ? 2 : 4)) void conv3_wino_kernel(vae_igemm_args p, const float* __restrict__ U, int tiles_x, int tiles_y, int xcd_sp) {
  constexpr int WBN = 32 * NB;  // output channels per workgroup
  __shared__ __attribute__((aligned(16))) float wsm[WINO_LDS / 4];
  float* const sV = wsm;           // [2][SV]
  float* const sH = wsm + 2 * SV;  // [2][SHL]: the chunk's input halo, transformed
#ifdef VAE_WINO_TIMING  // debug build (tools/wino_timing.py): shader-clock stamps of waves 0 and 4 of workgroups 0..7, per step, into p.track
  __shared__ unsigned long long sT[2][64][6];
  __shared__ unsigned long long sE[2][10];  // kernel entry, loop entry, loop exit, per channel block: accumulators in LDS / outputs stored, end
#define WEDGE(k) do { if ((threadIdx.x & 255) == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sE[threadIdx.x >> 8][k] = __builtin_amdgcn_s_memtime(); } } while (0)
#if VAE_WINO_TIMING >= 2  // per-step stamps too (more intrusive: every stamp waits for the wave's LDS operations)
#define WSTAMP(k) do { if ((threadIdx.x & 255) == 0 && wts < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sT[threadIdx.x >> 8][wts][k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif
  int wts = 0;
#else
#define WSTAMP(k) do { } while (0)
#define WEDGE(k) do { } while (0)
#endif
  WEDGE(0);
#ifdef VAE_WINO_TIMING
  if ((threadIdx.x & 255) == 0) sE[threadIdx.x >> 8][8] = __builtin_amdgcn_s_memrealtime();  // 100 MHz: the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = (p.N + WBN - 1) / WBN;
  // Workgroup id -> (spatial tile, channel block).  Consecutive ids go round-robin over the 8 XCDs (one L2 each).  With tn
  // fastest an XCD sees one channel block's slice of U (what fits its L2) but every XCD pulls the whole input: Cout/64-fold
  // L2 fills.  When the whole U image is small (<= 4 MB: 128 and 256 channels) the channel
  // blocks of a spatial tile get ids congruent mod 8 instead, so one L2 fetches that tile's halo once.  Measured (rocprofv3
  // FETCH_SIZE/WRITE_SIZE, bytes per launch averaged over the step's 96 launches): tn fastest everywhere 1351 MB, this rule
  // 1142 MB, spatial-major everywhere (512 channels too: the U slices then cycle through L2) 1207 MB; same speed in all three.
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
  const int y0 = ty * WTH, x0 = tx * WTW, n0 = tn * WBN;
  const int nsteps = p.K / WBK;

  // ---- halo role (threads 0..359): pixel hp of the 10 x 18 input halo, channel quad hq of the chunk.  The chunk's halo is
  // loaded once (16 bytes per thread), GroupNorm(+SiLU) is applied once per element, and the result goes to sH two steps
  // ahead of its use; the V role then reads its 4x4 patches from LDS (up to 4 tiles share a pixel) ----
  const bool hrole = tid < 360;
  const int hp = tid >> 1, hq = tid & 1;
  const int hy = y0 - 1 + hp / 18, hx = x0 - 1 + hp % 18;
  const bool hin = hrole && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws);
  const auto rsA = VAE_BUF_RSRC(p.A + (int64_t)b * g.Hs * g.Ws * g.Cs, (size_t)g.Hs * g.Ws * g.Cs * 4u);
  const unsigned hbase = hin ? (unsigned)(((hy * g.Ws + hx) * g.Cs + hq * 4) * 4) : BUF_OOB;
  // GroupNorm scale / shift rows of image b: staged in LDS once per workgroup ([2][K], behind the halo buffers), read at store time
  float* const sS = wsm + 2 * SV + 2 * SHL;
  if (XF != VAE_XF_NONE) {
    for (int i = tid; i < p.K; i += WNT) {
      sS[i] = p.scale[(int64_t)b * g.Cs + i];
      sS[p.K + i] = p.shift[(int64_t)b * g.Cs + i];
    }
    __syncthreads();
  }
  struct Halo {
    f32x4 v;
  };
  Halo rh{{0.f, 0.f, 0.f, 0.f}};
  auto load_halo_into = [&](int step, Halo& h) {
    const bool ok = hin && step < nsteps;
    h.v = VAE_BUF_LOAD4(rsA, ok ? hbase + (unsigned)(step * WBK * 4) : BUF_OOB);
  };
  auto store_halo_from = [&](float* dst, const Halo& h, int step) {
    if (!hrole) return;
    f32x4 v = h.v;
    if (XF != VAE_XF_NONE) {
      const int c = min(step, nsteps - 1) * WBK + hq * 4;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(&sS[c]), sh = *reinterpret_cast<const f32x4*>(&sS[p.K + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * sc[e] + sh[e];
        if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
        v[e] = (hin && step < nsteps) ? u : 0.f;  // padding stays zero AFTER the transform
      }
    }
    *reinterpret_cast<f32x4*>(&dst[hp * WBK + hq * 4]) = v;
  };
  auto load_halo = [&](int step) { load_halo_into(step, rh); };
  auto store_halo = [&](float* dst, int step) { store_halo_from(dst, rh, step); };

  // ---- V role (threads 0..255): tile vt, channel vc of the chunk: B^T d B of its 4x4 patch of the staged halo ----
  const bool vrole = tid < 256;
  const int vt = (tid >> 3) & 31, vc = tid & 7;
  const int vorg = ((2 * (vt >> 3)) * 18 + 2 * (vt & 7)) * WBK + vc;  // patch origin in the halo image
  auto write_v = [&](const float* sHc, float* dst) {
    if (!vrole) return;
    float d[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i * 4 + j] = sHc[vorg + (i * 18 + j) * WBK];
    float r[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // rows: B^T d
      r[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
      r[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
      r[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
      r[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // columns: (B^T d) B
      dst[((i * 4 + 0) * NTL + vt) * WBK + vc] = r[i * 4 + 0] - r[i * 4 + 2];
      dst[((i * 4 + 1) * NTL + vt) * WBK + vc] = r[i * 4 + 1] + r[i * 4 + 2];
      dst[((i * 4 + 2) * NTL + vt) * WBK + vc] = r[i * 4 + 2] - r[i * 4 + 1];
      dst[((i * 4 + 3) * NTL + vt) * WBK + vc] = r[i * 4 + 1] - r[i * 4 + 3];
    }
  };

  // ---- U fragments: the B operand of position pos, channel block nb is U[step][pos][n0 + 32 nb + lr][4 lh .. 4 lh + 3]: 16
  // contiguous bytes of the transformed-weight image per lane, 1 KB per wave and (pos, nb).  Only the wave that owns the
  // position reads them, so they go from global memory (L2) straight into registers, one step ahead, and never touch LDS ----
  const auto rsU = VAE_BUF_RSRC(U, (size_t)nsteps * 16 * p.N * 8 * 4u);
  // address = per-thread constant (one per channel block; out of range for channels beyond N) + a workgroup-uniform part in
  // a scalar register: one instruction per load
  unsigned bvo[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) bvo[q] = (n0 + q * 32 + lr < p.N) ? (unsigned)(((n0 + q * 32 + lr) * 8 + lh * 4) * 4) : BUF_OOB;
  const unsigned bpos = (unsigned)p.N * 32u;  // bytes per position of the U image
  // ONE register set: the fragment of (position pi, channel block nb) for step s+1 is requested right behind the four MFMAs
  // that consume step s's (two sets -- 32 registers -- did not fit the 128-register budget of two workgroups per CU next to the
  // accumulators and the staging role: 7 registers spilled).  Every step issues the same 2 NB requests; beyond the last chunk
  // the last one is requested again and never used.
  f32x4 bq[2 * NB];
  auto load_b1 = [&](int step, int i) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(min(step, nsteps - 1) * 16 + 2 * wave) * bpos);
    bq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, bvo[i % NB], so + (i / NB) * bpos, 0));
  };

  f32x16 acc[2][NB];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pi][nb][e] = 0.f;

  // prologue: halo(0), halo(1) in LDS, V(0) from halo(0); halo(2) and the U fragments of step 0 in registers
#pragma unroll
  for (int i = 0; i < 2 * NB; ++i) load_b1(0, i);
  {  // the three halo requests of the prologue go out together (one memory latency, not two)
    Halo h0 = rh, h1 = rh;
    load_halo_into(0, h0);
    load_halo_into(1, h1);
    load_halo(2);
    store_halo_from(sH, h0, 0);
    store_halo_from(sH + SHL, h1, 1);
  }
  __syncthreads();
  write_v(sH, sV);
  __syncthreads();
  // The two waves of a SIMD (w and w+4) run the step in OPPOSITE order so that one multiplies while the other stages: waves
  // 0..3 (which own the V transform) stage first and multiply afterwards, waves 4..7 multiply first.  Step s: V(s+1) from
  // halo(s+1) [published by the previous barrier]; halo(s+2) -> the buffer halo(s) left; requests for the U fragments of step
  // s+1 (into the register set step s-1 used) and halo(s+3).  One barrier per step: it publishes V(s+1) and halo(s+2).
  auto multiply = [&](const f32x4* a4, int s) {  // step s; requests step s+1's fragments as it goes
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[pi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[pi][e], bq[pi * NB + nb][e], acc[pi][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_b1(s + 1, pi * NB + nb);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  auto stage_next = [&](int s, int par) {
    if (s + 1 < nsteps) write_v(sH + (par ^ 1) * SHL, sV + (par ^ 1) * SV);  // V(s+1): nobody reads that buffer now
    store_halo(sH + par * SHL, s + 2);                                       // halo(s+2)
    load_halo(s + 3);
  };
  // Waves 0..3 and 4..7 run SEPARATE copies of the loop (same number of barriers): inside one copy the order of the memory
  // requests is fixed, so hipcc's wait counts are exact -- with both orders in one loop body it merged the two states at every
  // join and put vmcnt(0) in front of the staging and at the loop head: each step then waited for the round trip of the
  // requests it had just issued.
  auto step = [&](int s, int par, auto first_c) {
    constexpr bool STAGE_FIRST = decltype(first_c)::value;
    const float* cV = sV + par * SV;
    f32x4 a4[2];
    auto read_a = [&]() {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) a4[pi] = *reinterpret_cast<const f32x4*>(&cV[((2 * wave + pi) * NTL + lr) * WBK + 4 * lh]);
    };
    WSTAMP(0);
    WSTAMP(1);
    if (STAGE_FIRST) {
      stage_next(s, par);
      WSTAMP(2);
      read_a();
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(3);
      multiply(a4, s);
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(4);
    } else {
      read_a();
      WSTAMP(2);
      multiply(a4, s);
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(3);
      stage_next(s, par);
      WSTAMP(4);
    }
    __syncthreads();
    WSTAMP(5);
#ifdef VAE_WINO_TIMING
    ++wts;
#endif
  };
  WEDGE(1);
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

  WEDGE(2);
#if defined(VAE_WINO_TIMING) && VAE_WINO_TIMING >= 2
  if (p.track && blockIdx.x < 8 && (tid & 255) < 64) {  // [workgroup][wave 0 / 4][step][6] as uint64; entry [.][.][63][5] = end of the kernel (below)
    unsigned long long* out = reinterpret_cast<unsigned long long*>(p.track) + (blockIdx.x * 2 + (tid >> 8)) * 64 * 6;
    for (int i = tid & 255; i < 64 * 6; i += 64) out[i] = (i / 6 < wts) ? sT[tid >> 8][i / 6][i % 6] : 0ull;
  }
#endif
  // ---- epilogue: per 32-channel block, M through LDS, then Y = A^T M A ----
  float* const sM = wsm;  // [16][32 tiles][SMLD], over the V / halo buffers (the last step's barrier has passed)
  const size_t obytes = (size_t)g.Ho * g.Wo * p.ldc * 4u;
  const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)b * g.Ho * g.Wo * p.ldc, obytes);
  const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)b * g.Ho * g.Wo * p.ldc, obytes);
  // GroupNorm-backward epilogue (dgrad launches, vaehip.h gnb_*): x at this thread's output positions, stored fp32 or bf16
  const bool gnb = p.gnb_ws != nullptr;  // uniform
  const unsigned xes = p.gnb_x_bf16 ? 2u : 4u;
  const auto rsX = VAE_BUF_RSRC(reinterpret_cast<const char*>(gnb ? p.gnb_x : (const void*)p.C) + (int64_t)b * g.Ho * g.Wo * p.ldc * xes,
                                (size_t)g.Ho * g.Wo * p.ldc * xes);
  // byte offsets of this thread's 8 outputs per channel block (2 tile slots x 2 x 2 pixels, channel n0 + (tid & 31))
  unsigned offp[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int tile = (tid >> 5) + 16 * (q >> 2), oy = y0 + 2 * (tile >> 3) + ((q >> 1) & 1), ox = x0 + 2 * (tile & 7) + (q & 1);
    offp[q] = (oy < g.Ho && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + n0 + (tid & 31)) * 4) : BUF_OOB;
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    // the residual values of the block are requested before the LDS round trip below, not one by one in front of each store
    const bool cok = n0 + nb * 32 + (tid & 31) < p.N;
    unsigned off[8];
    float rres[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      off[q] = (cok && offp[q] != BUF_OOB) ? offp[q] + nb * 128u : BUF_OOB;
      rres[q] = p.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off[q], 0, 0)) : 0.f;
    }
    float xin[8];  // the GroupNorm input at the same 8 positions (requested here, used after the LDS round trip)
    if (gnb) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned eo = off[q] == BUF_OOB ? BUF_OOB : (p.gnb_x_bf16 ? off[q] >> 1 : off[q]);
        xin[q] = p.gnb_x_bf16 ? __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsX, eo, 0, 0) << 16)
                              : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, eo, 0, 0));
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int pos = 2 * wave + pi;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tile = (e & 3) + 8 * (e >> 2) + 4 * lh;
        sM[(pos * NTL + tile) * SMLD + lr] = acc[pi][nb][e];
      }
    }
    __syncthreads();
    WEDGE(3 + 2 * nb);
    float gpv = 0.f, gs1 = 0.f, gs2 = 0.f;  // GroupNorm statistics of this thread's 8 outputs of channel `col`: shifted sums
    float bs1 = 0.f, bs2 = 0.f;             // GroupNorm backward: sum dz, sum dz * xhat over the same 8 outputs
    float bmu = 0.f, brs = 0.f, bga = 0.f, bbe = 0.f;
    if (gnb) {
      const int colb = min(n0 + nb * 32 + (tid & 31), p.N - 1);
      const int grp = colb / (p.N / p.gnb_groups);
      bmu = p.gnb_mean[b * p.gnb_groups + grp];
      brs = p.gnb_rstd[b * p.gnb_groups + grp];
      bga = p.gnb_gamma[colb];
      bbe = p.gnb_beta[colb];
    }
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      const int co = tid & 31, tile = (tid >> 5) + 16 * rnd;
      float m[16];
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) m[pos] = sM[(pos * NTL + tile) * SMLD + co];
      float h[8];  // A^T M: rows {0: m0+m1+m2, 1: m1-m2-m3} per column
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[0 * 4 + j] = (m[0 * 4 + j] + m[1 * 4 + j]) + m[2 * 4 + j];
        h[1 * 4 + j] = (m[1 * 4 + j] - m[2 * 4 + j]) - m[3 * 4 + j];
      }
      const int col = n0 + nb * 32 + co;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float yv[2] = {(h[a * 4 + 0] + h[a * 4 + 1]) + h[a * 4 + 2], (h[a * 4 + 1] - h[a * 4 + 2]) - h[a * 4 + 3]};
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int q = rnd * 4 + a * 2 + bb;
          float v = yv[bb] + bv;
          if (p.res) v += rres[q];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off[q], 0, 0);
          if (rnd == 0 && a == 0 && bb == 0) gpv = v;
          const float dv = v - gpv;  // (the statistics epilogue only runs on full tiles)
          gs1 += dv;
          gs2 += dv * dv;
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
    float* const red = sM + 16 * NTL * SMLD;  // [8 waves][groups of the 32-channel block][2]
    float* const redb = red + 8 * 8 * 2;      // [16 tile slots][32 channels][2]
    if (gnb) {
      redb[((tid >> 5) * 32 + (tid & 31)) * 2] = bs1;
      redb[((tid >> 5) * 32 + (tid & 31)) * 2 + 1] = bs2;
    }
    const int cpg = p.gstat ? p.N / p.gstat_groups : 4, ng = 32 / cpg;
    if (p.gstat) {  // uniform: centred moments of the block's groups; a wave holds 2 tile slots x 32 channels
      const MeanM2 a = mm2_wave_group(mm2_from_shifted(gpv, gs1, gs2, 8.f), cpg, 8.f);
      if (lh == 0 && (lr & (cpg - 1)) == 0) {
        red[(wave * ng + lr / cpg) * 2] = a.m;
        red[(wave * ng + lr / cpg) * 2 + 1] = a.M2;
      }
    }
    __syncthreads();
    if (gnb && tid < 32 && n0 + nb * 32 + tid < p.N) {  // the 16 tile slots of a channel, fixed order
      float a1 = redb[tid * 2], a2 = redb[tid * 2 + 1];
#pragma unroll
      for (int w = 1; w < 16; ++w) {
        a1 += redb[(w * 32 + tid) * 2];
        a2 += redb[(w * 32 + tid) * 2 + 1];
      }
      float* o = p.gnb_ws + (((int64_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * p.N + n0 + nb * 32 + tid) * 2;
      o[0] = a1;
      o[1] = a2;
    }
    if (p.gstat && tid < ng) {  // the 8 waves (16 outputs x cpg channels each), fixed order
      const float nw = 16.f * (float)cpg;
      MeanM2 a{red[tid * 2], red[tid * 2 + 1]};
#pragma unroll
      for (int w = 1; w < 8; ++w) a = mm2_merge(a, nw * (float)w, MeanM2{red[(w * ng + tid) * 2], red[(w * ng + tid) * 2 + 1]}, nw);
      float* o = p.gstat + (((int64_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * p.gstat_groups + (n0 + nb * 32) / cpg + tid) * 2;
      o[0] = a.m;
      o[1] = a.M2;
    }
  }
#ifdef VAE_WINO_TIMING
  WEDGE(7);
  if ((tid & 255) == 0) sE[tid >> 8][9] = __builtin_amdgcn_s_memrealtime();
  if (p.track && blockIdx.x < 8 && (tid & 255) < 10)
    reinterpret_cast<unsigned long long*>(p.track)[8 * 2 * 64 * 6 + (blockIdx.x * 2 + (tid >> 8)) * 10 + (tid & 255)] = sE[tid >> 8][tid & 255];
#endif
}

}  // namespace

// forward / dgrad of a plain 3x3 stride-1 pad-1 layer in fp32, 8x16-pixel tiles, channel chunks of 8
bool conv3_wino_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_F32 || a.A16 != nullptr || a.batch != 1 || a.alpha != 1.0f) return false;
  if (g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1) return false;
  if (a.tapmask != 0 || a.a_step > 1 || a.c_step > 1 || a.out_bf16) return false;
#ifndef VAE_WINO_TIMING  // (the instrumented build writes its stamps through `track`)
  if (a.track != nullptr) return false;
#endif
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.mode == VAE_MODE_DGRAD && a.xf != VAE_XF_NONE) return false;
  if (g.Ho % WTH != 0 || g.Wo % WTW != 0 || a.K % WBK != 0 || a.K < 64 || a.K > 1024 || a.N < 32 || a.N % 4 != 0 || g.Cs < a.K) return false;
  if (!aligned16(a.A) || !aligned16(a.C)) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return false;
  if ((size_t)a.K * 16 * a.N * 4u >= BUF_MAX) return false;
  return true;
}

// channel blocks (of 32) per workgroup: 2 = 64 channels, 128 VGPRs per wave, TWO workgroups per CU (the template also
// instantiates with 4 = 128 channels, one workgroup per CU: measured 4-10 % slower in round 2 and no longer built)
int conv3_wino_nb() { return 2; }

// chunks per image of the statistics epilogue (0 = not available for these arguments)
int conv3_wino_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % (32 * conv3_wino_nb()) != 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / WTW) * (g.Ho / WTH);
}

// chunks per image of the GroupNorm-backward epilogue (0 = not available for these arguments): full tiles of a dgrad launch
// whose output has the GroupNorm input's shape
int conv3_wino_gnb_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (g.mode != VAE_MODE_DGRAD || a.gnb_x == nullptr || a.gnb_groups <= 0 || a.N % a.gnb_groups != 0 || a.ldc != a.N) return 0;
  if (a.res != nullptr || a.bias != nullptr || a.out_bf16) return 0;
  if ((size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return 0;
  return (g.Wo / WTW) * (g.Ho / WTH);
}

int launch_wino_weights(const vae_igemm_args& a, float* U, hipStream_t st) {
  const int64_t n = (int64_t)a.N * a.K;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.W, a.N, a.K, a.g.mode == VAE_MODE_DGRAD ? 1 : 0,
                     a.sn, a.sk, a.st, U);
  return 0;
}

template <int XF, int NB>
static int launch_wino_t(const vae_igemm_args& a, const float* U, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / WTW, ty = g.Ho / WTH;
  const int64_t nt = (int64_t)((a.N + 32 * NB - 1) / (32 * NB)) * tx * ty * g.B;
  if (nt > 0x7fffffffLL) return VAE_EINVAL;
  constexpr size_t xcd_u = (size_t)4 << 20;  // U images up to 4 MB: the channel blocks of a tile share an XCD (lowest measured traffic)
  const int tilesN = (a.N + 32 * NB - 1) / (32 * NB);
  const int xcd_sp = (tilesN > 1 && (size_t)a.K * 16 * a.N * 4u <= xcd_u && ((int64_t)tx * ty * g.B) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((conv3_wino_kernel<XF, NB>), dim3((unsigned)nt), dim3(WNT), 0, st, a, U, tx, ty, xcd_sp);
  return 0;
}

int launch_conv3_wino(const vae_igemm_args& a, const float* U, hipStream_t st) {
  if (a.xf == VAE_XF_NONE) return launch_wino_t<VAE_XF_NONE, 2>(a, U, st);
  if (a.xf == VAE_XF_AFFINE) return launch_wino_t<VAE_XF_AFFINE, 2>(a, U, st);
  return launch_wino_t<VAE_XF_AFFINE_SILU, 2>(a, U, st);
}
