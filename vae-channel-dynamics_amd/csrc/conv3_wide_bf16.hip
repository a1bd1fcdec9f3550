// 3x3 stride-1 convolution, forward and dgrad, for bf16 mode when BOTH operands already are bf16 images in HBM
// (forward: the transformed activation image A16 of vae_gn_apply_bf16; dgrad: the gradient image a bf16 GroupNorm
// backward / dgrad left, also passed as A16) and the weights come from their bf16 image Wh.  Output fp32 or bf16 (out_bf16),
// both with the bias, residual (stored like the output) and GroupNorm-statistics epilogues.
//
// Why another kernel next to conv3_tile_bf16.hip: that one feeds 64x64 wave tiles from a 128-pixel halo and re-stages
// the whole 128-channel weight tile per 128 pixels -- one KB of LDS operand reads per MFMA plus 24 KB of weight writes per
// 24 MFMAs per wave, i.e. the LDS array ~75 % busy and the matrix pipe 35-45 % (profiles/r01_pmc_mfma_bf16.txt).
// Here a workgroup (4 waves, ONE per SIMD, the whole 512-entry register file each) owns 8 rows x 32 pixels x 128
// channels and a wave 4 rows x 32 pixels x 64 channels (128 accumulator registers):
//   * a pipeline stage is one kernel COLUMN kw (its 3 taps kh) of a 32-channel chunk.  Per 16-channel k-group a wave
//     reads 6 halo-row fragments (rows 4wm .. 4wm+5 at column shift kw) and 6 weight fragments (3 kh x 2 channel blocks)
//     and issues 24 MFMAs: output row r meets halo row r + kh for every kh, so each halo fragment serves up to 3 kernel
//     rows and each weight fragment 4 output rows -- 0.5 KB of LDS reads per MFMA instead of 1 KB;
//   * the weight stage (3 x 128 x 32) is written once per 256 pixels instead of once per 128: half the LDS writes and
//     half the L2 weight traffic per MFMA; the 10 x 34 halo has 1.33 staged pixels per output pixel instead of 1.6;
//   * with one wave per SIMD nothing hides behind another wave, so the instruction stream is laid out by hand: every
//     group of 4 MFMAs is followed in program order by its share of the stage's other work (2 fragment reads for the
//     next k-group, one weight store or load, one halo load or store), pinned with sched_barrier;
//   * three weight stages in a ring and two halo buffers: the data of stage s+1 is visible before stage s ends, so the
//     first fragments of the next stage are read BEFORE the barrier and no LDS round trip is exposed behind it;
//   * persistent workgroups, one software pipeline across channel chunks AND tiles (the first two weight stages and the
//     first halo of the next tile are staged during the last chunk of the current one).
// KS = 2 serves the phase convolutions of an upsampler (vaehip.h, tapmask / a_step / c_step): a 2x2 block {kh0, kh0+1} x
// {kw0, kw0+1} of the 3x3 window on the low-resolution grid -- 2 stages per chunk, 2 kernel rows per stage (16 of the 36
// tap-MACs per low-resolution pixel are computed, none masked), operand pixel (y,x) at (y*a_step+a_oy, x*a_step+a_ox) of the
// image, output pixel at (y*c_step+c_oy, x*c_step+c_ox).
#include "bf16_frag.h"
#ifndef VAE_ABLATE
#define VAE_ABLATE 0
#endif
#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int BK = 32, TH = 8, TW = 32, HW_ = TW + 2, HROWS = TH + 2, HP = HROWS * HW_;  // 340 halo pixels
constexpr int LDH = BK + 8;        // halo pixel stride in bf16 (80 B: conflict-free 16-byte row reads)
constexpr int SH = HP * LDH;       // one halo buffer (13600 bf16 = 27200 B)
constexpr int BN = 128, NT = 256;
constexpr int LDBK = BK + 8;       // forward weight tile [n][k]
constexpr int LDBN = BN + 32;      // dgrad weight tile [k][n] (320 B rows: the transposing reads are conflict-free)
constexpr int SB1 = BN * LDBK;     // one tap (5120 bf16); BN * LDBK == BK * LDBN
static_assert(BN * LDBK == BK * LDBN, "forward and dgrad weight tiles have the same LDS size");
constexpr int SB = 3 * SB1;        // one stage: the 3 taps of a kernel column (30720 B)
constexpr int LDS_BYTES = (2 * SH + 3 * SB) * 2;  // 146560 B
constexpr int HQ = HP * (BK / 8);                 // 1360 x 16 B halo slots
constexpr int HI = (HQ + NT - 1) / NT;            // 6 per thread

struct Tile { int b, y0, x0, n0, lin; };

template <bool DG, int KS>
__global__ __launch_bounds__(NT, 1) void conv3_wide_bf16_kernel(vae_igemm_args p, int tiles_x, int tiles_y, int ntiles, int kh0, int kw0) {
  constexpr int NW = KS * BN * BK / 8 / NT;  // 16-byte weight pieces per thread and stage (6 / 4)
  constexpr int NPK = 2 * KS;                // pieces (of 4 MFMAs) per k-group: KS kernel rows x 2 output-row pairs
  constexpr int NA = 3 + KS, NB = 2 * KS;    // halo-row / weight fragments per k-group
  static_assert(KS == 2 || KS == 3, "3x3 kernels or their 2x2 phase blocks");
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* const sHalo = smem;            // [2][SH]
  u16* const sW = smem + 2 * SH;      // [3][SB]
  constexpr int LDB = DG ? LDBN : LDBK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const vae_conv_geom g = p.g;
  const int tilesN = (p.N + BN - 1) / BN;
  const int nch = (p.K + BK - 1) / BK;
  const int as = (KS == 2 && p.a_step > 1) ? p.a_step : 1, cs = (KS == 2 && p.c_step > 1) ? p.c_step : 1;
  const size_t img_bytes = (size_t)(g.Hs * as) * (g.Ws * as) * g.Cs * 2u;
  const auto rsW = VAE_BUF_RSRC(p.Wh, (size_t)(DG ? p.K * p.sk : p.N * p.sn) * 2u);

  const int G = gridDim.x;
  const int first = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;  // neighbours share an XCD's L2
  auto decode = [&](int t) {
    Tile id;
    id.lin = t / tilesN;
    const int tn = t - id.lin * tilesN;
    int r = id.lin;
    const int tx = r % tiles_x; r /= tiles_x;
    const int ty = r % tiles_y;
    id.b = r / tiles_y;
    id.y0 = ty * TH; id.x0 = tx * TW; id.n0 = tn * BN;
    return id;
  };

  f32x16 acc[4][2];  // [output row of the wave][32-channel block]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][ni][e] = 0.f;

  // ---------------- staging: global -> registers -> LDS, one piece (16 B per thread) at a time ----------------
  // weight stream: the position of the NEXT stage to request (tile, chunk, kernel column)
  // two register sets for the weight stream: a set is requested during the first half of stage s and stored during the second
  // half of stage s+1 (1.5 stages = ~1 us of lead: an L2 hit takes 0.5-0.8 us under load; half a stage was not enough)
  uint4 rw[2][NW], rh[HI];
  int w_t = first, w_c = 0, w_kw = 0, w_n0 = 0;  // w_kw: column index inside the tap block
  bool w_ok = first < ntiles;
  if (w_ok) w_n0 = decode(first).n0;
  auto w_load_piece = [&](int set, int i) {  // piece i of the stage at the stream position
    const int kh = i >> 1, rem = tid + NT * (i & 1);
    const int tap = (kh0 + kh) * 3 + kw0 + w_kw, c0 = w_c * BK;
    unsigned off;
    if (!DG) {
      const int n = w_n0 + (rem >> 2), c = c0 + (rem & 3) * 8;
      off = (w_ok && n < p.N && c < p.K) ? (unsigned)((n * (int)p.sn + tap * (int)p.st + c) * 2) : BUF_OOB;
    } else {
      const int k = c0 + (rem >> 4), n = w_n0 + (rem & 15) * 8;
      off = (w_ok && k < p.K && n < p.N) ? (unsigned)((k * (int)p.sk + tap * (int)p.st + n) * 2) : BUF_OOB;
    }
    rw[set][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0));
  };
  auto w_advance = [&]() {
    if (++w_kw == KS) {
      w_kw = 0;
      if (++w_c == nch) {
        w_c = 0;
        w_t += G;
        w_ok = w_t < ntiles;
        if (w_ok) w_n0 = decode(w_t).n0;
      }
    }
  };
  auto w_store_piece = [&](int set, int i, u16* sB) {
    const int rem = tid + NT * (i & 1);
    u16* dst = sB + (i >> 1) * SB1 + (DG ? (rem >> 4) * LDB + (rem & 15) * 8 : (rem >> 2) * LDB + (rem & 3) * 8);
    *reinterpret_cast<uint4*>(dst) = rw[set][i];
  };
  // halo stream: the NEXT channel chunk to request
  int h_t = first, h_c = 0;
  Tile h_id = decode(first < ntiles ? first : 0);
  bool h_ok = first < ntiles;
  auto h_load_piece = [&](int i) {
    const int q = tid + NT * i;
    const int pp = q >> 2, k8 = q & 3;
    const int ir = pp / HW_, jc = pp - ir * HW_;
    const int hy = h_id.y0 - 1 + ir, hx = h_id.x0 - 1 + jc;
    const int c = h_c * BK + k8 * 8;
    const bool ok = h_ok && (q < HQ) && ((unsigned)hy < (unsigned)g.Hs) && ((unsigned)hx < (unsigned)g.Ws) && (c < p.K);
    const auto rsA = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.A16) + (int64_t)h_id.b * (g.Hs * as) * (g.Ws * as) * g.Cs, img_bytes);
    const unsigned off = (KS == 2) ? (unsigned)((((hy * as + p.a_oy) * (g.Ws * as) + hx * as + p.a_ox) * g.Cs + c) * 2)
                                   : (unsigned)(((hy * g.Ws + hx) * g.Cs + c) * 2);
    rh[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : BUF_OOB, 0, 0));
  };
  auto h_advance = [&]() {
    if (++h_c == nch) {
      h_c = 0;
      h_t += G;
      h_ok = h_t < ntiles;
      if (h_ok) h_id = decode(h_t);
    }
  };
  auto h_store_piece = [&](int i, u16* sH) {
    const int q = tid + NT * i;
    if (q < HQ) *reinterpret_cast<uint4*>(&sH[(q >> 2) * LDH + (q & 3) * 8]) = rh[i];
  };

  // ---------------- fragments ----------------
  // this lane's element offsets into a halo buffer and a weight stage (the rest are compile-time constants)
  // A slot j (0..NA-1) is halo row 4wm + j + roff: output row r meets kernel row kh0 + khi at halo row r + kh0 + khi
  // (forward, slot r + khi) or r + 2 - kh0 - khi (dgrad, slot r + KS-1-khi); the column shift of block column kwi is
  // dxb + kwi (forward) or dxb - kwi (dgrad)
  const int roff = DG ? 3 - KS - kh0 : kh0;
  const int dxb = DG ? 2 - kw0 : kw0;
  const int aoff = ((4 * wm + roff) * HW_ + lr + dxb) * LDH + lh * 8;
  const int boff = DG ? (lh * 8 + trq) * LDB + wn * 64 + trh * 16 + trp * 4 : (wn * 64 + lr) * LDB + lh * 8;
  bf16x8 fa[2][NA], fb[2][NB];  // two sets: the k-group being multiplied and the one being fetched
  // fragment j of A: slot j at block column kwi; of B: kernel row j >> 1 of the block, channel block j & 1
  auto fetch_a = [&](bf16x8* set, int j, const u16* sH, int kwi, int kg) { set[j] = frag_direct(sH + aoff + (j * HW_ + (DG ? -kwi : kwi)) * LDH + kg * 16); };
  auto fetch_b = [&](bf16x8* set, int j, const u16* sB, int kg) {
    const int kh = j >> 1, ni = j & 1;
    if (!DG) set[j] = frag_direct(sB + boff + kh * SB1 + ni * 32 * LDB + kg * 16);
    else set[j] = frag_tr(sB + boff + kh * SB1 + kg * 16 * LDB + ni * 32, LDB);
  };

  int t = first;
  if (t >= ntiles) return;  // uniform
  Tile cur = decode(t);
  int hpar = 0;  // halo buffer of the chunk being multiplied

  // prologue: the first two stages and the first halo into LDS, the third stage into registers
  u16* ringB[3] = {sW, sW + SB, sW + 2 * SB};  // [0] the stage being multiplied, [1] the next one, [2] the one being written
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(0, i);
  w_advance();
#pragma unroll
  for (int i = 0; i < HI; ++i) h_load_piece(i);
  h_advance();
#pragma unroll
  for (int i = 0; i < NW; ++i) w_store_piece(0, i, ringB[0]);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(0, i);
  w_advance();
#pragma unroll
  for (int i = 0; i < HI; ++i) h_store_piece(i, sHalo);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_store_piece(0, i, ringB[1]);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(1, i);  // the third stage: stored during the second half of the first stage
  w_advance();
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NA; ++j) fetch_a(fa[0], j, sHalo, 0, 0);
#pragma unroll
  for (int j = 0; j < NB; ++j) fetch_b(fb[0], j, ringB[0], 0);

  // One stage = block column KWI of the current chunk: 2 k-groups x NPK pieces; a piece = 4 MFMAs (kernel row q >> 1 of the
  // block, output rows 2(q&1), 2(q&1)+1, both channel blocks) followed by its share of the other work of the stage
  // (piece index pi = kg * NPK + q of 2 NPK):
  //   every piece        fragment q of A (the last piece also the remaining slots) and of B for the next k-group
  //                      (k-group 1: of the NEXT stage)
  //   pi < NW            request a piece of weight stage +3 into register set PAR
  //   pi >= 2 NPK - NW   store a piece of weight stage +2 (set PAR^1, requested 1.5 stages ago) into the ring slot stage -1 used
  //   KWI == 0           the first HI pieces request the next chunk's halo;  KWI == 1  the last HI pieces store it
  auto stage = [&](auto kw_c, auto par_c) {
    constexpr int KWI = decltype(kw_c)::value;
    constexpr int PAR = decltype(par_c)::value;  // parity of the stage counter: which weight register set is requested
    constexpr int KWN = (KWI + 1) % KS;
    const u16* sH = sHalo + hpar * SH;
    const u16* sHn = (KWI == KS - 1) ? sHalo + (hpar ^ 1) * SH : sH;  // the next stage's halo buffer
    const u16* sB = ringB[0];
    const u16* sBn = ringB[1];
    u16* sBw = ringB[2];
    u16* sHw = sHalo + (hpar ^ 1) * SH;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int q = 0; q < NPK; ++q) {
        const int khi = q >> 1, r0 = 2 * (q & 1);
        const int pi = kg * NPK + q;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[r0 + rr][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kg][r0 + rr + (DG ? KS - 1 - khi : khi)], fb[kg][khi * 2 + ni],
                                                                      acc[r0 + rr][ni], 0, 0, 0);
        if (kg == 0) {
          fetch_a(fa[1], q, sH, KWI, 1);
          if (q == NPK - 1)
#pragma unroll
            for (int j = NPK; j < NA; ++j) fetch_a(fa[1], j, sH, KWI, 1);
          fetch_b(fb[1], q, sB, 1);
        } else {
          // (KS == 2: the next chunk's halo is being STORED during this very stage when it is the chunk's last one -- a
          // 2-stage chunk has no stage between the store and the first read -- so those fragments wait for the barrier)
          if (!(KS == 2 && KWI == KS - 1)) {
            fetch_a(fa[0], q, sHn, KWN, 0);
            if (q == NPK - 1)
#pragma unroll
              for (int j = NPK; j < NA; ++j) fetch_a(fa[0], j, sHn, KWN, 0);
          }
          fetch_b(fb[0], q, sBn, 0);
        }
        // (VAE_ABLATE: diagnostic builds of tools/ablation_builds.sh, wrong results -- bit 0 no weight loads, 1 no weight LDS stores,
        // 2 no halo loads, 3 no halo LDS stores, 4 no barrier: what the staging costs)
        if (!(VAE_ABLATE & 1) && pi < NW) w_load_piece(PAR, pi);                                       // stage +3
        if (!(VAE_ABLATE & 2) && pi >= 2 * NPK - NW) w_store_piece(PAR ^ 1, pi - (2 * NPK - NW), sBw);  // stage +2, requested a stage and a half ago
        if (!(VAE_ABLATE & 4) && KWI == 0 && pi < HI) h_load_piece(pi);
        if (!(VAE_ABLATE & 8) && KWI == 1 && pi >= 2 * NPK - HI) h_store_piece(pi - (2 * NPK - HI), sHw);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    w_advance();
    if (KWI == 0) h_advance();
    u16* t0 = ringB[0];  // rotate the ring
    ringB[0] = ringB[1];
    ringB[1] = ringB[2];
    ringB[2] = t0;
    if (!(VAE_ABLATE & 16)) __syncthreads();
    if (KS == 2 && KWI == KS - 1) {
#pragma unroll
      for (int j = 0; j < NA; ++j) fetch_a(fa[0], j, sHn, KWN, 0);
    }
  };
  constexpr std::integral_constant<int, 0> kw0_c{};
  constexpr std::integral_constant<int, 1> kw1_c{};
  constexpr std::integral_constant<int, 2> kw2_c{};

  constexpr std::integral_constant<int, 0> p0{};
  constexpr std::integral_constant<int, 1> p1{};
#ifdef VAE_WIDE_TIMING  // debug build (tools/wide_timing.py): shader-clock stamps of workgroup 0's tiles into p.track (as uint64)
  unsigned long long* tstamp = reinterpret_cast<unsigned long long*>(p.track);
  int titer = 0;
#define TSTAMP(k) do { if (tstamp && blockIdx.x == 0 && tid == 0 && titer < 64) tstamp[titer * 4 + (k)] = clock64(); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif
  while (true) {
    TSTAMP(0);
    // the bias of the lane's channel pair, both channel blocks, requested BEFORE the tile's main loop: requested in the epilogue its
    // L2 round trip stood in front of the first store of every tile with nothing to hide it (one wave per SIMD) -- half of what
    // "bias + statistics" cost a 128-channel forward launch (16 tiles per workgroup)
    float pbe[2], pbo[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = cur.n0 + wn * 64 + ni * 32 + lr;
      pbe[ni] = (p.bias && col < p.N) ? p.bias[col & ~1] : 0.f;
      pbo[ni] = (p.bias && col < p.N) ? p.bias[col | 1] : 0.f;
    }
    if constexpr (KS == 3) {  // 3 stages per chunk: the stage parity repeats every 2 chunks (the launcher checks nch % 2 == 0)
      for (int c = 0; c < nch; c += 2) {
        stage(kw0_c, p0);
        stage(kw1_c, p1);
        stage(kw2_c, p0);
        hpar ^= 1;
        stage(kw0_c, p1);
        stage(kw1_c, p0);
        stage(kw2_c, p1);
        hpar ^= 1;
      }
    } else {
      for (int c = 0; c < nch; ++c) {
        stage(kw0_c, p0);
        stage(kw1_c, p1);
        hpar ^= 1;
      }
    }

    // ---------------- epilogue ----------------
    TSTAMP(1);
    const size_t obytes = (size_t)(g.Ho * cs) * (g.Wo * cs) * p.ldc * 4u;
    const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
    const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
    float gs1[4][2], gs2[4][2], gpv[4][2];  // statistics as shifted sums around the lane's first value
    if (p.out_bf16) {
      // bf16 output (uniform): adjacent lanes hold adjacent channels of the same 16 pixels; they swap every other register, so a
      // lane ends up with BOTH channels of its pair at 8 pixels, adds the bias of both (and the bf16 residual, res_bf16: 4-byte
      // loads) and rounds once.  The statistics epilogue below sums the ROUNDED values (the tensor as stored); a lane's 16 values
      // still belong to one group, so the group merge is the same as for fp32 outputs.
      // Without a residual (every dgrad, conv1 of a block) the rounded pairs then go through two more exchanges -- lanes 2 apart
      // (quad permute, 4 channels x 4 pixels), lanes 4 apart (bank-masked row shifts, 8 channels x 2 pixels) -- and leave as TWO
      // 16-byte stores per 32 x 32 block instead of eight 4-byte ones.  One wave per SIMD, nothing beside it: the 64 store
      // instructions of a tile cost 3.1k of its epilogue's 9.2k cycles and kept the next tile's first loads waiting (1.3-1.9k of its
      // main loop); a quarter as many 16-byte stores cost 1.0k (diagnostic builds, tools/wide_timing.py).  The variants are compiled
      // per (residual, statistics, wide stores): without them the residual and statistics arithmetic ran on zeros (21 vector
      // instructions per pair of values where a dgrad needs 11).
      const bool odd = lr & 1, bit1 = lr & 2;
      const size_t ob16 = (size_t)(g.Ho * cs) * (g.Wo * cs) * p.ldc * 2u;
      const auto rsC16 = VAE_BUF_RSRC(reinterpret_cast<u16*>(p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, ob16);
      const auto rsR16 = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.res ? p.res : p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, ob16);
      const unsigned pstep = (unsigned)(cs * p.ldc * 2);  // bytes per pixel step of the row grid
      // (the bias: once per tile, see the top of the tile loop.  Loaded per 16-pixel block, each load sat behind the previous block's
      // stores and its wait -- vmcnt counts stores too -- made every block wait for the write acknowledgements of the one before)
      const float pb0[2] = {pbe[0], pbe[1]}, pb1[2] = {pbo[0], pbo[1]};
      f32x2 sv1[2] = {{0.f, 0.f}, {0.f, 0.f}}, sv2[2] = {{0.f, 0.f}, {0.f, 0.f}};  // statistics: shifted sums of the lane's 64 values per channel block
      float spv[2] = {0.f, 0.f};
      auto epi16 = [&](auto res_c, auto gst_c, auto wide_c) {
        constexpr bool RES = decltype(res_c)::value, GST = decltype(gst_c)::value, WIDE = decltype(wide_c)::value;
        static_assert(!(RES && WIDE), "the residual is added before the rounding, in the 4-byte layout");
        // byte offset of a store of block (r, ni): a per-lane base plus a wave-uniform step -- the tiles are full (eligibility), only
        // a channel beyond N needs masking: its base is 2 GB, beyond any tensor the 32-bit offsets address, and stays out of range
        // when the step is added.  (The address arithmetic per store was a quarter of the epilogue's VALU work.)
        //   4-byte stores: the pair's two channels at pixel  odd + 4 lh + 2 (j & 1) + 8 (j >> 1),  j = 0..7
        //   16-byte stores: channels 8 (lr >> 3) .. + 7 at pixel  (lr & 3) + 4 lh + 8 ((lr >> 2) & 1) + 16 t,  t = 0, 1
        unsigned b2[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int oy = cur.y0 + 4 * wm + r, col = cur.n0 + wn * 64 + ni * 32 + lr;
            const int ox = cur.x0 + 4 * lh + (WIDE ? (lr & 3) + 2 * (lr & 4) : (odd ? 1 : 0));
            const int c0 = WIDE ? (col & ~7) : (col & ~1);
            b2[r][ni] = col < p.N ? (unsigned)((((oy * cs + (KS == 2 ? p.c_oy : 0)) * (g.Wo * cs) + ox * cs + (KS == 2 ? p.c_ox : 0)) * p.ldc + c0) * 2)
                                  : 0x80000000u;
          }
        auto off2 = [&](int r, int ni, int j) -> unsigned { return b2[r][ni] + (unsigned)(2 * (j & 1) + 8 * (j >> 1)) * pstep; };
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          unsigned rr[4][8];
          if constexpr (RES) {  // the residual of half the wave's tile in flight before any of it is consumed
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int j = 0; j < 8; ++j) rr[q][j] = __builtin_amdgcn_raw_buffer_load_b32(rsR16, off2(2 * hf + (q >> 1), q & 1, j), 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 2 * hf + (q >> 1), ni = q & 1;
            const float b0 = pb0[ni], b1 = pb1[ni];
            gs1[r][ni] = gs2[r][ni] = gpv[r][ni] = 0.f;
            unsigned P[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float a0 = acc[r][ni][2 * j], a1 = acc[r][ni][2 * j + 1];
              const float recv = lane_xor1(odd ? a0 : a1);
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              bf16x2_t h;
              float v0 = (odd ? recv : a0) + b0, v1 = (odd ? a1 : recv) + b1;
              if constexpr (RES) {
                v0 += __builtin_bit_cast(float, rr[q][j] << 16);
                v1 += __builtin_bit_cast(float, rr[q][j] & 0xffff0000u);
              }
              h[0] = (__bf16)v0;
              h[1] = (__bf16)v1;
              P[j] = __builtin_bit_cast(unsigned, h);
              if constexpr (!WIDE) __builtin_amdgcn_raw_buffer_store_b32(P[j], rsC16, off2(r, ni, j), 0, 0);
              if constexpr (GST) {  // ONE pivot for the lane's 64 values of a channel block (its first), two independent chains of sums
                const float q0 = (float)h[0], q1 = (float)h[1];
                if (r == 0 && j == 0) spv[ni] = q0;
                const float d0 = q0 - spv[ni], d1 = q1 - spv[ni];  // (the statistics epilogue only runs on full tiles)
                sv1[ni][0] += d0;
                sv1[ni][1] += d1;
                sv2[ni][0] = fmaf(d0, d0, sv2[ni][0]);
                sv2[ni][1] = fmaf(d1, d1, sv2[ni][1]);
              }
              acc[r][ni][2 * j] = 0.f;
              acc[r][ni][2 * j + 1] = 0.f;
            }
            if constexpr (WIDE) {
              // lanes 2 apart: lane (lr & 2) == 0 keeps the even j of each (j, j + 1) and gets its neighbour's: 4 channels at pixel j
              unsigned Q[4][2];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const unsigned A = P[2 * jj], Bv = P[2 * jj + 1];
                const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)(bit1 ? A : Bv), 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
                Q[jj][0] = bit1 ? got : A;
                Q[jj][1] = bit1 ? Bv : got;
              }
              // lanes 4 apart (the neighbouring bank of the row): bank-masked row shifts write only the lanes that receive
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                  const unsigned X = Q[2 * t][k], Y = Q[2 * t + 1][k];
                  o[k] = (unsigned)__builtin_amdgcn_update_dpp((int)X, (int)Y, 0x114, 0xF, 0xA, false);      // banks 1, 3: Y of lane - 4
                  o[2 + k] = (unsigned)__builtin_amdgcn_update_dpp((int)Y, (int)X, 0x104, 0xF, 0x5, false);  // banks 0, 2: X of lane + 4
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rsC16, b2[r][ni] + (unsigned)(16 * t) * pstep, 0, 0);
              }
            }
          }
        }
      };
      constexpr std::true_type yes{};
      constexpr std::false_type no{};
      const bool wide_st = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15u) == 0);
      if (p.res) {  // uniform
        if (p.gstat) epi16(yes, yes, no); else epi16(yes, no, no);
      } else if (wide_st) {
        if (p.gstat) epi16(no, yes, yes); else epi16(no, no, yes);
      } else {
        if (p.gstat) epi16(no, yes, no); else epi16(no, no, no);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {  // (the rows of a channel block are already merged: slot 0 carries all 64 values)
        gpv[0][ni] = spv[ni];
        gs1[0][ni] = sv1[ni][0] + sv1[ni][1];
        gs2[0][ni] = sv2[ni][0] + sv2[ni][1];
      }
    } else {
      // fp32 output (+ bias, + residual).  The residual is HBM-cold: its loads are issued for HALF of the wave's tile (two
      // rows x two channel blocks, 64 registers) before any of them is consumed -- with one wave per SIMD nothing else hides
      // that latency; issued per 16-element block they cost 8 round trips per tile (46 % of the 128-channel layers' time)
      auto offset = [&](int r, int ni, int e) -> unsigned {
        const int oy = cur.y0 + 4 * wm + r, col = cur.n0 + wn * 64 + ni * 32 + lr;
        const int ox = cur.x0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (KS == 2)
          return (col < p.N && oy < g.Ho && ox < g.Wo) ? (unsigned)((((oy * cs + p.c_oy) * (g.Wo * cs) + ox * cs + p.c_ox) * p.ldc + col) * 4) : BUF_OOB;
        return (col < p.N && oy < g.Ho && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + col) * 4) : BUF_OOB;
      };
      const bool oddl = lr & 1;
      const float pbv[2] = {oddl ? pbo[0] : pbe[0], oddl ? pbo[1] : pbe[1]};  // (once per tile: see the top of the tile loop)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float rv[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[q][e] = 0.f;
        if (p.res) {  // uniform
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              rv[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, offset(2 * hf + (q >> 1), q & 1, e), 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 2 * hf + (q >> 1), ni = q & 1;
          const float bv = pbv[ni];
          gs1[r][ni] = gs2[r][ni] = gpv[r][ni] = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = acc[r][ni][e] + bv + rv[q][e];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, offset(r, ni, e), 0, 0);
            if (e == 0) gpv[r][ni] = v;
            const float dv = v - gpv[r][ni];  // (the statistics epilogue only runs on full tiles)
            gs1[r][ni] += dv;
            gs2[r][ni] += dv * dv;
            acc[r][ni][e] = 0.f;
          }
        }
      }
    }
    TSTAMP(2);
    if (p.gstat) {  // uniform: centred moments (mean, M2) of this tile's outputs per group, layout of vae_gn_stats_partial with one
      // chunk per wave-row band of the tile (4 rows x 32 pixels): the lane's four rows are merged in place, the group's lanes by
      // DPP moves, and the first lane of a group writes -- no LDS round trip, no workgroup barrier (the tile-level merge of round 2
      // cost 3.5 us per tile, tools/wide_timing.py)
      const int cpg = p.N / p.gstat_groups;  // channels per group (4, 8 or 16)
      const int tile_in_img = cur.lin - cur.b * (tiles_x * tiles_y);
      float* gbase = p.gstat + (((int64_t)cur.b * (tiles_x * tiles_y) + tile_in_img) * 2 + wm) * p.gstat_groups * 2;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        MeanM2 lane64;
        if (p.out_bf16) {  // uniform: one set of shifted sums over the lane's 64 values
          lane64 = mm2_from_shifted(gpv[0][ni], gs1[0][ni], gs2[0][ni], 64.f);
        } else {
          const MeanM2 r01 = mm2_merge_equal(mm2_from_shifted(gpv[0][ni], gs1[0][ni], gs2[0][ni], 16.f),
                                             mm2_from_shifted(gpv[1][ni], gs1[1][ni], gs2[1][ni], 16.f), 16.f);
          const MeanM2 r23 = mm2_merge_equal(mm2_from_shifted(gpv[2][ni], gs1[2][ni], gs2[2][ni], 16.f),
                                             mm2_from_shifted(gpv[3][ni], gs1[3][ni], gs2[3][ni], 16.f), 16.f);
          lane64 = mm2_merge_equal(r01, r23, 32.f);
        }
        const MeanM2 a = mm2_wave_group(lane64, cpg, 64.f);
        if (lh == 0 && (lr & (cpg - 1)) == 0) {
          float* o = gbase + ((cur.n0 + wn * 64 + ni * 32 + lr) / cpg) * 2;
          o[0] = a.m;
          o[1] = a.M2;
        }
      }
    }
    TSTAMP(3);
#ifdef VAE_WIDE_TIMING
    ++titer;
#endif
    t += G;
    if (t >= ntiles) break;
    cur = decode(t);
  }
}

}  // namespace

// a tap mask that is a 2x2 block {kh0, kh0+1} x {kw0, kw0+1} of the 3x3 window (the phase convolutions of an upsampler)
static bool phase_block(int tapmask, int* kh0, int* kw0) {
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int m = 0;
      for (int kh = a; kh < a + 2; ++kh)
        for (int kw = b; kw < b + 2; ++kw) m |= 1 << (kh * 3 + kw);
      if (m == tapmask) {
        *kh0 = a;
        *kw0 = b;
        return true;
      }
    }
  return false;
}

// both operands as bf16 images, forward or dgrad of a 3x3 stride-1 layer -- plain, or one phase convolution of an upsampler
// (2x2 tap block, sub-sampled operand / output view) -- whose spatial size the 8 x 32 tile divides, with enough tiles to
// give every CU one
bool conv3_wide_bf16_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || a.A16 == nullptr || a.Wh == nullptr || a.xf != VAE_XF_NONE) return false;
  if (a.batch != 1 || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || a.alpha != 1.0f) return false;
#ifndef VAE_WIDE_TIMING
  if (a.track != nullptr) return false;
#endif
  // a residual input on a 128-channel contraction: the tile's main loop (12 stages, ~9 us) is shorter than what one CU needs
  // to pull the 128 KB residual tile and push the 128 KB output (~10 us at a CU's ~26 GB/s), and with one workgroup per CU
  // nothing overlaps the two -- the 128-pixel kernel's second workgroup does (measured 0.505 vs 0.588 ms at 128->128 @256^2)
  if (a.res != nullptr && a.K <= 128 && a.tapmask == 0) return false;
  const bool phase = a.tapmask != 0 || a.a_step > 1 || a.c_step > 1;
  if (phase) {
    int kh0, kw0;
    if (!phase_block(a.tapmask, &kh0, &kw0) || a.gstat) return false;
    if ((a.a_step > 1 && a.a_step != 2) || (a.c_step > 1 && a.c_step != 2)) return false;
  }
  const size_t as = a.a_step > 1 ? a.a_step : 1, cs = a.c_step > 1 ? a.c_step : 1;
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.Wo % TW != 0 || g.Ho % TH != 0 || a.K % (2 * BK) != 0 || a.N % 8 != 0 || a.N <= 32 || g.Cs % 8 != 0 || a.st % 8 != 0) return false;
  if (g.mode == VAE_MODE_FWD && !(a.sk == 1 && a.sn % 8 == 0)) return false;
  if (g.mode == VAE_MODE_DGRAD && !(a.sn == 1 && a.sk % 8 == 0)) return false;
  // storage: the operand comes as an image (A16; a_bf16 is the flat kernels' flag); a bf16 output takes a bf16 residual, an fp32 one an fp32 one
  if (a.a_bf16 || (a.out_bf16 && a.ldc % 2 != 0) || (a.res != nullptr && (a.res_bf16 != 0) != (a.out_bf16 != 0))) return false;
  if (!aligned16(a.A16) || !aligned16(a.Wh)) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u * as * as >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u * cs * cs >= BUF_MAX) return false;
  if ((size_t)std::max((int64_t)a.K * a.sk, (int64_t)a.N * a.sn) * 2u >= BUF_MAX) return false;
  const int64_t nt = (int64_t)((a.N + BN - 1) / BN) * (g.Wo / TW) * (g.Ho / TH) * g.B;
  return nt >= 192 && nt <= 0x7fffffffLL;
}
int conv3_wide_bf16_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % BN != 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD || a.c_step > 1 || a.tapmask != 0) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / TW) * (g.Ho / 4);  // one chunk per wave-row band of a tile (4 rows x 32 pixels x the group's channels)
}

template <bool DG, int KS>
static int launch_wide(const vae_igemm_args& a, int tx, int ty, int64_t nt, int kh0, int kw0, hipStream_t st) {
  auto kern = conv3_wide_bf16_kernel<DG, KS>;
  VAE_RESERVE_LDS(kern, LDS_BYTES, "conv3_wide_bf16");
  // persistent: one 4-wave workgroup per CU; library option "wide_reserved_cus" leaves that many CUs free (RCCL's workgroups
  // under a data-parallel backward pass): tiles are dealt t = blockIdx.x, += gridDim.x, so any grid size covers them all
  dim3 grid((unsigned)std::min<int64_t>(nt, 256 - vae_opt().wide_reserved_cus));
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS_BYTES, st, a, tx, ty, (int)nt, kh0, kw0);
  return 0;
}

int launch_conv3_wide_bf16(const vae_igemm_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / TW, ty = g.Ho / TH;
  const int64_t nt = (int64_t)((a.N + BN - 1) / BN) * tx * ty * g.B;
  const bool dg = g.mode == VAE_MODE_DGRAD;
  int kh0 = 0, kw0 = 0;
  if (a.tapmask != 0) {
    if (!phase_block(a.tapmask, &kh0, &kw0)) return VAE_EINVAL;
    return dg ? launch_wide<true, 2>(a, tx, ty, nt, kh0, kw0, st) : launch_wide<false, 2>(a, tx, ty, nt, kh0, kw0, st);
  }
  return dg ? launch_wide<true, 3>(a, tx, ty, nt, 0, 0, st) : launch_wide<false, 3>(a, tx, ty, nt, 0, 0, st);
}
