// 3x3 stride-1 convolution, forward and dgrad, for bf16 mode when BOTH operands already are bf16 images in HBM
// (forward: the transformed activation image A16 of vae_gn_apply_bf16; dgrad: the gradient image a bf16 GroupNorm
// backward / dgrad left, also passed as A16) and the weights come from their bf16 image Wh.  Output fp32 (bias,
// residual, tracker and GroupNorm-statistics epilogues) or bf16 (out_bf16, dgrad).
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
#include "bf16_frag.h"
#include <algorithm>
#include <type_traits>

namespace {

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
constexpr int NW = 3 * BN * BK / 8 / NT;          // 6 x 16 B of weights per thread and stage
constexpr int HQ = HP * (BK / 8);                 // 1360 x 16 B halo slots
constexpr int HI = (HQ + NT - 1) / NT;            // 6 per thread

struct Tile { int b, y0, x0, n0, lin; };

template <bool DG>
__global__ __launch_bounds__(NT, 1) void conv3_wide_bf16_kernel(vae_igemm_args p, int tiles_x, int tiles_y, int ntiles) {
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
  const size_t img_bytes = (size_t)g.Hs * g.Ws * g.Cs * 2u;
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
  uint4 rw[NW], rh[HI];
  int w_t = first, w_c = 0, w_kw = 0, w_n0 = 0;
  bool w_ok = first < ntiles;
  if (w_ok) w_n0 = decode(first).n0;
  auto w_load_piece = [&](int i) {  // piece i of the stage at the stream position
    const int kh = i >> 1, rem = tid + NT * (i & 1);
    const int tap = kh * 3 + w_kw, c0 = w_c * BK;
    unsigned off;
    if (!DG) {
      const int n = w_n0 + (rem >> 2), c = c0 + (rem & 3) * 8;
      off = (w_ok && n < p.N && c < p.K) ? (unsigned)((n * (int)p.sn + tap * (int)p.st + c) * 2) : BUF_OOB;
    } else {
      const int k = c0 + (rem >> 4), n = w_n0 + (rem & 15) * 8;
      off = (w_ok && k < p.K && n < p.N) ? (unsigned)((k * (int)p.sk + tap * (int)p.st + n) * 2) : BUF_OOB;
    }
    rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0));
  };
  auto w_advance = [&]() {
    if (++w_kw == 3) {
      w_kw = 0;
      if (++w_c == nch) {
        w_c = 0;
        w_t += G;
        w_ok = w_t < ntiles;
        if (w_ok) w_n0 = decode(w_t).n0;
      }
    }
  };
  auto w_store_piece = [&](int i, u16* sB) {
    const int rem = tid + NT * (i & 1);
    u16* dst = sB + (i >> 1) * SB1 + (DG ? (rem >> 4) * LDB + (rem & 15) * 8 : (rem >> 2) * LDB + (rem & 3) * 8);
    *reinterpret_cast<uint4*>(dst) = rw[i];
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
    const auto rsA = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.A16) + (int64_t)h_id.b * g.Hs * g.Ws * g.Cs, img_bytes);
    rh[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (unsigned)(((hy * g.Ws + hx) * g.Cs + c) * 2) : BUF_OOB, 0, 0));
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
  const int aoff = ((4 * wm) * HW_ + lr) * LDH + lh * 8;
  const int boff = DG ? (lh * 8 + trq) * LDB + wn * 64 + trh * 16 + trp * 4 : (wn * 64 + lr) * LDB + lh * 8;
  bf16x8 fa[2][6], fb[2][6];  // two sets: the k-group being multiplied and the one being fetched
  // fragment j (0..5) of A: halo row 4wm + j, column shift dx; of B: kernel row j >> 1, channel block j & 1
  auto fetch_a = [&](bf16x8* set, int j, const u16* sH, int dx, int kg) { set[j] = frag_direct(sH + aoff + (j * HW_ + dx) * LDH + kg * 16); };
  auto fetch_b = [&](bf16x8* set, int j, const u16* sB, int kg) {
    const int kh = j >> 1, ni = j & 1;
    if (!DG) set[j] = frag_direct(sB + boff + kh * SB1 + ni * 32 * LDB + kg * 16);
    else set[j] = frag_tr(sB + boff + kh * SB1 + kg * 16 * LDB + ni * 32, LDB);
  };

  int t = first;
  if (t >= ntiles) return;  // uniform
  Tile cur = decode(t);
  int hpar = 0;  // halo buffer of the chunk being multiplied

  // prologue: stages (0,0) and (0,1) and the first halo into LDS, stage (0,2) into registers
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(i);
  w_advance();
#pragma unroll
  for (int i = 0; i < HI; ++i) h_load_piece(i);
  h_advance();
#pragma unroll
  for (int i = 0; i < NW; ++i) w_store_piece(i, sW);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(i);
  w_advance();
#pragma unroll
  for (int i = 0; i < HI; ++i) h_store_piece(i, sHalo);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_store_piece(i, sW + SB);
#pragma unroll
  for (int i = 0; i < NW; ++i) w_load_piece(i);
  w_advance();
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    fetch_a(fa[0], j, sHalo, DG ? 2 : 0, 0);
    fetch_b(fb[0], j, sW, 0);
  }

  // One stage = kernel column KW of the current chunk: 2 k-groups x 6 pieces; a piece = 4 MFMAs (kernel row q >> 1, output
  // rows 2(q&1), 2(q&1)+1, both channel blocks) followed by its share of the other work of the stage:
  //   every piece      fragment j = q of A and of B for the next k-group (k-group 1: of the NEXT stage)
  //   pieces 0..5      store piece q of weight stage +2 (requested a stage ago) into the ring slot that stage -1 used
  //   pieces 6..11     request piece q of weight stage +3
  //   KW == 0          pieces 0..5 request the next chunk's halo;  KW == 1  pieces 6..11 store it into the other buffer
  auto stage = [&](auto kw_c) {
    constexpr int KW = decltype(kw_c)::value;
    constexpr int KWN = (KW + 1) % 3;
    const u16* sH = sHalo + hpar * SH;
    const u16* sHn = (KW == 2) ? sHalo + (hpar ^ 1) * SH : sH;  // the next stage's halo buffer
    const u16* sB = sW + KW * SB;
    const u16* sBn = sW + KWN * SB;
    u16* sBw = sW + ((KW + 2) % 3) * SB;
    u16* sHw = sHalo + (hpar ^ 1) * SH;
    constexpr int dxn = DG ? 2 - KWN : KWN;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int kh = q >> 1, r0 = 2 * (q & 1);
        const int dy = DG ? 2 - kh : kh;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[r0 + rr][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kg][r0 + rr + dy], fb[kg][kh * 2 + ni], acc[r0 + rr][ni], 0, 0, 0);
        if (kg == 0) {
          fetch_a(fa[1], q, sH, DG ? 2 - KW : KW, 1);
          fetch_b(fb[1], q, sB, 1);
          w_store_piece(q, sBw);
          if (KW == 0) h_load_piece(q);
        } else {
          fetch_a(fa[0], q, sHn, dxn, 0);
          fetch_b(fb[0], q, sBn, 0);
          w_load_piece(q);
          if (KW == 1) h_store_piece(q, sHw);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    w_advance();
    if (KW == 0) h_advance();
    __syncthreads();
  };
  constexpr std::integral_constant<int, 0> kw0{};
  constexpr std::integral_constant<int, 1> kw1{};
  constexpr std::integral_constant<int, 2> kw2{};

  while (true) {
    for (int c = 0; c < nch; ++c) {
      stage(kw0);
      stage(kw1);
      stage(kw2);
      hpar ^= 1;
    }

    // ---------------- epilogue ----------------
    float* scratch = reinterpret_cast<float*>(sHalo + (hpar ^ 1) * SH);  // the halo buffer of the chunk just finished
    const size_t obytes = (size_t)g.Ho * g.Wo * p.ldc * 4u;
    const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)cur.b * g.Ho * g.Wo * p.ldc, obytes);
    const auto rsC16 = VAE_BUF_RSRC(reinterpret_cast<u16*>(p.C) + (int64_t)cur.b * g.Ho * g.Wo * p.ldc, obytes / 2);
    const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)cur.b * g.Ho * g.Wo * p.ldc, obytes);
    float gs1[4][2], gs2[4][2], gpv[4][2];  // statistics as shifted sums around the lane's first value
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oy = cur.y0 + 4 * wm + r;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        gs1[r][ni] = gs2[r][ni] = gpv[r][ni] = 0.f;
        const int col = cur.n0 + wn * 64 + ni * 32 + lr;
        const bool colok = col < p.N && oy < g.Ho;
        if (p.out_bf16) {  // uniform: adjacent lanes (adjacent channels) swap every other register: 4-byte stores
          const bool odd = lr & 1;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a0 = acc[r][ni][2 * j], a1 = acc[r][ni][2 * j + 1];
            const float recv = __shfl_xor(odd ? a0 : a1, 1, 64);
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            bf16x2_t h;
            h[0] = (__bf16)(odd ? recv : a0);
            h[1] = (__bf16)(odd ? a1 : recv);
            const int e = 2 * j + (odd ? 1 : 0);
            const int ox = cur.x0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const unsigned o16 = (colok && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + (col & ~1)) * 2) : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rsC16, o16, 0, 0);
            acc[r][ni][2 * j] = 0.f;
            acc[r][ni][2 * j + 1] = 0.f;
          }
          continue;
        }
        const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
        unsigned off[16];
        float rv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ox = cur.x0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          off[e] = (colok && ox < g.Wo) ? (unsigned)(((oy * g.Wo + ox) * p.ldc + col) * 4) : BUF_OOB;
          rv[e] = 0.f;
        }
        if (p.res) {  // uniform
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off[e], 0, 0));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[r][ni][e] + bv + rv[e];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off[e], 0, 0);
          if (e == 0) gpv[r][ni] = v;
          const float dv = v - gpv[r][ni];  // (the statistics epilogue only runs on full tiles)
          gs1[r][ni] += dv;
          gs2[r][ni] += dv * dv;
          acc[r][ni][e] = 0.f;
        }
      }
    }
    if (p.gstat) {  // uniform: centred moments (mean, M2) of this tile's outputs per group (layout of vae_gn_stats_partial, one chunk per tile)
      const int cpg = p.N / p.gstat_groups, gpt = BN / cpg;  // channels per group (4, 8 or 16), groups per 128-channel tile
      float* red2 = scratch;                                  // [8 rows][gpt][2]
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const MeanM2 a = mm2_wave_group(mm2_from_shifted(gpv[r][ni], gs1[r][ni], gs2[r][ni], 16.f), cpg, 16.f);
          if (lh == 0 && (lr & (cpg - 1)) == 0) {
            const int gl = (wn * 64 + ni * 32 + lr) / cpg;
            red2[((4 * wm + r) * gpt + gl) * 2] = a.m;
            red2[((4 * wm + r) * gpt + gl) * 2 + 1] = a.M2;
          }
        }
      __syncthreads();
      if (tid < gpt) {  // the 8 rows of the tile, fixed order; each holds 32 pixels x cpg channels
        const float nrow = 32.f * (float)cpg;
        MeanM2 a{red2[tid * 2], red2[tid * 2 + 1]};
#pragma unroll
        for (int rr = 1; rr < TH; ++rr) a = mm2_merge(a, nrow * (float)rr, MeanM2{red2[(rr * gpt + tid) * 2], red2[(rr * gpt + tid) * 2 + 1]}, nrow);
        const int tile_in_img = cur.lin - cur.b * (tiles_x * tiles_y);
        float* o = p.gstat + (((int64_t)cur.b * (tiles_x * tiles_y) + tile_in_img) * p.gstat_groups + cur.n0 / cpg + tid) * 2;
        o[0] = a.m;
        o[1] = a.M2;
      }
      __syncthreads();
    }

    t += G;
    if (t >= ntiles) break;
    cur = decode(t);
  }
}

}  // namespace

// both operands as bf16 images, forward or dgrad of a plain (no sub-sampled view / tap subset) 3x3 stride-1 layer whose
// spatial size the 8 x 32 tile divides, with enough tiles to give every CU one
bool conv3_wide_bf16_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || a.A16 == nullptr || a.Wh == nullptr || a.xf != VAE_XF_NONE) return false;
  if (a.batch != 1 || g.taps != 9 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || a.alpha != 1.0f) return false;
  if (a.tapmask != 0 || a.a_step > 1 || a.c_step > 1 || a.track != nullptr) return false;
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (g.Wo % TW != 0 || g.Ho % TH != 0 || a.K % 8 != 0 || a.N % 8 != 0 || a.N <= 32 || g.Cs % 8 != 0 || a.st % 8 != 0) return false;
  if (g.mode == VAE_MODE_FWD && !(a.sk == 1 && a.sn % 8 == 0)) return false;
  if (g.mode == VAE_MODE_DGRAD && !(a.sn == 1 && a.sk % 8 == 0)) return false;
  if (a.out_bf16 && (a.bias || a.res || a.track || a.gstat || a.ldc % 2 != 0)) return false;
  if (!aligned16(a.A16) || !aligned16(a.Wh)) return false;
  if ((size_t)g.Hs * g.Ws * g.Cs * 4u >= BUF_MAX || (size_t)g.Ho * g.Wo * a.ldc * 4u >= BUF_MAX) return false;
  if ((size_t)std::max((int64_t)a.K * a.sk, (int64_t)a.N * a.sn) * 2u >= BUF_MAX) return false;
  const int64_t nt = (int64_t)((a.N + BN - 1) / BN) * (g.Wo / TW) * (g.Ho / TH) * g.B;
  return nt >= 192 && nt <= 0x7fffffffLL;
}
int conv3_wide_bf16_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % BN != 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / TW) * (g.Ho / TH);
}
int launch_conv3_wide_bf16(const vae_igemm_args& a, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  const int tx = g.Wo / TW, ty = g.Ho / TH;
  const int64_t nt = (int64_t)((a.N + BN - 1) / BN) * tx * ty * g.B;
  static bool attr_set[2] = {false, false};
  const bool dg = g.mode == VAE_MODE_DGRAD;
  if (!attr_set[dg]) {
    const void* fn = dg ? reinterpret_cast<const void*>(conv3_wide_bf16_kernel<true>) : reinterpret_cast<const void*>(conv3_wide_bf16_kernel<false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      vae_set_error("conv3_wide_bf16: cannot reserve %d bytes of LDS", LDS_BYTES);
      return VAE_ELAUNCH;
    }
    attr_set[dg] = true;
  }
  dim3 grid((unsigned)std::min<int64_t>(nt, 256));  // persistent: one 4-wave workgroup per CU
  if (dg) hipLaunchKernelGGL(conv3_wide_bf16_kernel<true>, grid, dim3(NT), LDS_BYTES, st, a, tx, ty, (int)nt);
  else hipLaunchKernelGGL(conv3_wide_bf16_kernel<false>, grid, dim3(NT), LDS_BYTES, st, a, tx, ty, (int)nt);
  return 0;
}
