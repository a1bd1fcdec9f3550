// Blockwise (flash-style) single-head self-attention of the VAE mid blocks, head width d = 512, forward and backward,
// without the T x T score matrix (reference call sites: vae.encode / vae.decode, src/models/sdxl_vae_wrapper.py:60,71;
// at 1024x1024 the sequence is T = 16384 tokens and one fp32 score matrix per image would be 1 GB).
//
//   forward :  O = softmax(Q K^T * scale) V,  LSE = row log-sum-exp          (online softmax over key blocks)
//   backward:  P = exp(Q K^T * scale - LSE) is recomputed per block from the saved LSE; D = rowsum(dO * O)
//              dV = P^T dO;  dS = P * (dO V^T - D);  dQ = dS K * scale;  dK = dS^T Q * scale
//
// ONE kernel skeleton serves four roles.  A workgroup owns BR rows of the "resident" side R (queries for forward / dQ,
// keys for dV / dK), keeps their operands in REGISTERS as MFMA B fragments, and streams 32-row blocks of the other
// side X through LDS:
//     score tile  S[x][r]  = sum_d  Xs[x][d] * R1[r][d]        (A = LDS rows, read 16 B per lane; B = registers)
//     (dQ, dK)    dP[x][r] = sum_d  Xp[x][d] * R2[r][d]
//     W[x][r]     = elementwise(S, dP, statistics)               (P or dS; stays in registers)
//     acc^T[d][r] += sum_x  Xg[x][d] * W[x][r]                   (A = Xg^T through the transposing LDS read; B = W)
//   role      R (regs)    Xs     Xp    Xg     W     statistics
//   forward   Q           K      -     V      P     running max / sum per r (= per lane), online rescale of acc
//   dQ        Q, dO       K      V     K      dS    LSE[r], D[r] per lane
//   dV        K           Q      -     dO     P     LSE[x] per score row (LDS table)
//   dK        K, V        Q      dO    Q      dS    LSE[x], D[x] per score row
// The score tile is produced as [x][r] on purpose: its MFMA result layout (lane = column r, registers = rows x) IS the
// B-operand layout of the accumulating product, so P / dS never travel through LDS; the only twist is the order in which a
// lane's 8 k-slots enumerate x, which the transposing reads of Xg follow (x = (j&3) + 8*(j>>2) + 4*half + 16*kstep).
//
// The head is 512 wide, too wide for one wave's accumulators, so the waves of a workgroup split d: wave (rw, dw) owns
// 32 rows of R and a d-slice of 512/NDW for BOTH contractions.  The score contraction over d is therefore partial per
// wave: the NDW partial tiles are exchanged through LDS and every wave adds them in the same fixed order (bitwise the
// same scores, hence the same statistics, in all NDW waves).  A wave only ever touches its own d-slice of every tile.
//
// Precisions: bf16 (operands are bf16 images, v_mfma_f32_32x32x16_bf16; 8 waves, BR = 64) and fp32 (exact fp32 products on
// v_mfma_f32_32x32x2_f32; 4 waves, BR = 32); NDW = 4 d-slices of 128 in both.  Statistics, softmax and accumulators are fp32.
// LDS tiles are UNPADDED [32][512] with an XOR swizzle of the 16-byte chunk index by the row, chosen so that the
// 16-byte row reads (score contraction) and the transposing / column reads (accumulating product) of the SAME tile are
// both bank-conflict free; the exchange buffer aliases the tile that is dead after the score contraction when it fits.
#include "bf16_frag.h"

namespace {

constexpr int HD = 512;          // head width
constexpr int BX = 32;           // streamed rows per block
enum { ROLE_FWD = 0, ROLE_DQ = 1, ROLE_DV = 2, ROLE_DK = 3 };

struct AttnArgs {
  const void* R1; const void* R2;   // resident-side operands [B][T][512]
  const void* Xs; const void* Xp; const void* Xg;  // streamed-side operands [B][T][512] (Xg may equal Xs)
  const float* lse; const float* dsum;  // [B][T] statistics (per query)
  float* out;                        // [B][T][512]
  float* lse_out;                    // forward only
  int T; float scale;
};

// swizzled byte offset of 16-byte chunk `chunk` of row `row` in an unpadded [32][512] tile
template <int ESZ>
__device__ __forceinline__ int swz(int row, int chunk) {
  const int f = ((row & 3) << 2) | ((row >> 2) & 3);
  return row * (HD * ESZ) + (((chunk & ~15) | ((chunk & 15) ^ f)) << 4);
}

// bf16: 8 waves = 2 row groups x 4 d-slices (256 VGPRs per wave at 2 waves per SIMD); fp32: 4 waves = 1 row group x 4 d-slices,
// one wave per SIMD, so that the wave may use the whole 512-entry register file (fp32 fragments, tiles in flight and
// accumulators are twice as wide; the fp32 MFMA runs at 1/16 of the bf16 rate, a single wave keeps it busy)
template <bool BF> struct AttnShape {
  static constexpr int NW = BF ? 8 : 4, NDW = 4, NRW = NW / NDW, NT = NW * 64, DSL = HD / NDW, BR = 32 * NRW;
};

template <bool BF, int ROLE>
__global__ __launch_bounds__(AttnShape<BF>::NT, 1) void attn_kernel(AttnArgs p) {
  constexpr int ESZ = BF ? 2 : 4;
  constexpr int NW = AttnShape<BF>::NW, NDW = AttnShape<BF>::NDW, NRW = AttnShape<BF>::NRW, DSL = AttnShape<BF>::DSL, BR = AttnShape<BF>::BR;
  constexpr int ANT = AttnShape<BF>::NT;
  constexpr bool TWO = (ROLE == ROLE_DQ || ROLE == ROLE_DK);
  constexpr bool XSTAT = (ROLE == ROLE_DV || ROLE == ROLE_DK);   // statistics indexed by the streamed row
  constexpr bool G_IS_S = (ROLE == ROLE_DQ || ROLE == ROLE_DK);  // the accumulating product re-uses the score tile's operand
  constexpr int TILE = BX * HD * ESZ;                              // 32 KB / 64 KB
  constexpr int EXCH = NW * (TWO ? 2 : 1) * 16 * 256;              // partial tiles of all waves
  constexpr bool ALIAS = TILE >= EXCH;                             // exchange lives in the tile that is dead after the contraction
  constexpr int NTILE = G_IS_S ? 2 : 2;                            // tiles resident: {Xs|Xp-dead-tile, Xg}
  constexpr int CPR = HD * ESZ / 16;                               // 16-byte chunks per row
  constexpr int NLD = TILE / 16 / ANT;                             // 16-byte loads per thread and tile (4 / 8)
  constexpr int NDT = DSL / 32;                                    // 32-wide d tiles of the wave's accumulators (4 / 2)
  constexpr int NRF = BF ? DSL / 16 : DSL / 8;                     // resident fragments per operand (8 / 16), 4 VGPRs each

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // tile 0: the operand that is dead after the contractions (fwd: K, dQ: V, dV: Q, dK: dO); tile 1: the product's operand
  unsigned char* const sT0 = smem;
  unsigned char* const sT1 = smem + TILE;
  float* const sEX = reinterpret_cast<float*>(ALIAS ? smem : smem + NTILE * TILE);
  float* const sST = reinterpret_cast<float*>(smem + NTILE * TILE + (ALIAS ? 0 : EXCH));  // [2][32] LSE / D of the streamed block

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int rw = wave / NDW, dw = wave % NDW;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * BR + rw * 32;
  const int T = p.T;
  const int64_t img = (int64_t)b * T * HD;
  const int d0 = dw * DSL;
  const int nblk = T / BX;

  // operands of the two tiles, by role
  const unsigned char* g0 = reinterpret_cast<const unsigned char*>(ROLE == ROLE_FWD ? p.Xs : ROLE == ROLE_DQ ? p.Xp : ROLE == ROLE_DV ? p.Xs : p.Xp) + img * ESZ;
  const unsigned char* g1 = reinterpret_cast<const unsigned char*>(p.Xg) + img * ESZ;
  const auto rs0 = VAE_BUF_RSRC(g0, (size_t)T * HD * ESZ);
  const auto rs1 = VAE_BUF_RSRC(g1, (size_t)T * HD * ESZ);

  // ---- resident fragments (B operands): R1[r][d-slice] (and R2) ----
  uint4 rf1[NRF], rf2[TWO ? NRF : 1];
  {
    const unsigned char* R1 = reinterpret_cast<const unsigned char*>(p.R1) + (img + (int64_t)(r0 + lr) * HD + d0) * ESZ;
    const unsigned char* R2 = TWO ? reinterpret_cast<const unsigned char*>(p.R2) + (img + (int64_t)(r0 + lr) * HD + d0) * ESZ : nullptr;
#pragma unroll
    for (int i = 0; i < NRF; ++i) {
      rf1[i] = *reinterpret_cast<const uint4*>(R1 + i * 32 + lh * 16);  // bf16: k-step i, 8 channels; fp32: group i, 4 channels
      if (TWO) rf2[i] = *reinterpret_cast<const uint4*>(R2 + i * 32 + lh * 16);
    }
  }
  // per-lane statistics of the resident row (dQ) / running statistics (forward)
  float st_m = -INFINITY, st_l = 0.f;
  if (ROLE == ROLE_DQ) {
    st_m = p.lse[(int64_t)b * T + r0 + lr];
    st_l = p.dsum[(int64_t)b * T + r0 + lr];
  }

  f32x16 acc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- tile staging: global -> registers (one block ahead) -> swizzled LDS ----
  uint4 pre0[NLD], pre1[NLD];
  float pst = 0.f;  // LSE / D of the block in flight (threads 0..63)
  auto prefetch = [&](int blk) {
    const bool ok = blk < nblk;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const unsigned off = ok ? (unsigned)(((blk * BX * CPR) + i * ANT + tid) << 4) : BUF_OOB;
      pre0[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
      pre1[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
    }
    if (XSTAT && tid < 64) {
      const float* src = (tid < 32) ? p.lse : p.dsum;
      pst = (ok && (tid < 32 || TWO)) ? src[(int64_t)b * T + blk * BX + (tid & 31)] : 0.f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * ANT + tid;
      const int row = c / CPR, ch = c % CPR;
      *reinterpret_cast<uint4*>(sT0 + swz<ESZ>(row, ch)) = pre0[i];
      *reinterpret_cast<uint4*>(sT1 + swz<ESZ>(row, ch)) = pre1[i];
    }
    if (XSTAT && tid < 64) sST[tid] = pst;
  };

  // partial contraction over this wave's d-slice: tile rows x (A, 16 B per lane from LDS) against resident fragments
  auto contract = [&](const unsigned char* sT, const uint4* rf) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NRF; ++i) {
      const int chunk = d0 * ESZ / 16 + i * 2 + lh;
      const uint4 a = *reinterpret_cast<const uint4*>(sT + swz<ESZ>(lr, chunk));
      if (BF) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, rf[i]), s, 0, 0, 0);
      } else {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, rf[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], s, 0, 0, 0);
      }
    }
    return s;
  };
  auto ex_write = [&](int which, const f32x16& s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]};
      *reinterpret_cast<f32x4*>(sEX + (((which * NW + wave) * 4 + q) * 64 + lane) * 4) = v;
    }
  };
  auto ex_reduce = [&](int which, f32x16& s) {  // partials of the NDW waves of this row group (own included), fixed order
#pragma unroll
    for (int d2 = 0; d2 < NDW; ++d2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sEX + (((which * NW + rw * NDW + d2) * 4 + q) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * q + e] = (d2 == 0) ? v[e] : s[4 * q + e] + v[e];
      }
    }
  };
  // acc^T[d][r] += sum_x Xg[x][d] * W[x][r]; W in the score tile's register layout: register i of half lh is row
  // x = (i&3) + 8*(i>>2) + 4*lh
  auto product = [&](const f32x16& w) {
    if (BF) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wb;
#pragma unroll
        for (int j = 0; j < 8; ++j) wb[j] = (__bf16)w[8 * ks + j];
        const int row = 16 * ks + 4 * lh + trq;  // rows row..(+3 across the quad) and row+8..
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
          const int chunk = (d0 + t * 32) / 8 + 2 * trh + (trp >> 1);
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sT1 + swz<2>(row, chunk) + (trp & 1) * 8));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sT1 + swz<2>(row + 8, chunk) + (trp & 1) * 8));
          const s16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), wb, acc[t], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
          const int col = d0 + t * 32 + lr;
          const float a = *reinterpret_cast<const float*>(sT1 + swz<4>(row, col >> 2) + (col & 3) * 4);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[i], acc[t], 0, 0, 0);
        }
      }
    }
  };

  prefetch(0);
  stage();
  prefetch(1);
  __syncthreads();

  for (int blk = 0; blk < nblk; ++blk) {
    // ---- contractions over d (partial per wave) ----
    f32x16 s = contract(G_IS_S ? sT1 : sT0, rf1);
    f32x16 dp;
    if (TWO) dp = contract(sT0, rf2);
    if (ALIAS) __syncthreads();  // every wave is done with tile 0 before the exchange overwrites it
    ex_write(0, s);
    if (TWO) ex_write(1, dp);
    __syncthreads();
    ex_reduce(0, s);
    if (TWO) ex_reduce(1, dp);

    // ---- elementwise: P / dS in registers ----
    if (ROLE == ROLE_FWD) {
      float mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] *= p.scale;
        mloc = fmaxf(mloc, s[r]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(st_m, mloc);
      const float alpha = __expf(st_m - mnew);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __expf(s[r] - mnew);
        psum += s[r];
      }
      psum += __shfl_xor(psum, 32, 64);
      st_l = st_l * alpha + psum;
      st_m = mnew;
#pragma unroll
      for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float L = XSTAT ? sST[x] : st_m;
        const float pr = __expf(s[r] * p.scale - L);
        if (TWO) {
          const float Dv = XSTAT ? sST[32 + x] : st_l;
          s[r] = pr * (dp[r] - Dv);
        } else {
          s[r] = pr;
        }
      }
    }
    product(s);
    __syncthreads();  // tile 1 (and the exchange / statistics) fully consumed
    stage();
    prefetch(blk + 2);
    __syncthreads();
  }

  // ---- output: out[r][d] from acc^T[d][r] (lane = r, registers = d) ----
  float mul = 1.f;
  if (ROLE == ROLE_FWD) mul = 1.f / st_l;
  if (TWO) mul = p.scale;
  float* o = p.out + img + (int64_t)(r0 + lr) * HD + d0;
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[t][4 * q] * mul, acc[t][4 * q + 1] * mul, acc[t][4 * q + 2] * mul, acc[t][4 * q + 3] * mul};
      *reinterpret_cast<f32x4*>(o + t * 32 + 8 * q + 4 * lh) = v;
    }
  if (ROLE == ROLE_FWD && dw == 0 && lh == 0) p.lse_out[(int64_t)b * T + r0 + lr] = st_m + __logf(st_l);
}

// D[b][t] = sum_d dO[b][t][d] * O[b][t][d]  (one wave per row)
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ a, const float* __restrict__ bb, int64_t rows,
                                                          float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4* pa = reinterpret_cast<const f32x4*>(a + row * HD);
  const f32x4* pb = reinterpret_cast<const f32x4*>(bb + row * HD);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 x = pa[lane + 64 * i], y = pb[lane + 64 * i];
    s += (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

template <bool BF, int ROLE>
int launch_role(const AttnArgs& a, int B, hipStream_t st) {
  constexpr int ESZ = BF ? 2 : 4;
  constexpr int BR = AttnShape<BF>::BR, ANT = AttnShape<BF>::NT;
  constexpr bool TWO = (ROLE == ROLE_DQ || ROLE == ROLE_DK);
  constexpr int TILE = BX * HD * ESZ, EXCH = AttnShape<BF>::NW * (TWO ? 2 : 1) * 16 * 256;
  constexpr int LDS = 2 * TILE + (TILE >= EXCH ? 0 : EXCH) + 256;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = attn_kernel<BF, ROLE>;
  VAE_RESERVE_LDS(kern, LDS, "vae_attn");
  dim3 grid((unsigned)(a.T / BR), (unsigned)B);
  hipLaunchKernelGGL(kern, grid, dim3(ANT), LDS, st, a);
  return 0;
}

}  // namespace

extern "C" int vae_attn_supported(int32_t T, int32_t C) { return (C == HD && T >= 64 && T % 64 == 0) ? 1 : 0; }

extern "C" int vae_attn_fwd(const void* q, const void* k, const void* v, int32_t B, int32_t T, int32_t C, float scale, int32_t prec,
                            float* o, float* lse, void* stream) {
  VAE_CHECK(vae_attn_supported(T, C), "vae_attn_fwd: needs C == 512 and T a multiple of 64 (got T=%d C=%d)", T, C);
  VAE_CHECK(q && k && v && o && lse && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), "vae_attn_fwd: null or unaligned pointer");
  VAE_CHECK((size_t)T * HD * 4u < BUF_MAX, "vae_attn_fwd: one image exceeds a buffer descriptor");
  AttnArgs a{};
  a.R1 = q; a.Xs = k; a.Xg = v; a.out = o; a.lse_out = lse; a.T = T; a.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  int rc = (prec == VAE_PREC_BF16) ? launch_role<true, ROLE_FWD>(a, B, st) : launch_role<false, ROLE_FWD>(a, B, st);
  if (rc) return rc;
  VAE_LAUNCH_CHECK("vae_attn_fwd");
  return VAE_OK;
}

extern "C" int vae_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* o32, const float* do32,
                            const float* lse, int32_t B, int32_t T, int32_t C, float scale, int32_t prec,
                            float* dq, float* dk, float* dv, float* dsum, void* stream) {
  VAE_CHECK(vae_attn_supported(T, C), "vae_attn_bwd: needs C == 512 and T a multiple of 64 (got T=%d C=%d)", T, C);
  VAE_CHECK(q && k && v && dout && o32 && do32 && lse && dq && dk && dv && dsum, "vae_attn_bwd: null pointer");
  VAE_CHECK(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(dout) && aligned16(o32) && aligned16(do32) && aligned16(dq) &&
            aligned16(dk) && aligned16(dv), "vae_attn_bwd: unaligned pointer");
  VAE_CHECK((size_t)T * HD * 4u < BUF_MAX, "vae_attn_bwd: one image exceeds a buffer descriptor");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, do32, o32, rows, dsum);
  const bool bf = prec == VAE_PREC_BF16;
  AttnArgs a{};
  a.T = T; a.scale = scale; a.lse = lse; a.dsum = dsum;
  int rc;
  // dQ: R = (Q, dO); tiles: V (dP only), K (scores and the product)
  a.R1 = q; a.R2 = dout; a.Xp = v; a.Xs = k; a.Xg = k; a.out = dq;
  rc = bf ? launch_role<true, ROLE_DQ>(a, B, st) : launch_role<false, ROLE_DQ>(a, B, st);
  if (rc) return rc;
  // dV: R = K; tiles: Q (scores only), dO (product)
  a.R1 = k; a.R2 = nullptr; a.Xs = q; a.Xp = nullptr; a.Xg = dout; a.out = dv;
  rc = bf ? launch_role<true, ROLE_DV>(a, B, st) : launch_role<false, ROLE_DV>(a, B, st);
  if (rc) return rc;
  // dK: R = (K, V); tiles: dO (dP only), Q (scores and the product)
  a.R1 = k; a.R2 = v; a.Xp = dout; a.Xs = q; a.Xg = q; a.out = dk;
  rc = bf ? launch_role<true, ROLE_DK>(a, B, st) : launch_role<false, ROLE_DK>(a, B, st);
  if (rc) return rc;
  VAE_LAUNCH_CHECK("vae_attn_bwd");
  return VAE_OK;
}
