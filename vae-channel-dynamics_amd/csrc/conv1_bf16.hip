// 1x1 convolutions (the resnet shortcuts) and their dgrads in bf16 mode as a STREAMING GEMM with the weights resident in LDS:
//     out[m][n] = sum_k A[m][k] * W(n, k) (+ bias[n])        A, out: bf16 tensors [M][K], [M][N]; W: the bf16 image of the weights
// M = B*H*W runs to 2M rows while K, N are 128 .. 512: 0.3 ms of HBM traffic per launch.  The flat kernel (igemm_bf16.hip) cuts
// such a product into 128 x 128 tiles of 2 .. 8 k-steps: workgroups that live for two memory round trips, 0.5 - 0.7 ms
// (190 - 330 TFLOP/s).  Here
//   * a workgroup (4 waves) keeps a 128-column slice of W in LDS for its whole life (k-contiguous rows for the forward, [k][n]
//     rows read through the transposing load for the dgrad) and walks over the rows: after the one barrier behind the weight
//     staging the waves never synchronise again;
//   * a wave owns blocks of 32 rows: its A operand comes from global memory STRAIGHT into MFMA fragments (lane (row, k-half)
//     reads 16 contiguous bytes per 16-channel group: the rows are k-contiguous as stored) -- the whole K extent of the NEXT
//     block is requested before the current one is multiplied, so 8 .. 32 KB per wave are in flight all the time;
//   * output as bf16 pairs through a lane swap (conv3_wide_bf16.hip), bias loaded once per workgroup.
// Served: bf16 arithmetic, both tensors stored as bf16 (the activation-storage mode), taps = 1, stride 1, no transform / residual
// / statistics, K in {128, 256, 512}, N % 128 == 0, a weight slice that fits 150 KB of LDS; everything else stays on the flat kernel.
#include "bf16_frag.h"
#include <algorithm>

namespace {

constexpr int C1_NT = 256, C1_BN = 128;
// K = 16 * KG <= 128: compiled for two workgroups per CU (<= 256 registers per wave); the launch computes its grid from the same predicate
#define C1_TWO_PER_CU(KG) ((KG) <= 8)

template <bool DG, int KG>  // KG = K / 16
__global__ __launch_bounds__(C1_NT, (C1_TWO_PER_CU(KG) ? 2 : 1)) void conv1_bf16_kernel(vae_igemm_args p, int nblocks, int wpn) {
  constexpr int K = KG * 16;
  constexpr int LDB = DG ? (C1_BN + 32) : (K + 8);  // dgrad rows: 320 B (64 B mod 256 B: conflict-free transposing reads); forward: K + 8
  extern __shared__ __attribute__((aligned(16))) u16 sB[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  const int slice = blockIdx.x / wpn, wgi = blockIdx.x - slice * wpn;
  const int n0 = slice * C1_BN;

  // ---- the weight slice: 16-byte pieces of the bf16 image, once ----
  {
    const u16* __restrict__ Wh = reinterpret_cast<const u16*>(p.Wh);
    if (!DG) {  // W[n][k], k contiguous: rows n0 .. n0 + 127
      constexpr int PPR = K / 8;  // pieces per row
      for (int i = tid; i < C1_BN * PPR; i += C1_NT) {
        const int n = i / PPR, c = (i - n * PPR) * 8;
        *reinterpret_cast<uint4*>(&sB[n * LDB + c]) = *reinterpret_cast<const uint4*>(&Wh[(int64_t)(n0 + n) * p.sn + c]);
      }
    } else {    // W[k][n], n contiguous: columns n0 .. n0 + 127 of every row
      constexpr int PPR = C1_BN / 8;
      for (int i = tid; i < K * PPR; i += C1_NT) {
        const int k = i / PPR, c = (i - k * PPR) * 8;
        *reinterpret_cast<uint4*>(&sB[k * LDB + c]) = *reinterpret_cast<const uint4*>(&Wh[(int64_t)k * p.sk + n0 + c]);
      }
    }
  }
  // bias of this lane's channel pair, per 32-channel block
  const bool odd = lr & 1;
  float pb0[4], pb1[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int col = n0 + nb * 32 + lr;
    pb0[nb] = p.bias ? p.bias[col & ~1] : 0.f;
    pb1[nb] = p.bias ? p.bias[col | 1] : 0.f;
  }
  __syncthreads();

  const size_t abytes = (size_t)p.M * K * 2u, cbytes = (size_t)p.M * p.ldc * 2u;
  const auto rsA = VAE_BUF_RSRC(p.A, abytes);  // (bf16 storage: p.A is the bf16 tensor, vae_igemm_args.a_bf16)
  const auto rsC = VAE_BUF_RSRC(p.C, cbytes);
  const unsigned aoff = (unsigned)((lr * K + lh * 8) * 2);  // this lane's bytes inside a 32-row block, k-group 0
  const int boff = DG ? (lh * 8 + trq) * LDB + trh * 16 + trp * 4 : lr * LDB + lh * 8;

  auto load_a = [&](int blk, uint4 (&a)[KG]) {  // the whole K extent of a 32-row block (beyond the last block: out of range, zeros)
    const unsigned base = blk < nblocks ? (unsigned)blk * (unsigned)(32 * K * 2) + aoff : BUF_OOB;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
      a[kg] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, blk < nblocks ? base + kg * 32 : BUF_OOB, 0, 0));
  };
  auto run_block = [&](int blk, const uint4 (&a)[KG]) {
    // (the offset is made opaque per block: the weights in LDS never change after the barrier, so hipcc would otherwise hoist
    // all 4 KG fragment reads out of the loop over blocks and try to keep the whole slice in registers)
    int bo = boff;
    asm volatile("" : "+v"(bo));
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const bf16x8 af = __builtin_bit_cast(bf16x8, a[kg]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const bf16x8 bf = DG ? frag_tr(sB + bo + kg * 16 * LDB + nb * 32, LDB) : frag_direct(sB + bo + nb * 32 * LDB + kg * 16);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[nb], 0, 0, 0);
      }
    }
    // bf16 pairs: adjacent lanes hold adjacent channels of the same 16 rows and swap every other register
    const int m0 = blk * 32;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int col = n0 + nb * 32 + (lr & ~1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 2 * j + (odd ? 1 : 0);
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float a0 = acc[nb][2 * j], a1 = acc[nb][2 * j + 1];
        const float recv = lane_xor1(odd ? a0 : a1);
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        bf16x2_t h;
        h[0] = (__bf16)((odd ? recv : a0) + pb0[nb]);
        h[1] = (__bf16)((odd ? a1 : recv) + pb1[nb]);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rsC, row < p.M ? (unsigned)(row * p.ldc + col) * 2u : BUF_OOB, 0, 0);
      }
    }
  };

  // blocks wgi * 4 + wave, + stride, ...: two register sets for A, the next block in flight while this one is multiplied
  const int stride = wpn * 4;
  uint4 a0[KG], a1[KG];
  int blk = wgi * 4 + wave;
  load_a(blk, a0);
  while (blk < nblocks) {
    load_a(blk + stride, a1);
    run_block(blk, a0);
    blk += stride;
    if (blk >= nblocks) break;
    load_a(blk + stride, a0);
    run_block(blk, a1);
    blk += stride;
  }
}

size_t conv1_lds_bytes(bool dg, int K) { return dg ? (size_t)K * (C1_BN + 32) * 2u : (size_t)C1_BN * (K + 8) * 2u; }

}  // namespace

// after rows_canon: the operand is a bf16 TENSOR (a_bf16), the output a bf16 tensor, the weights come from their bf16 image
bool conv1_bf16_eligible(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.prec != VAE_PREC_BF16 || a.Wh == nullptr || a.A16 != nullptr || !a.a_bf16 || !a.out_bf16 || a.xf != VAE_XF_NONE) return false;
  if (g.taps != 1 || g.stride != 1 || a.batch != 1 || a.alpha != 1.0f || a.res || a.track || a.gstat || a.gnb_ws) return false;
  if (a.tapmask != 0 || a.a_step > 1 || a.c_step > 1) return false;
  if (!(g.mode == VAE_MODE_FWD || g.mode == VAE_MODE_DGRAD) || g.Ho != g.Hs || g.Wo != g.Ws) return false;
  if (!(a.K == 128 || a.K == 256 || a.K == 512) || a.N % C1_BN != 0 || a.ldc != a.N || g.Cs != a.K || a.M % 32 != 0 || a.M < 32768) return false;
  const bool dg = g.mode == VAE_MODE_DGRAD;
  if (!dg && !(a.sk == 1 && a.sn == a.K)) return false;
  if (dg && !(a.sn == 1 && a.sk == a.N)) return false;
  if (conv1_lds_bytes(dg, a.K) > 150 * 1024) return false;
  if (!aligned16(a.A) || !aligned16(a.C) || !aligned16(a.Wh)) return false;
  if ((size_t)a.M * a.K * 2u >= BUF_MAX || (size_t)a.M * a.ldc * 2u >= BUF_MAX) return false;
  return true;
}

template <bool DG, int KG>
static int launch_conv1_t(const vae_igemm_args& a, hipStream_t st) {
  auto kern = conv1_bf16_kernel<DG, KG>;
  const size_t lds = conv1_lds_bytes(DG, KG * 16);
  VAE_RESERVE_LDS(kern, lds, "conv1_bf16");
  const int slices = a.N / C1_BN, nblocks = a.M / 32;
  const int per_cu = (KG <= 16 && 2 * lds <= 150 * 1024) ? 2 : 1;  // (K <= 256: < 256 registers per wave, two workgroups fit a CU where the weight slices do)
  const int wpn = std::max(1, std::min((256 * per_cu) / slices, (nblocks + 3) / 4));
  hipLaunchKernelGGL(kern, dim3((unsigned)(slices * wpn)), dim3(C1_NT), lds, st, a, nblocks, wpn);
  return 0;
}

int launch_conv1_bf16(const vae_igemm_args& a, hipStream_t st) {
  const bool dg = a.g.mode == VAE_MODE_DGRAD;
  switch (a.K) {
    case 128: return dg ? launch_conv1_t<true, 8>(a, st) : launch_conv1_t<false, 8>(a, st);
    case 256: return dg ? launch_conv1_t<true, 16>(a, st) : launch_conv1_t<false, 16>(a, st);
    default: return dg ? launch_conv1_t<true, 32>(a, st) : launch_conv1_t<false, 32>(a, st);
  }
}
