// bf16-compute variant of conv3_tile.hip (3x3 stride-1 convolution forward / forward over a virtual
// nearest-2x upsample / dgrad) for `training.mixed_precision: bf16`:
//   * the activation operand is fp32 (transformed and rounded while staged) or a bf16 image (A16); the output fp32 or bf16
//     (out_bf16, with a bf16 residual); weights are read from `Wh`, a bf16 image of the fp32 master copy with the same
//     layout (vae_pack_bf16, once per step; without it the flat kernel serves the layer);
//   * operands are rounded to bf16 while they are staged into LDS (after the fp32 GroupNorm+SiLU transform);
//   * products run on v_mfma_f32_32x32x16_bf16 (16x the fp32-input MFMA rate), accumulation is fp32.
// Tiling: 4x32-pixel output tile x 128 channels per workgroup; the halo (6x34 pixels x 32 channels) is shared by the
// 9 taps.  At this MFMA rate LDS bandwidth and latency are the scarce resources (a 32x32x16 MFMA consumes 2 KB of
// operands in 32 cycles; the LDS delivers 128 B/cycle to the CU's 4 SIMDs), so
//   * a workgroup is FOUR waves with 64 pixel x 64 channel wave tiles (two image rows x two 32-channel blocks:
//     4 fragments feed 4 MFMAs, 1 KB per MFMA), two workgroups per CU at 2 waves per SIMD (256-VGPR budget);
//   * one barrier covers a whole kernel row: the weight stage holds the 3 taps of one kh (3 x 128 x 32), so a wave
//     issues 24 MFMAs per barrier, and inside that block the fragments of MFMA group i+1 are read while group i runs;
//   * the workgroups are PERSISTENT and run one software pipeline across tiles: the first halo and the first two
//     weight stages of the next output tile are fetched during the last channel chunk of the current one, and the
//     output stores of a tile drain under the next tile's MFMAs.  The GroupNorm scale/shift of the 4 channels a
//     thread stages travel in registers with the halo (no LDS table, no extra barrier);
//   * a step issues its MFMAs first, then waits for the weight stage of the next step (requested a step earlier) and
//     requests the one after; the halo of the next channel chunk is requested two steps before it is needed.
//     Vector-memory results return in order per wave, so the halo request goes BEHIND the weight request of its step
//     (the next step's weight wait then does not wait for the slower HBM halo loads);
// Forward reads both operands as 16-byte k-contiguous fragments; dgrad keeps the weight tile in its memory order
// ([k = co][n = ci]) and reads it with the transposing LDS load (ds_read_b64_tr_b16).
#include "bf16_frag.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BK = 32, TH = 4, TW = 32, HW_ = TW + 2, HP = (TH + 2) * HW_;  // 204 halo pixels
constexpr int LDH = BK + 8;                 // halo row stride in bf16 (80 B: conflict-free ds_read_b128)
constexpr int BN = 128, NT = 256;
constexpr int SH = HP * LDH;                // halo stage (bf16 elements)
constexpr int HQ = HP * (BK / 4);           // float4 slots of one halo (1632)
constexpr int HI = (HQ + NT - 1) / NT;      // 7
constexpr int LDBK = BK + 8;                // weight tile [n][k] row stride (forward)
constexpr int LDBN = BN + 32;               // weight tile [k][n] row stride (dgrad): 320 B => tr reads conflict-free
constexpr int SB1 = BN * LDBK;              // one tap of a weight stage; BN * LDBK == BK * LDBN
static_assert(BN * LDBK == BK * LDBN, "forward and dgrad weight tiles have the same LDS size");
constexpr int SB = 3 * SB1;                 // weight stage: the 3 taps of one kernel row

struct TileId { int b, y0, x0, n0, lin; };

template <bool DG, bool UP, int XF, bool A16>
__global__ __launch_bounds__(NT, 2) void conv3_tile_bf16_kernel(vae_igemm_args p, int tiles_x, int tiles_y, int ntiles) {
  constexpr int LDB = DG ? LDBN : LDBK;
  __shared__ __attribute__((aligned(16))) u16 smem[SH + 2 * SB];
  u16* sH = smem;
  u16* sBst = smem + SH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const vae_conv_geom g = p.g;
  const int tilesN = (p.N + BN - 1) / BN;
  // operand streams through buffer descriptors (common.h); the activation descriptor is made per tile (one image)
  // sub-sampled views and tap subsets (phase convolutions of an upsampler, vaehip.h): pixel (y,x) of A is (y*as+a_oy, x*as+a_ox)
  // of a tensor `as` times larger, likewise C / res with cs; masked kernel rows skip their MFMA step, masked columns their groups
  // (only the instantiation without an input transform carries this: the upsamplers have no GroupNorm in front)
  constexpr bool PHASE = (XF == VAE_XF_NONE) && !A16;
  const int as = (PHASE && p.a_step > 1) ? p.a_step : 1, cs = (PHASE && p.c_step > 1) ? p.c_step : 1;
  const int tmask = (PHASE && p.tapmask) ? p.tapmask : 0x1ff;
  const int rmask = ((tmask & 7) ? 1 : 0) | (((tmask >> 3) & 7) ? 2 : 0) | (((tmask >> 6) & 7) ? 4 : 0);
  const int cmask = (tmask | (tmask >> 3) | (tmask >> 6)) & 7;
  const int kw0 = (cmask & 1) ? 0 : 1, ngrp = 2 * __builtin_popcount(cmask);  // active columns kw0.., 2 k-groups each
  const size_t img_bytes = (size_t)(g.Hs * as) * (g.Ws * as) * g.Cs * 4u;
  const auto rsW = VAE_BUF_RSRC(p.Wh, (size_t)(DG ? p.K * p.sk : p.N * p.sn) * 2u);
  const int Hb = UP ? 2 * g.Hs : g.Hs, Wb = UP ? 2 * g.Ws : g.Ws;
  const int kchunks = (p.K + BK - 1) / BK;
  const int steps = 3 * kchunks;            // one step = one kernel row (3 taps) of one channel chunk

  // persistent schedule: consecutive logical ids (co-tile / x neighbours, which share halo rows) run on the same
  // XCD (hardware places workgroup i on XCD i % 8) and therefore meet in the same L2
  const int G = gridDim.x;
  const int first = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  auto decode = [&](int t) {
    TileId id;
    id.lin = t / tilesN;
    const int tn = t - id.lin * tilesN;
    int r = id.lin;
    const int tx = r % tiles_x; r /= tiles_x;
    const int ty = r % tiles_y;
    id.b = r / tiles_y;
    id.y0 = ty * TH; id.x0 = tx * TW; id.n0 = tn * BN;
    return id;
  };

  f32x16 acc[2][2];  // [row of the wave's row pair][32-channel block]
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- operand staging (register-staged) ----
  constexpr int NW = 3 * BN * BK / 8 / NT;  // 6 uint4 of bf16 weights per thread and step
  // with a bf16 activation image (A16) a slot is 8 channels = one 16-byte load, written to LDS as it is
  constexpr int HQ16 = HP * (BK / 8), HI16 = (HQ16 + NT - 1) / NT;  // 816 slots, 4 per thread
  uint4 rh16[A16 ? HI16 : 1];
  f32x4 rh[HI];                             // halo slots in flight (fp32, as loaded)
  uint2 rhp[HI];                            // the same slots transformed and rounded to bf16, waiting for the barrier
  uint4 rw[NW];                             // weight stage in flight
  f32x4 rsc = {0.f, 0.f, 0.f, 0.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  int hmask = 0;
  // Every staging routine starts from an opaque copy of the thread id: hipcc would otherwise hoist the ~25 per-slot
  // addresses and masks out of the persistent loop and keep them in VGPRs (60+ registers, i.e. spills at the
  // 256-register budget of 2 waves per SIMD); recomputing them costs a few VALU instructions per step.
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto load_halo = [&](const TileId& id, int c0, bool valid) {
    const unsigned Hv = valid ? (unsigned)Hb : 0u;  // an invalid request: every row out of range (no branch on `valid`)
    const int ltid = opaque(tid), hk4 = ltid & (BK / 4 - 1);  // the thread's 4 channels: same for all its slots
    if (A16) {
      const int hk8 = ltid & (BK / 8 - 1);
      const int c8 = c0 + hk8 * 8;
      const auto rsA16 = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.A16) + (int64_t)id.b * g.Hs * g.Ws * g.Cs, img_bytes / 2);
#pragma unroll
      for (int i = 0; i < HI16; ++i) {
        const int q = ltid + NT * i;
        const int pp = q / (BK / 8);
        const int ir = pp / HW_, jc = pp - ir * HW_;
        const int hy = id.y0 - 1 + ir, hx = id.x0 - 1 + jc;
        const bool ok = (q < HQ16) && ((unsigned)hy < Hv) && ((unsigned)hx < (unsigned)Wb) && (c8 < p.K);
        const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
        rh16[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA16, ok ? (unsigned)(((sy * g.Ws + sx) * g.Cs + c8) * 2) : BUF_OOB, 0, 0));
      }
      return;
    }
    hmask = 0;
    const int c = c0 + hk4 * 4;
    const auto rsA = VAE_BUF_RSRC(p.A + (int64_t)id.b * (g.Hs * as) * (g.Ws * as) * g.Cs, img_bytes);
    if (XF != VAE_XF_NONE) {
      const int cs = min(c, p.K - 4);
      rsc = *reinterpret_cast<const f32x4*>(p.scale + (int64_t)id.b * g.Cs + cs);
      rsh = *reinterpret_cast<const f32x4*>(p.shift + (int64_t)id.b * g.Cs + cs);
    }
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = ltid + NT * i;
      const int pp = q / (BK / 4);
      const int ir = pp / HW_, jc = pp - ir * HW_;
      const int hy = id.y0 - 1 + ir, hx = id.x0 - 1 + jc;
      const bool ok = (q < HQ) && ((unsigned)hy < Hv) && ((unsigned)hx < (unsigned)Wb) && (c < p.K);
      const int sy = UP ? (hy >> 1) : hy, sx = UP ? (hx >> 1) : hx;
      rh[i] = VAE_BUF_LOAD4(rsA, ok ? (unsigned)((((sy * as + (PHASE ? p.a_oy : 0)) * (g.Ws * as) + sx * as + (PHASE ? p.a_ox : 0)) * g.Cs + c) * 4) : BUF_OOB);
      hmask |= (ok ? 1 : 0) << i;
    }
  };
  // GroupNorm(+SiLU) and the rounding to bf16 happen in registers BEFORE the barrier that frees the halo stage, one
  // slot per MFMA group of the chunk's last step: the VALU instructions (two quarter-rate transcendentals per
  // element) issue between that group's MFMAs and run while the matrix pipe works; behind the barrier only the
  // ds_writes remain
  auto xform_slot = [&](int i) {
    if (A16) return;  // nothing to transform: the image holds the transformed, rounded operand
    f32x4 v = rh[i];
    if (XF != VAE_XF_NONE) {  // padding must stay zero AFTER the transform
      const bool ok = (hmask >> i) & 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * rsc[e] + rsh[e];
        if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
        v[e] = ok ? u : 0.f;
      }
    }
    rhp[i] = pack4(v);
  };
  auto xform_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HI; ++i) xform_slot(i);
  };
  auto write_halo = [&]() {
    if (A16) {
      const int ltid = opaque(tid), hk8 = ltid & (BK / 8 - 1);
#pragma unroll
      for (int i = 0; i < HI16; ++i) {
        const int q = ltid + NT * i;
        if (q < HQ16) *reinterpret_cast<uint4*>(&sH[(q / (BK / 8)) * LDH + hk8 * 8]) = rh16[i];
      }
      return;
    }
    const int ltid = opaque(tid), hk4 = ltid & (BK / 4 - 1);
#pragma unroll
    for (int i = 0; i < HI; ++i) {
      const int q = ltid + NT * i;
      if (q < HQ)
        *reinterpret_cast<uint2*>(&sH[(q / (BK / 4)) * LDH + hk4 * 4]) = rhp[i];
    }
  };
  // the 3 taps of kernel row kh for one channel chunk: 3 x (128 x 32) bf16 (forward [n][k], dgrad [k][n])
  auto load_w = [&](uint4* dstreg, int n0, int c0, int kh, bool valid) {
    const int ltid = opaque(tid);
    const int Nv = valid ? p.N : 0;  // an invalid request: every column out of range (no branch on `valid`)
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int tap = kh * 3 + (i >> 1);
      const int rem = ltid + NT * (i & 1);
      unsigned off;
      if (!DG) {
        const int n = n0 + (rem >> 2), c = c0 + (rem & 3) * 8;
        off = (n < Nv && c < p.K) ? (unsigned)((n * (int)p.sn + tap * (int)p.st + c) * 2) : BUF_OOB;
      } else {
        const int k = c0 + (rem >> 4), n = n0 + (rem & 15) * 8;
        off = (k < p.K && n < Nv) ? (unsigned)((k * (int)p.sk + tap * (int)p.st + n) * 2) : BUF_OOB;
      }
      dstreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0));
    }
  };
  auto store_w = [&](const uint4* srcreg, u16* sB) {
    const int ltid = opaque(tid);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int rem = ltid + NT * (i & 1);
      u16* dst = sB + (i >> 1) * SB1 + (DG ? (rem >> 4) * LDB + (rem & 15) * 8 : (rem >> 2) * LDB + (rem & 3) * 8);
      *reinterpret_cast<uint4*>(dst) = srcreg[i];
    }
  };

  // lane's address pattern for the transposing read: group row q, column quad pp
  const int trq = (lane & 15) >> 2, trp = lane & 3, trh = (lane >> 4) & 1;
  // one step: 6 MFMA groups (3 taps x 2 k-groups of 16 channels), each 2 pixel-row x 2 channel-block fragments and
  // 4 MFMAs; the fragments of group i+1 are requested before the MFMAs of group i are issued
  auto compute = [&](auto with_xform, int kh, const u16* sB) {
    if (PHASE && !((rmask >> kh) & 1)) {  // uniform: a kernel row outside the tap mask
      if (decltype(with_xform)::value) xform_halo();
      return;
    }
    const int dy = DG ? 2 - kh : kh;
    const u16* aBase = sH + ((2 * wm + dy) * HW_ + lr) * LDH + lh * 8;
    const u16* bBase = DG ? sB + (lh * 8 + trq) * LDB + wn * 64 + trh * 16 + trp * 4 : sB + (wn * 64 + lr) * LDB + lh * 8;
    bf16x8 fa[2][2], fb[2][2];
    auto fetch = [&](int grp, bf16x8* a, bf16x8* b) {
      const int kw = kw0 + (grp >> 1), kg = grp & 1;
      const int dx = DG ? 2 - kw : kw;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = frag_direct(aBase + (mi * HW_ + dx) * LDH + kg * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (!DG) b[ni] = frag_direct(bBase + kw * SB1 + ni * 32 * LDB + kg * 16);
        else b[ni] = frag_tr(bBase + kw * SB1 + kg * 16 * LDB + ni * 32, LDB);
      }
    };
    fetch(0, fa[0], fb[0]);
#pragma unroll
    for (int grp = 0; grp < 6; ++grp) {
      if (grp + 1 < ngrp) fetch(grp + 1, fa[(grp + 1) & 1], fb[(grp + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks each read to just before its MFMA and waits for it there
      if (grp < ngrp) {  // uniform (4 groups when a kernel column is masked)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[grp & 1][mi], fb[grp & 1][ni], acc[mi][ni], 0, 0, 0);
      }
      if (decltype(with_xform)::value) {  // the next chunk's halo, transformed under this group's MFMAs
        static_assert(HI <= 12, "two halo slots per MFMA group at most");
        if (grp < HI) xform_slot(grp);
        if (grp + 6 < HI) xform_slot(grp + 6);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr std::integral_constant<bool, false> plain{};
  constexpr std::integral_constant<bool, true> fused_xform{};

  int t = first;
  if (t >= ntiles) return;  // uniform per workgroup
  TileId cur = decode(t);
  // Every step issues the same loads in the same order (a request that has no target -- nothing follows the last
  // tile -- is made with every lane out of range): hipcc's waitcnt pass then knows exactly how many younger loads may
  // stay in flight at each wait.  With conditional loads it assumes the fewest and waits for the halo at every step.
  load_w(rw, cur.n0, 0, 0, true);
  load_halo(cur, 0, true);
  store_w(rw, sBst);
  xform_halo();
  write_halo();
  load_w(rw, cur.n0, 0, 1, true);
  __syncthreads();
  int par = 0;  // weight buffer holding the current step
  while (true) {
    const int tnext = t + G;
    const bool has_next = tnext < ntiles;
    const TileId nxt = decode(has_next ? tnext : t);
    for (int cch = 0; cch < kchunks; ++cch) {
      const bool last = cch + 1 == kchunks;
      // where the chunk after this one lives: this tile's next chunk, or the first chunk of the next tile (if any)
      const int c0n = last ? 0 : (cch + 1) * BK, n0n = last ? nxt.n0 : cur.n0;
      const bool vn = !last || has_next;
      // Each step issues its MFMAs first and only then waits for the weight stage of the next step (requested one
      // step earlier), so the wait runs under the MFMAs; then it requests the stage after that.  The next chunk's
      // (or next tile's first) halo is requested at the END of kernel row 0, behind that step's weight request.
      compute(plain, 0, sBst + par * SB);
      store_w(rw, sBst + (par ^ 1) * SB);
      load_w(rw, cur.n0, cch * BK, 2, true);
      load_halo(last ? nxt : cur, c0n, vn);
      __syncthreads();
      par ^= 1;
      compute(plain, 1, sBst + par * SB);
      store_w(rw, sBst + (par ^ 1) * SB);
      load_w(rw, n0n, c0n, 0, vn);
      __syncthreads();
      par ^= 1;
      compute(fused_xform, 2, sBst + par * SB);  // + transform of the halo requested two steps ago
      store_w(rw, sBst + (par ^ 1) * SB);
      load_w(rw, n0n, c0n, 1, vn);
      __syncthreads();
      par ^= 1;
      if (!last) {
        write_halo();
        __syncthreads();
      }
    }

    // ---------------- epilogue (fp32); the stores drain under the next tile's main loop ----------------
    // outputs and the residual go through buffer descriptors over this tile's image: a pixel / channel outside the
    // tensor is an out-of-range offset (load reads 0, store is dropped), so the 16 residual loads of a block are issued
    // back to back and there is no branch per element
    const size_t obytes = (size_t)(g.Ho * cs) * (g.Wo * cs) * p.ldc * 4u;
    const auto rsC = VAE_BUF_RSRC(p.C + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
    const auto rsR = VAE_BUF_RSRC((p.res ? p.res : p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, obytes);
    float tsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gs1[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gs2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float gpv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // statistics as shifted sums around the lane's first value
    // the bias values of this lane's channels, loaded once per tile before the first store (per block each load sat behind the
    // previous block's stores, and its wait -- vmcnt counts stores too -- waited for their acknowledgements: conv3_wide_bf16.hip)
    float pbias[2][2];  // [channel block][channel of the lane pair / the lane's own channel]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = cur.n0 + wn * 64 + ni * 32 + lr;
      const bool cok = p.bias && col < p.N;
      if (p.out_bf16) {  // uniform
        pbias[ni][0] = cok ? p.bias[col & ~1] : 0.f;
        pbias[ni][1] = cok ? p.bias[col | 1] : 0.f;
      } else {
        pbias[ni][0] = pbias[ni][1] = cok ? p.bias[col] : 0.f;
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int oy = cur.y0 + 2 * wm + mi;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = cur.n0 + wn * 64 + ni * 32 + lr;
        const bool colok = col < p.N && oy < g.Ho;
        if (p.out_bf16) {  // uniform: bf16 output.  Adjacent lanes hold adjacent channels of the same pixels: they swap every other
          // register, so each lane ends up with two channels of 8 pixels (4-byte stores, 4-byte loads of the bf16 residual); the
          // statistics describe the ROUNDED values (see conv3_wide_bf16.hip); no tracker with it
          const size_t ob16 = (size_t)(g.Ho * cs) * (g.Wo * cs) * p.ldc * 2u;
          const auto rsC16 = VAE_BUF_RSRC(reinterpret_cast<u16*>(p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, ob16);
          const auto rsR16 = VAE_BUF_RSRC(reinterpret_cast<const u16*>(p.res ? p.res : p.C) + (int64_t)cur.b * (g.Ho * cs) * (g.Wo * cs) * p.ldc, ob16);
          const bool odd = lr & 1;
          const float b0 = pbias[ni][0], b1 = pbias[ni][1];  // (rows beyond the image: the stores are out of range)
          unsigned o16[8], rr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int e = 2 * j + (odd ? 1 : 0);
            const int ox = cur.x0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            o16[j] = (colok && ox < g.Wo) ? (unsigned)((((oy * cs + (PHASE ? p.c_oy : 0)) * (g.Wo * cs) + ox * cs + (PHASE ? p.c_ox : 0)) * p.ldc + (col & ~1)) * 2) : BUF_OOB;
            rr[j] = 0u;
          }
          if (p.res) {  // uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) rr[j] = __builtin_amdgcn_raw_buffer_load_b32(rsR16, o16[j], 0, 0);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a0 = p.alpha * acc[mi][ni][2 * j], a1 = p.alpha * acc[mi][ni][2 * j + 1];
            const float recv = lane_xor1(odd ? a0 : a1);
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            bf16x2_t h;
            h[0] = (__bf16)((odd ? recv : a0) + b0 + __builtin_bit_cast(float, rr[j] << 16));
            h[1] = (__bf16)((odd ? a1 : recv) + b1 + __builtin_bit_cast(float, rr[j] & 0xffff0000u));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rsC16, o16[j], 0, 0);
            const float q0 = (float)h[0], q1 = (float)h[1];
            if (j == 0) gpv[mi][ni] = q0;
            const float d0 = q0 - gpv[mi][ni], d1 = q1 - gpv[mi][ni];
            gs1[mi][ni] += d0 + d1;
            gs2[mi][ni] += d0 * d0 + d1 * d1;
            acc[mi][ni][2 * j] = 0.f;
            acc[mi][ni][2 * j + 1] = 0.f;
          }
          continue;
        }
        const float bv = pbias[ni][0];
        unsigned off[16];
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ox = cur.x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          off[r] = (colok && ox < g.Wo) ? (unsigned)((((oy * cs + (PHASE ? p.c_oy : 0)) * (g.Wo * cs) + ox * cs + (PHASE ? p.c_ox : 0)) * p.ldc + col) * 4) : BUF_OOB;
          rv[r] = 0.f;
        }
        if (p.res) {  // uniform
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = p.alpha * acc[mi][ni][r] + bv + rv[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, off[r], 0, 0);
          tsum[mi][ni] += (off[r] != BUF_OOB) ? fabsf(v) : 0.f;
          if (r == 0) gpv[mi][ni] = v;
          const float dv = v - gpv[mi][ni];  // (the statistics epilogue only runs on full tiles)
          gs1[mi][ni] += dv;
          gs2[mi][ni] += dv * dv;
          acc[mi][ni][r] = 0.f;
        }
      }
    }
    if (p.gstat) {  // uniform: GroupNorm moments of this tile's outputs, layout of vae_gn_stats_partial with one chunk per wave-row
      // band of the tile (2 rows x 32 pixels): the lane's two rows are merged in place, the group's lanes by DPP moves, and the
      // first lane writes -- no LDS round trip, no barrier (see conv3_wide_bf16.hip)
      const int cpg = p.N / p.gstat_groups;  // channels per group (4, 8 or 16)
      const int tile_in_img = cur.lin - cur.b * (tiles_x * tiles_y);
      float* gbase = p.gstat + (((int64_t)cur.b * (tiles_x * tiles_y) + tile_in_img) * 2 + wm) * p.gstat_groups * 2;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const MeanM2 rows = mm2_merge_equal(mm2_from_shifted(gpv[0][ni], gs1[0][ni], gs2[0][ni], 16.f),
                                            mm2_from_shifted(gpv[1][ni], gs1[1][ni], gs2[1][ni], 16.f), 16.f);
        const MeanM2 a = mm2_wave_group(rows, cpg, 32.f);
        if (lh == 0 && (lr & (cpg - 1)) == 0) {
          float* o = gbase + ((cur.n0 + wn * 64 + ni * 32 + lr) / cpg) * 2;
          o[0] = a.m;
          o[1] = a.M2;
        }
      }
    }
    if (p.track) {  // uniform; the last loop barrier separated the halo reads from this reuse of its space
      float* red = reinterpret_cast<float*>(smem);  // [4 rows][BN] fp32 = 2 KB of the 16 KB halo stage
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float s2 = tsum[mi][ni] + __shfl_xor(tsum[mi][ni], 32, 64);
          if (lh == 0) red[(2 * wm + mi) * BN + wn * 64 + ni * 32 + lr] = s2;
        }
      __syncthreads();
      if (tid < BN && cur.n0 + tid < p.N)
        p.track[(int64_t)cur.lin * p.N + cur.n0 + tid] = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
      __syncthreads();
    }
    if (!has_next) break;
    write_halo();  // next tile's first channel chunk (requested during the last chunk's first kernel row, transformed since)
    __syncthreads();
    t = tnext;
    cur = nxt;
  }
}

template <bool DG, bool UP>
void launch_xf(const vae_igemm_args& a, dim3 grid, int tx, int ty, int nt, hipStream_t st) {
  if (a.A16 != nullptr) {  // transformed bf16 activation image (xf == NONE checked by the caller)
    hipLaunchKernelGGL((conv3_tile_bf16_kernel<DG, UP, VAE_XF_NONE, true>), grid, dim3(NT), 0, st, a, tx, ty, nt);
    return;
  }
  switch (a.xf) {
    case VAE_XF_NONE: hipLaunchKernelGGL((conv3_tile_bf16_kernel<DG, UP, VAE_XF_NONE, false>), grid, dim3(NT), 0, st, a, tx, ty, nt); break;
    case VAE_XF_AFFINE: hipLaunchKernelGGL((conv3_tile_bf16_kernel<DG, UP, VAE_XF_AFFINE, false>), grid, dim3(NT), 0, st, a, tx, ty, nt); break;
    default: hipLaunchKernelGGL((conv3_tile_bf16_kernel<DG, UP, VAE_XF_AFFINE_SILU, false>), grid, dim3(NT), 0, st, a, tx, ty, nt); break;
  }
}

}  // namespace

// chunks per image of the statistics epilogue: one per 2-row band of a 4 x 32-pixel tile (0 = not available for these arguments)
int conv3_tile_bf16_gstat_chunks(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  if (a.gstat_groups <= 0 || a.N % BN != 0 || a.N % a.gstat_groups != 0 || g.mode == VAE_MODE_DGRAD || a.c_step > 1) return 0;
  const int cpg = a.N / a.gstat_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16) return 0;
  return (g.Wo / TW) * (g.Ho / 2);
}

// the kernel reads the weights from their bf16 image in 16-byte (8-element) pieces: they must be aligned and never
// straddle a row end
bool conv3_tile_bf16_packed(const vae_igemm_args& a) {
  const vae_conv_geom& g = a.g;
  const size_t as = a.a_step > 1 ? a.a_step : 1, cs = a.c_step > 1 ? a.c_step : 1;
  const bool fits32 = (size_t)g.Hs * g.Ws * g.Cs * 4u * as * as < BUF_MAX && (size_t)g.Ho * g.Wo * a.ldc * 4u * cs * cs < BUF_MAX &&  // one image per descriptor
                      (size_t)std::max((int64_t)a.K * a.sk, (int64_t)a.N * a.sn) * 2u < BUF_MAX;
  return a.Wh != nullptr && aligned16(a.Wh) && aligned16(a.A) && fits32 && a.K % 8 == 0 && a.N % 8 == 0 && a.st % 8 == 0 &&
         (a.sn == 1 || a.sn % 8 == 0) && (a.sk == 1 || a.sk % 8 == 0);
}

int launch_conv3_tile_bf16(const vae_igemm_args& a, bool bkm, hipStream_t st) {
  const vae_conv_geom& g = a.g;
  if (!conv3_tile_bf16_packed(a)) return VAE_EINVAL;
  const int tx = g.Wo / TW, ty = g.Ho / TH;
  const int64_t nt = (int64_t)((a.N + BN - 1) / BN) * tx * ty * g.B;
  if (nt > 0x7fffffffLL) return VAE_EINVAL;
  dim3 grid((unsigned)std::min<int64_t>(nt, 512));  // persistent: two 4-wave workgroups per CU (LDS 76 KB each)
  if (g.mode == VAE_MODE_DGRAD) launch_xf<true, false>(a, grid, tx, ty, (int)nt, st);
  else if (g.mode == VAE_MODE_UP2X) launch_xf<false, true>(a, grid, tx, ty, (int)nt, st);
  else launch_xf<false, false>(a, grid, tx, ty, (int)nt, st);
  return 0;
}
