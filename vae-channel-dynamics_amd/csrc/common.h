// Shared device helpers for libvaehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "common_host.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VAE_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      vae_set_error(__VA_ARGS__);            \
      return VAE_EINVAL;                     \
    }                                        \
  } while (0)

#define VAE_LAUNCH_CHECK(name)                                                 \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      vae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
      return VAE_ELAUNCH;                                                      \
    }                                                                          \
  } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float silu_f(float u) {
  // u * sigmoid(u); v_exp_f32 + v_rcp_f32 (1 ulp), ~1e-7 relative; __frcp_rn would expand to a full IEEE division
  return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
}
__device__ __forceinline__ float silu_grad_f(float u) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// row (b,y,x) + tap (kh,kw) -> source pixel; returns validity (zero padding otherwise)
__device__ __forceinline__ bool src_pixel(const vae_conv_geom& g, int y, int x, int kh, int kw,
                                          int& sy, int& sx) {
  if (g.mode == VAE_MODE_FWD) {
    sy = y * g.stride + kh - g.pad_t;
    sx = x * g.stride + kw - g.pad_l;
    return ((unsigned)sy < (unsigned)g.Hs) && ((unsigned)sx < (unsigned)g.Ws);
  } else if (g.mode == VAE_MODE_UP2X) {
    int uy = y + kh - 1, ux = x + kw - 1;
    bool ok = ((unsigned)uy < (unsigned)(2 * g.Hs)) && ((unsigned)ux < (unsigned)(2 * g.Ws));
    sy = uy >> 1;
    sx = ux >> 1;
    return ok;
  } else if (g.mode == VAE_MODE_DGRAD_S2) {  // (y - kh, x - kw) are even by construction of the class tap list
    sy = (y - kh) >> 1;
    sx = (x - kw) >> 1;
    return ((unsigned)sy < (unsigned)g.Hs) && ((unsigned)sx < (unsigned)g.Ws);
  } else {
    int ty = y + g.pad_t - kh, tx = x + g.pad_l - kw;
    if (ty < 0 || tx < 0) return false;
    if (g.stride == 2) {
      if ((ty | tx) & 1) return false;
      ty >>= 1;
      tx >>= 1;
    }
    sy = ty;
    sx = tx;
    return (sy < g.Hs) && (sx < g.Ws);
  }
}

// The same mapping without a branch on g.mode: the four geometries as one affine form t = y a + kh b + c with a range check,
// a parity mask and a shift, from wave-uniform constants built once per kernel.  (The flat kernels call the mapping for every
// operand load of every step; with the mode switch inside, each call compiled into ~60 scalar instructions and ten branches:
// address generation, not the matrix pipe, bounded the bf16 flat kernels -- 135-300 TFLOP/s.)
struct SrcMap {
  int a, b, cy, cx, pm, sh, Hs, Ws;
  unsigned limy, limx;
};
__device__ __forceinline__ SrcMap make_srcmap(const vae_conv_geom& g) {
  // (selects, not an if / else-if chain over g.mode: hipcc 7.2 compiled the chain into a structurised flow whose DGRAD arm
  // came out with the UP2X constants -- tools/srcmap_check.hip compares both mappings on the device)
  const bool fwd = g.mode == VAE_MODE_FWD, up = g.mode == VAE_MODE_UP2X, s2 = g.mode == VAE_MODE_DGRAD_S2;
  const bool dg = !(fwd || up || s2);
  SrcMap m;
  m.Hs = g.Hs; m.Ws = g.Ws;
  m.a = fwd ? g.stride : 1;
  m.b = (fwd || up) ? 1 : -1;
  m.cy = fwd ? -g.pad_t : (up ? -1 : (s2 ? 0 : g.pad_t));
  m.cx = fwd ? -g.pad_l : (up ? -1 : (s2 ? 0 : g.pad_l));
  m.pm = (dg && g.stride == 2) ? 1 : 0;
  m.sh = (up || s2) ? 1 : m.pm;
  m.limy = up ? 2u * (unsigned)g.Hs : 0x7fffffffu;  // (0x7fffffff: rejects negative t)
  m.limx = up ? 2u * (unsigned)g.Ws : 0x7fffffffu;
  return m;
}
__device__ __forceinline__ bool src_pixel(const SrcMap& m, int y, int x, int kh, int kw, int& sy, int& sx) {
  const int ty = y * m.a + kh * m.b + m.cy, tx = x * m.a + kw * m.b + m.cx;
  sy = ty >> m.sh;
  sx = tx >> m.sh;
  return ((unsigned)ty < m.limy) & ((unsigned)tx < m.limx) & (((ty | tx) & m.pm) == 0) & ((unsigned)sy < (unsigned)m.Hs) & ((unsigned)sx < (unsigned)m.Ws);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics as (mean, M2 = sum of squared deviations) per chunk, merged with Chan's formula.  Sums of x and x^2
// cancel catastrophically once |mean| >> std (var = E[x^2] - mean^2 loses |mean/std|^2 * 1e-7 of relative accuracy in
// fp32); torch's CPU group_norm the reference runs on is a Welford pass.  Every producer (the streaming pass and the conv
// epilogues) therefore leaves centred moments, and the final pass merges them in fp64.
// ---------------------------------------------------------------------------------------
struct MeanM2 { float m, M2; };
// two sets of n_each elements each
__device__ __forceinline__ MeanM2 mm2_merge_equal(MeanM2 a, MeanM2 b, float n_each) {
  const float d = b.m - a.m;
  return MeanM2{a.m + 0.5f * d, a.M2 + b.M2 + d * d * (0.5f * n_each)};
}
// set a of na elements with set b of nb elements (na + nb > 0)
__device__ __forceinline__ MeanM2 mm2_merge(MeanM2 a, float na, MeanM2 b, float nb) {
  const float n = na + nb, d = b.m - a.m;
  return MeanM2{a.m + d * (nb / n), a.M2 + b.M2 + d * d * (na * nb / n)};
}
// a lane's shifted sums (pivot pv, s1 = sum(v - pv), s2 = sum((v - pv)^2) over n values) -> centred moments
__device__ __forceinline__ MeanM2 mm2_from_shifted(float pv, float s1, float s2, float n) {
  const float dm = s1 / n;
  return MeanM2{pv + dm, fmaxf(s2 - s1 * dm, 0.f)};
}
// Cross-lane moves that stay in the VALU.  A __shfl_xor compiles to ds_bpermute_b32 -- the LDS pipe plus an s_waitcnt per use;
// in the bf16 epilogues (64 swaps per wave tile + the statistics merges) that was 8 of 12 us per tile (tools/wide_timing.py).
// lane_xor1: the value of lane i ^ 1 (DPP quad_perm [1,0,3,2]); lane_plus<O>: of lane i + O inside its row of 16 lanes (DPP
// row_shl; lanes past the row read 0); lane_xor32: of lane i ^ 32 (v_permlane32_swap, gfx950).
__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <int O>
__device__ __forceinline__ float lane_plus(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x100 + O, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0]: lanes 32.. got lanes 0..31; r[1]: lanes 0..31 got lanes 32..
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
// merge over the `cpg` adjacent lanes of a group (cpg = 1, 2, 4, 8 or 16: groups are aligned inside a 16-lane row) and the two
// lane halves (each lane starts with n0 elements).  Lane i merges lane i + o for o = 1, 2, .. cpg/2: afterwards the FIRST lane
// of a group holds the group's moments (the other lanes partial ones); callers read lanes with lh == 0 && lr % cpg == 0 only.
__device__ __forceinline__ MeanM2 mm2_wave_group(MeanM2 a, int cpg, float n0) {
  float n = n0;
  if (cpg > 1) {  // (uniform)
    a = mm2_merge_equal(a, MeanM2{lane_plus<1>(a.m), lane_plus<1>(a.M2)}, n);
    n *= 2.f;
  }
  if (cpg > 2) {
    a = mm2_merge_equal(a, MeanM2{lane_plus<2>(a.m), lane_plus<2>(a.M2)}, n);
    n *= 2.f;
  }
  if (cpg > 4) {
    a = mm2_merge_equal(a, MeanM2{lane_plus<4>(a.m), lane_plus<4>(a.M2)}, n);
    n *= 2.f;
  }
  if (cpg > 8) {
    a = mm2_merge_equal(a, MeanM2{lane_plus<8>(a.m), lane_plus<8>(a.M2)}, n);
    n *= 2.f;
  }
  return mm2_merge_equal(a, MeanM2{lane_xor32(a.m), lane_xor32(a.M2)}, n);
}

// ---------------------------------------------------------------------------------------
// operand staging helpers shared by the contraction kernels
// ---------------------------------------------------------------------------------------
constexpr int SS_HALF = 512;  // floats of GroupNorm scale (and of shift) kept in LDS per workgroup

// Unvectorised guarded load (channel counts that are not a multiple of 4, unaligned rows): element by element.  The
// vectorised paths use the buffer-descriptor loads below instead.
__device__ __forceinline__ f32x4 load4s(const float* p, bool ok, int c, int C) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) v[e] = p[e];
  }
  return v;
}

// Buffer-descriptor loads.  The descriptor covers [base, base + bytes) with a 32-bit per-lane byte offset; an offset
// at or beyond `bytes` (use BUF_OOB) reads zeros.  Padding pixels, channel tails and rows beyond the matrix then need
// neither a branch nor a select on the loaded value -- a select makes hipcc wait for the load right behind it, instead
// of at the LDS write a pipeline step later.  Build the descriptor from wave-uniform values only.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
constexpr size_t BUF_MAX = 0xFFFFFFF0u;  // bytes one descriptor can cover
#define VAE_BUF_RSRC(ptr, bytes) \
  __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(ptr)), 0, (unsigned)(bytes), 0x00020000)
#define VAE_BUF_LOAD4(rsrc, off) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (off), 0, 0))
// `ok ? <address arithmetic> : BUF_OOB` as the offset of a load: hipcc turns the select into a branch around the arithmetic
// and duplicates the LOAD into both arms; the arm that computes the address then waits (vmcnt(0)) for the other arm's
// destination registers, so every load of a pipeline step sits behind the previous one's round trip (the flat kernels' loops,
// round 3).  With the computed offset made opaque first the select stays one v_cndmask and the loads go out back to back.
__device__ __forceinline__ unsigned oob_unless(bool ok, unsigned off) {
  asm volatile("" : "+v"(off));
  return ok ? off : BUF_OOB;
}
__device__ __forceinline__ int neg_unless(bool ok, int idx) {  // the same for element indices (-1 = do not request)
  asm volatile("" : "+v"(idx));
  return ok ? idx : -1;
}

// LDS-table variant: no bounds branches (table entries beyond the valid columns are zero-filled)
template <int XF>
__device__ __forceinline__ f32x4 xform4_tab(f32x4 v, const float* scale, const float* shift, bool ok) {
  f32x4 sc = *reinterpret_cast<const f32x4*>(scale);
  f32x4 sh = *reinterpret_cast<const f32x4*>(shift);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float u = v[e] * sc[e] + sh[e];
    if (XF == VAE_XF_AFFINE_SILU) u = silu_f(u);
    v[e] = ok ? u : 0.f;
  }
  return v;
}

