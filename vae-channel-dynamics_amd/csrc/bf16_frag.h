// bf16 operand helpers shared by the bf16-compute kernels (gfx950).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// 4 fp32 -> 4 bf16 (round to nearest even, v_cvt_pk_bf16_f32), packed for one ds_write_b64
__device__ __forceinline__ uint2 pack4(f32x4 v) {
  bf16x4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
  return __builtin_bit_cast(uint2, h);
}
// 4 bf16 (as loaded: 8 bytes) -> 4 fp32
__device__ __forceinline__ f32x4 unpack4(uint2 r) {
  f32x4 v;
  v[0] = __builtin_bit_cast(float, r.x << 16);
  v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, r.y << 16);
  v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
  return v;
}
// 4 consecutive elements of an operand stored as fp32 or bf16, through a buffer descriptor whose range is in BYTES of that
// storage.  ONE instruction for both storages -- a 16-byte load at element offset `eoff` (a bf16 operand's load also brings the
// next 4 elements along, unused) -- and the conversion happens where the value is CONSUMED (raw4_to_f32 at the LDS write a
// pipeline step later).  A storage-dependent load instruction plus an immediate conversion had made hipcc wait for every load
// right behind it (the flat kernels lost 20 % when bf16 storage came in).  eoff < 0 = out of range (reads zeros).
__device__ __forceinline__ uint4 buf_load4_raw(__amdgpu_buffer_rsrc_t rs, unsigned esize, int eoff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, eoff >= 0 ? (unsigned)eoff * esize : BUF_OOB, 0, 0));
}
__device__ __forceinline__ f32x4 raw4_to_f32(uint4 r, bool bf) {
  if (bf) return unpack4(uint2{r.x, r.y});
  return __builtin_bit_cast(f32x4, r);
}
// one element through a descriptor (fp32: 4-byte, bf16: 2-byte access); eoff < 0 = out of range
__device__ __forceinline__ float buf_load1_elem(__amdgpu_buffer_rsrc_t rs, bool bf, int eoff) {
  if (bf) return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, eoff >= 0 ? (unsigned)eoff * 2u : BUF_OOB, 0, 0) << 16);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, eoff >= 0 ? (unsigned)eoff * 4u : BUF_OOB, 0, 0));
}
__device__ __forceinline__ void buf_store1_elem(__amdgpu_buffer_rsrc_t rs, bool bf, int eoff, float v) {
  if (bf) {
    const __bf16 h = (__bf16)v;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rs, eoff >= 0 ? (unsigned)eoff * 2u : BUF_OOB, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, eoff >= 0 ? (unsigned)eoff * 4u : BUF_OOB, 0, 0);
  }
}
// MFMA operand from a k-contiguous image: 8 consecutive k at p (one ds_read_b128)
__device__ __forceinline__ bf16x8 frag_direct(const u16* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
// MFMA operand from a [k][col] image (k = the contraction index runs down the rows): 8 consecutive k for this
// lane's column through two transposing reads (ds_read_b64_tr_b16).  Within a 16-lane group, lane 4q+pp supplies
// the address of row q, columns 4pp..4pp+3 of a 4x16 block and lane i receives column i (cdna_hip_programming.md
// T10).  `p` = the address this lane supplies for rows k0..k0+3; rows k0+4..k0+7 are `4*ld` elements further.
// Row strides of 64 B mod 256 B (e.g. 160 or 96 bf16) put the four rows of a block in different bank ranges.
__device__ __forceinline__ bf16x8 frag_tr(const u16* p, int ld) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * ld));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
