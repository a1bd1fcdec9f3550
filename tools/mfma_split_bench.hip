// Matrix-pipe throughput of an fp32 contraction done (a) on v_mfma_f32_32x32x2_f32 and (b) as three-term bf16 splits on
// v_mfma_f32_32x32x16_bf16 (6 partial products per operand pair: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi).  Registers
// only (no memory traffic): the ceiling such a kernel could reach.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_split_bench.hip -o tools/bin/mfma_split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)  // 16 channels = 8 MFMAs of K = 2, per accumulator block
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + k, b + i, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_split(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a[3], b[4][3];
  for (int t = 0; t < 3; ++t)
    for (int e = 0; e < 8; ++e) {
      a[t][e] = (__bf16)(threadIdx.x * 1e-3f + t + e);
      for (int i = 0; i < 4; ++i) b[i][t][e] = (__bf16)(1.f + i + t * 0.5f + e * 0.25f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 16 channels = 6 MFMAs of K = 16, per accumulator block
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[i][0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[i][1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[i][0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[i][1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[i][2], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[i][0], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 2048 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000, blocks = 1024;  // 4 waves per workgroup, 4 workgroups per CU
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, out, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: 4 blocks x 32x32 outputs x 16 channels x 2 FLOP of fp32-equivalent work
      const double flop = (double)blocks * 4 * iters * 4 * 32 * 32 * 16 * 2;
      if (rep == 1) printf("%s: %.3f ms  %.1f fp32-equivalent TFLOP/s\n", which == 0 ? "fp32 MFMA 32x32x2 (8 per 16 channels)" : "bf16 3-term split, 6 x MFMA 32x32x16", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
