// Which MFMA shape does the chip hold the higher clock on?  Bare MFMA loops on RANDOM operands kept in registers, one 64x64
// output tile per wave in both shapes, 4 waves per workgroup, NWG workgroups per CU: fp32 32x32x2 vs 16x16x4, bf16 32x32x16
// vs 16x16x32.  Reports wall time, TFLOP/s and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, the
// guide's DVFS check).  Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_shape_bench.hip -o tools/bin/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NSET 2  // operand sets rotated through the loop (more sets: hipcc parks operands in AGPRs and moves them every step)

struct Stamp { unsigned long long c0, c1, r0, r1; };

__device__ inline void stamp_begin(Stamp& s) { s.c0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime(); }
__device__ inline void stamp_end(Stamp& s, Stamp* out) {
  s.c1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// fp32, 32x32x2: 2x2 blocks of 32x32; per k-step (K = 2) 2 A + 2 B registers, 4 MFMAs
__global__ __launch_bounds__(256) void k_f32_32(const float* src, float* out, Stamp* st, int iters) {
  f32x16 acc[2][2] = {};
  float a[NSET][2], b[NSET][2];
  for (int s = 0; s < NSET; ++s)
    for (int i = 0; i < 2; ++i) { a[s][i] = src[(s * 4 + i) * 256 + threadIdx.x]; b[s][i] = src[(s * 4 + 2 + i) * 256 + threadIdx.x]; }
  Stamp s; stamp_begin(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)  // two K = 2 steps per set = 4 channels
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i ^ kk], b[q][j], acc[i][j], 0, 0, 0);
  }
  stamp_end(s, st);
  float r = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// fp32, 16x16x4: 4x4 blocks of 16x16; per k-step (K = 4) 4 A + 4 B registers, 16 MFMAs
__global__ __launch_bounds__(256) void k_f32_16(const float* src, float* out, Stamp* st, int iters) {
  f32x4 acc[4][4] = {};
  float a[NSET][4], b[NSET][4];
  for (int s = 0; s < NSET; ++s)
    for (int i = 0; i < 4; ++i) { a[s][i] = src[(s * 8 + i) * 256 + threadIdx.x]; b[s][i] = src[(s * 8 + 4 + i) * 256 + threadIdx.x]; }
  Stamp s; stamp_begin(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
  }
  stamp_end(s, st);
  float r = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// bf16, 32x32x16: per k-step (K = 16) 2 A + 2 B fragments, 4 MFMAs
__global__ __launch_bounds__(256) void k_bf_32(const float* src, float* out, Stamp* st, int iters) {
  f32x16 acc[2][2] = {};
  bf16x8 a[NSET][2], b[NSET][2];
  for (int s = 0; s < NSET; ++s)
    for (int i = 0; i < 2; ++i)
      for (int e = 0; e < 8; ++e) {
        a[s][i][e] = (__bf16)src[((s * 4 + i) * 8 + e) * 256 + threadIdx.x];
        b[s][i][e] = (__bf16)src[((s * 4 + 2 + i) * 8 + e) * 256 + threadIdx.x];
      }
  Stamp s; stamp_begin(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)  // two K = 16 steps per set = 32 channels
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i ^ kk], b[q][j], acc[i][j], 0, 0, 0);
  }
  stamp_end(s, st);
  float r = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// bf16, 16x16x32: per k-step (K = 32) 4 A + 4 B fragments, 16 MFMAs
__global__ __launch_bounds__(256) void k_bf_16(const float* src, float* out, Stamp* st, int iters) {
  f32x4 acc[4][4] = {};
  bf16x8 a[NSET][4], b[NSET][4];
  for (int s = 0; s < NSET; ++s)
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 8; ++e) {
        a[s][i][e] = (__bf16)src[((s * 8 + i) * 8 + e) * 256 + threadIdx.x];
        b[s][i][e] = (__bf16)src[((s * 8 + 4 + i) * 8 + e) * 256 + threadIdx.x];
      }
  Stamp s; stamp_begin(s);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
  }
  stamp_end(s, st);
  float r = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) r += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main(int argc, char** argv) {
  const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const bool zeros = argc > 2 && atoi(argv[2]) == 1;
  const int blocks = 256 * wg_per_cu;
  const size_t nsrc = 64 * 8 * 256;
  std::vector<float> h(nsrc);
  srand(1);
  for (auto& v : h) v = zeros ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
  float *src, *out;
  Stamp* st;
  (void)hipMalloc(&src, nsrc * 4);
  (void)hipMalloc(&out, blocks * 256 * 4);
  (void)hipMalloc(&st, blocks * sizeof(Stamp));
  (void)hipMemcpy(src, h.data(), nsrc * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct Case { const char* name; int which; double flop_per_iter_wave; int iters; };
  // per iteration and wave: NSET sets; fp32: 4 channels per set; bf16: 32 channels per set; 64 x 64 outputs
  const Case cases[] = {
      {"fp32 32x32x2 ", 0, NSET * 4.0 * 64 * 64 * 2, 60000 / wg_per_cu},
      {"fp32 16x16x4 ", 1, NSET * 4.0 * 64 * 64 * 2, 60000 / wg_per_cu},
      {"bf16 32x32x16", 2, NSET * 32.0 * 64 * 64 * 2, 120000 / wg_per_cu},
      {"bf16 16x16x32", 3, NSET * 32.0 * 64 * 64 * 2, 120000 / wg_per_cu},
  };
  printf("%d workgroup(s) of 4 waves per CU, %s operands\n", wg_per_cu, zeros ? "ZERO" : "random");
  for (const Case& c : cases) {
    for (int rep = 0; rep < 4; ++rep) {  // the clock settles over the first launches
      (void)hipEventRecord(e0);
      switch (c.which) {
        case 0: hipLaunchKernelGGL(k_f32_32, dim3(blocks), dim3(256), 0, 0, src, out, st, c.iters); break;
        case 1: hipLaunchKernelGGL(k_f32_16, dim3(blocks), dim3(256), 0, 0, src, out, st, c.iters); break;
        case 2: hipLaunchKernelGGL(k_bf_32, dim3(blocks), dim3(256), 0, 0, src, out, st, c.iters); break;
        default: hipLaunchKernelGGL(k_bf_16, dim3(blocks), dim3(256), 0, 0, src, out, st, c.iters); break;
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep == 3) {
        std::vector<Stamp> hs(blocks);
        (void)hipMemcpy(hs.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        for (auto& s : hs) { ghz.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1); cyc.push_back((double)(s.c1 - s.c0)); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double flop = (double)blocks * 4 * c.iters * c.flop_per_iter_wave;
        printf("%s  %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz (median)  loop cycles %.0f (median)\n", c.name, ms, flop / ms / 1e9, ghz[blocks / 2], cyc[blocks / 2]);
      }
    }
  }
  return 0;
}
