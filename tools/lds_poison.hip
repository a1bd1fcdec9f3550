// Debug aid: fill the LDS of every CU with a NaN pattern (one workgroup per CU takes the whole 160 KB), so that a kernel which
// consumes LDS it never wrote shows up as NaN in its output instead of depending on what ran on that CU before.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_poison.hip -o tools/bin/liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, int words) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
  __syncthreads();
  if (lds[(threadIdx.x * 37) % words] != pattern) __builtin_trap();  // (keeps the stores)
}
extern "C" int lds_poison(void* stream, unsigned pattern) {
  const int bytes = 160 * 1024;
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    set = true;
  }
  hipLaunchKernelGGL(lds_poison_kernel, dim3(256), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
