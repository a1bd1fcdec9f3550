// Device check of the branch-free source-pixel mapping (common.h SrcMap) against the mode-switch form, every geometry the flat kernels
// serve.  Build: hipcc --offload-arch=gfx950 -O3 -I include tools/srcmap_check.hip -o tools/bin/srcmap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vae-channel-dynamics_amd/csrc/common.h"
__global__ void k(vae_conv_geom g, int* bad, int* first) {
  const SrcMap m = make_srcmap(g);
  int y = blockIdx.x, x = threadIdx.x;
  for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
    if (g.mode == 3 && (((y - kh) & 1) || ((x - kw) & 1))) continue;
    int a = -7, b = -7, c = -7, d = -7;
    bool o1 = src_pixel(g, y, x, kh, kw, a, b), o2 = src_pixel(m, y, x, kh, kw, c, d);
    if (o1 != o2 || (o1 && (a != c || b != d))) { if (atomicAdd(bad, 1) == 0) { first[0] = y; first[1] = x; first[2] = kh; first[3] = kw; first[4] = o1; first[5] = o2; first[6] = a; first[7] = c; } }
  }
}
int main() {
  int *bad, *first; hipMalloc(&bad, 4); hipMalloc(&first, 32);
  for (int mode = 0; mode < 4; ++mode) for (int stride = 1; stride <= 2; ++stride) for (int pad = 0; pad <= 1; ++pad) {
    if (mode == 1 && stride != 1) continue; if (mode == 3 && (stride != 2 || pad != 0)) continue;
    vae_conv_geom g{}; g.Hs = 8; g.Ws = 10; g.stride = stride; g.pad_t = pad; g.pad_l = pad; g.mode = mode;
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k, dim3(24), dim3(64), 0, 0, g, bad, first);
    int hb, hf[8]; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
    printf("mode %d stride %d pad %d: bad %d", mode, stride, pad, hb); if (hb) printf(" first y %d x %d kh %d kw %d o1 %d o2 %d sy %d vs %d", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5], hf[6], hf[7]); printf("\n");
  }
  return 0;
}
