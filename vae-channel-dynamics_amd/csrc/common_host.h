// Host-side declarations shared by every translation unit of libvaehip (no device code).
#pragma once
#include <stdint.h>
#include "../../include/vaehip.h"

void vae_set_error(const char* fmt, ...);

// process-wide kernel-selection switches (error.cpp): VAEHIP_FLAT_CONV / VAEHIP_NO_WINO / VAEHIP_NO_WIDE are read once at
// load time; vae_set_option changes them afterwards (tests compare two algorithms that way)
struct vae_options {
  int flat_conv;  // flat implicit-GEMM kernels everywhere (the second algorithm of the two-algorithm tests)
  int no_wino;    // fp32: direct halo-tile kernels instead of the Winograd ones (the parity reference)
  int no_wino4;   // fp32: Winograd F(2x2,3x3) also where F(4x4,3x3) would serve the layer (A/B switch, second Winograd algorithm of the tests)
  int no_thin_mfma;  // bf16: the <= 4-channel-side layers on the VALU kernels (skinny.hip) also where the matrix-pipe kernels would serve them
  int no_wide;    // bf16: the 128-pixel halo-tile kernel instead of the wide-tile one
  int no_wgrad_dma;  // bf16: the 3x3 weight gradient stages its bf16 images through registers (A/B switch for the LDS-DMA kernel)
  // bf16, data parallel: the persistent wide-tile kernel launches (256 - n) workgroups instead of one per CU, leaving n CUs
  // to RCCL's workgroups while gradient buckets are in flight (0 = all 256; a count 0..128, not a switch)
  int wide_reserved_cus;
};
const vae_options& vae_opt();

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize set once per kernel AND device
// (one bit per device ordinal; a local static per call site, i.e. per template instantiation).  Returns VAE_ELAUNCH from the caller.
#define VAE_RESERVE_LDS(kern, bytes, what)                                                                                  \
  do {                                                                                                                      \
    static unsigned long long done_ = 0;                                                                                    \
    int dev_ = 0;                                                                                                           \
    (void)hipGetDevice(&dev_);                                                                                              \
    if (!((done_ >> (dev_ & 63)) & 1ull)) {                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) { \
        vae_set_error("%s: cannot reserve %d bytes of LDS", what, (int)(bytes));                                            \
        return VAE_ELAUNCH;                                                                                                 \
      }                                                                                                                     \
      done_ |= 1ull << (dev_ & 63);                                                                                         \
    }                                                                                                                       \
  } while (0)
