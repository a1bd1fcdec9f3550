// thread-local last-error string of the C ABI, ABI version, process-wide kernel-selection switches
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common_host.h"

static thread_local char g_err[512] = "";

void vae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Switches: read from the environment ONCE (when the library is loaded), afterwards changed only through vae_set_option.
// No launch path calls getenv().
static vae_options init_options() {
  vae_options o;
  o.flat_conv = getenv("VAEHIP_FLAT_CONV") ? 1 : 0;
  o.no_wino = getenv("VAEHIP_NO_WINO") ? 1 : 0;
  o.no_wide = getenv("VAEHIP_NO_WIDE") ? 1 : 0;
  o.no_wino4 = getenv("VAEHIP_NO_WINO4") ? 1 : 0;
  o.no_thin_mfma = getenv("VAEHIP_NO_THIN_MFMA") ? 1 : 0;
  o.no_wgrad_dma = getenv("VAEHIP_NO_WGRAD_DMA") ? 1 : 0;
  const char* r = getenv("VAEHIP_WIDE_RESERVED_CUS");
  o.wide_reserved_cus = r ? atoi(r) : 0;
  if (o.wide_reserved_cus < 0 || o.wide_reserved_cus > 128) o.wide_reserved_cus = 0;
  return o;
}
static vae_options g_opt = init_options();
const vae_options& vae_opt() { return g_opt; }

static int* option_slot(const char* name) {
  if (!name) return nullptr;
  if (!strcmp(name, "flat_conv")) return &g_opt.flat_conv;
  if (!strcmp(name, "no_wino")) return &g_opt.no_wino;
  if (!strcmp(name, "no_wide")) return &g_opt.no_wide;
  if (!strcmp(name, "no_wino4")) return &g_opt.no_wino4;
  if (!strcmp(name, "no_thin_mfma")) return &g_opt.no_thin_mfma;
  if (!strcmp(name, "no_wgrad_dma")) return &g_opt.no_wgrad_dma;
  if (!strcmp(name, "wide_reserved_cus")) return &g_opt.wide_reserved_cus;
  return nullptr;
}

extern "C" const char* vae_last_error(void) { return g_err; }
extern "C" int vae_abi_version(void) { return 12; }
extern "C" int vae_sizeof_args(int32_t which) {
  return which == 0 ? (int)sizeof(vae_conv_geom) : which == 1 ? (int)sizeof(vae_igemm_args) : which == 2 ? (int)sizeof(vae_wgrad_args) : -1;
}
extern "C" int vae_set_option(const char* name, int32_t value) {
  int* s = option_slot(name);
  if (!s) {
    vae_set_error("set_option: unknown option '%s' (flat_conv, no_wino, no_wino4, no_wide, no_thin_mfma, no_wgrad_dma, wide_reserved_cus)", name ? name : "(null)");
    return VAE_EINVAL;
  }
  if (s == &g_opt.wide_reserved_cus) {  // a count, not a switch
    if (value < 0 || value > 128) {
      vae_set_error("set_option: wide_reserved_cus must be in 0..128, got %d", (int)value);
      return VAE_EINVAL;
    }
    *s = value;
    return VAE_OK;
  }
  *s = value ? 1 : 0;
  return VAE_OK;
}
extern "C" int vae_get_option(const char* name) {
  int* s = option_slot(name);
  return s ? *s : -1;
}
