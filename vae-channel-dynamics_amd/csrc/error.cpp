// thread-local last-error string of the C ABI
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vaehip.h"

static thread_local char g_err[512] = "";

void vae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vae_last_error(void) { return g_err; }
extern "C" int vae_abi_version(void) { return 8; }
extern "C" int vae_sizeof_args(int32_t which) {
  return which == 0 ? (int)sizeof(vae_conv_geom) : which == 1 ? (int)sizeof(vae_igemm_args) : which == 2 ? (int)sizeof(vae_wgrad_args) : -1;
}
