// Host-side error reporting shared by the addk launchers.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "addk.h"

static thread_local char g_err[512] = "";

void addk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int addk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    addk_set_error("%s: %s", what, hipGetErrorString(e));
    return ADDK_ERR_HIP;
  }
  return ADDK_OK;
}

extern "C" const char* addk_last_error(void) { return g_err; }
extern "C" int addk_version(void) { return 1; }
