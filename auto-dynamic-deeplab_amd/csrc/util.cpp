// Host-side error reporting shared by the addk launchers.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
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

// Environment switches (A/B aids; INTEGRATION.md lists them): every one is read ONCE, under a lock, through this function — the only
// getenv of the library.  An unset variable yields `dflt`; a set one its integer value.
int addk_env(const char* name, int dflt) {
  static std::mutex mu;
  static struct { const char* name; int value; } seen[64];
  static int n = 0;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n; ++i) if (seen[i].name == name || !strcmp(seen[i].name, name)) return seen[i].value;
  const char* e = getenv(name);
  const int v = (e && *e) ? atoi(e) : dflt;
  if (n < 64) { seen[n].name = name; seen[n].value = v; ++n; }
  return v;
}
// ADDK_MATH is the one switch with names for values: fp32 | bf16x6 | bf16x3 | tail_x3, or the mode number
int addk_env_math(int dflt) {
  static std::once_flag once; static int v;
  std::call_once(once, [&] {
    v = dflt;
    const char* e = getenv("ADDK_MATH");
    if (!e || !*e) return;
    if (!strcmp(e, "fp32") || !strcmp(e, "0")) v = 0;
    else if (!strcmp(e, "bf16x3") || !strcmp(e, "1")) v = 1;
    else if (!strcmp(e, "bf16x6") || !strcmp(e, "2")) v = 2;
    else if (!strcmp(e, "tail_x3") || !strcmp(e, "3")) v = 3;
  });
  return v;
}
