// Host-side error reporting shared by the addk launchers.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
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
// ADDK_MATH is the one switch with names for values: fp32 | f16x3 (old name: bf16x3) | bf16x6 | tail_x3, or the mode number
int addk_env_math(int dflt) {
  static std::once_flag once; static int v;
  std::call_once(once, [&] {
    v = dflt;
    const char* e = getenv("ADDK_MATH");
    if (!e || !*e) return;
    if (!strcmp(e, "fp32") || !strcmp(e, "0")) v = 0;
    else if (!strcmp(e, "f16x3") || !strcmp(e, "bf16x3") || !strcmp(e, "1")) v = 1;
    else if (!strcmp(e, "bf16x6") || !strcmp(e, "2")) v = 2;
    else if (!strcmp(e, "tail_x3") || !strcmp(e, "3")) v = 3;
  });
  return v;
}

// Diagnostic: native stack of a SIGABRT / SIGSEGV on stderr (backtrace_symbols_fd is async-signal-safe enough for a dying process), then the
// default action.  Python's faulthandler shows the Python frames only; two aborts of earlier rounds left no native frame behind.
static struct sigaction g_old_abrt, g_old_segv;
static void addk_abort_handler(int sig) {
  static const char msg[] = "\n[addk] fatal signal, native stack:\n";
  (void)!write(2, msg, sizeof msg - 1);
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  // hand over to whoever was installed before (Python's faulthandler prints the Python frames and re-raises), else die with the default action
  const struct sigaction& old = sig == SIGABRT ? g_old_abrt : g_old_segv;
  sigaction(sig, &old, nullptr);
  raise(sig);
}
extern "C" int addk_debug_trace_fatal_signals(void) {
  static bool done = false;
  if (done) return ADDK_OK;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = addk_abort_handler;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGABRT, &sa, &g_old_abrt);
  sigaction(SIGSEGV, &sa, &g_old_segv);
  done = true;
  return ADDK_OK;
}
