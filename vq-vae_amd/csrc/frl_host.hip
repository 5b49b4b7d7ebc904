// Library-level entry points: version, arch probe, thread-local error string.
#include "frl_host.hpp"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";
thread_local hipError_t g_frl_pending = hipSuccess;

int frl_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "unknown error");
  return code;
}

int frl_check_launch(const char* what) {
  hipError_t e = g_frl_pending;
  g_frl_pending = hipSuccess;
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -100 - (int)e;
  }
  return 0;
}

extern "C" {

int frl_version(void) { return 100; }  // 0.1.0

const char* frl_last_error(void) { return g_err; }

// Writes the gcnArchName of the current device (e.g. "gfx950:sramecc+:xnack-"); 0 on success.
int frl_device_arch(char* buf, int n) {
  int dev = 0;
  FRL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  FRL_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, (size_t)n, "%s", prop.gcnArchName);
  return 0;
}

}  // extern "C"
