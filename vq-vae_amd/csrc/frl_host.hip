// Library-level entry points: version, arch probe, thread-local error string.
#include "frl_host.hpp"
#include <string.h>
#include <stdio.h>
#include <thread>
#include <vector>

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

// Host-side staging copy of the tile loader: src -> dst (nbytes) split over `nthreads` std::threads.  One call per chunk from
// Python (ctypes drops the GIL for its duration), so filling the pinned upload buffer does not compete with the training thread
// for the interpreter -- the reference does this work in DataLoader worker processes (train_representation.py: num_workers).
int frl_host_parallel_copy(void* dst, const void* src, size_t nbytes, int nthreads) {
  if ((dst == nullptr || src == nullptr) && nbytes) return frl_fail(-2, "host_parallel_copy: null buffer");
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  const size_t grain = (size_t)1 << 20;                              // below 1 MiB per thread a plain memcpy wins
  if ((size_t)nthreads > nbytes / grain) nthreads = (int)(nbytes / grain);
  if (nthreads <= 1) { memcpy(dst, src, nbytes); return 0; }
  const size_t per = ((nbytes / nthreads) + 4095) & ~(size_t)4095;   // page-aligned shares
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int i = 0; i < nthreads; ++i) {
    const size_t lo = (size_t)i * per;
    if (lo >= nbytes) break;
    const size_t n = (lo + per > nbytes) ? nbytes - lo : per;
    th.emplace_back([=] { memcpy((char*)dst + lo, (const char*)src + lo, n); });
  }
  for (auto& t : th) t.join();
  return 0;
}

}  // extern "C"
