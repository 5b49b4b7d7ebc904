// Library-level entry points: version, arch probe, thread-local error string.
#include "frl_host.hpp"
#include <string.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
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

// ---- per-kernel event timing ------------------------------------------------------------------------------------------------
int g_frl_timing = 0;
namespace {
struct TimedLaunch { std::string name; hipEvent_t e0, e1; };
std::mutex g_tm_mu;
std::vector<TimedLaunch> g_tm_done;                 // recorded pairs, resolved by frl_kernel_timing_report
std::vector<hipEvent_t> g_tm_pool;                  // recycled events
thread_local TimedLaunch g_tm_cur;
hipEvent_t tm_event() {
  std::lock_guard<std::mutex> lk(g_tm_mu);
  if (!g_tm_pool.empty()) { hipEvent_t e = g_tm_pool.back(); g_tm_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void frl_timing_begin(const char* kernel, hipStream_t st) {
  g_tm_cur.name = kernel;
  g_tm_cur.e0 = tm_event();
  g_tm_cur.e1 = tm_event();
  (void)hipEventRecord(g_tm_cur.e0, st);
}
void frl_timing_end(hipStream_t st) {
  (void)hipEventRecord(g_tm_cur.e1, st);
  std::lock_guard<std::mutex> lk(g_tm_mu);
  g_tm_done.push_back(g_tm_cur);
}

extern "C" {

int frl_version(void) { return 100; }  // 0.1.0

// on != 0: every kernel launched by the library from now on is bracketed by a HIP event pair on its launch stream (two extra
// stream operations per launch: for measurement runs only).  Returns the previous setting.
int frl_kernel_timing_enable(int on) {
  const int was = g_frl_timing;
  g_frl_timing = on ? 1 : 0;
  return was;
}

// Synchronises the recorded events and writes one line per kernel expression, "name\tcalls\ttotal_ms\n", into buf (NUL terminated,
// truncated to n bytes); clears the record.  Returns the number of distinct kernels, or a negative code.
int frl_kernel_timing_report(char* buf, int n) {
  std::vector<TimedLaunch> done;
  {
    std::lock_guard<std::mutex> lk(g_tm_mu);
    done.swap(g_tm_done);
  }
  std::map<std::string, std::pair<long, double>> acc;
  for (auto& t : done) {
    float ms = 0.f;
    if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
      auto& a = acc[t.name];
      a.first += 1;
      a.second += (double)ms;
    }
    std::lock_guard<std::mutex> lk(g_tm_mu);
    g_tm_pool.push_back(t.e0);
    g_tm_pool.push_back(t.e1);
  }
  size_t off = 0;
  if (buf != nullptr && n > 0) buf[0] = 0;
  for (auto& kv : acc) {
    char line[768];
    const int len = snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    if (buf != nullptr && len > 0 && off + (size_t)len + 1 <= (size_t)n) { memcpy(buf + off, line, (size_t)len + 1); off += (size_t)len; }
  }
  return (int)acc.size();
}

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
