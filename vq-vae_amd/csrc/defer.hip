// Deferred weight-gradient reductions: one launch for all slab reductions of a backward pass (frl_reduce.hpp, frl_epi.hpp).
// A train step launches ~13 weight-gradient kernels, each followed by a ~5 us fixed-order reduction of its per-workgroup slabs that
// nothing but the optimizer consumes.  frl_defer_begin() makes those reductions register themselves instead; frl_defer_flush() runs
// them in one kernel whose job table travels in the kernel-argument segment (so a captured hipGraph replays it with the same pointers).
// The summation order of every column is the one of slab_reduce_t: results are bit-identical to the undeferred path.
#include "frl_reduce.hpp"
#include <mutex>
#include <vector>

static std::mutex g_defer_mu;
static std::vector<FrlDeferJob> g_defer_jobs;
static int g_defer_on = 0;

int frl_defer_active_() { return g_defer_on; }

int frl_defer_push_(const FrlDeferJob& job) {
  std::lock_guard<std::mutex> lk(g_defer_mu);
  if (!g_defer_on || g_defer_jobs.size() >= FRL_DEFER_MAX_JOBS) return 0;
  g_defer_jobs.push_back(job);
  return 1;
}

struct FrlDeferTable {
  int njobs;
  FrlDeferJob job[FRL_DEFER_MAX_JOBS];
};

template <class Epi> __device__ __forceinline__ void defer_epi(const unsigned char* payload, int64_t i, float s) {
  (*reinterpret_cast<const Epi*>(payload))(i, s);
}

// same thread layout and summation order as slab_reduce_t (frl_reduce.hpp): 256 threads = 32 columns x 8 slab groups
__global__ __launch_bounds__(256) void slab_reduce_jobs_kernel(const FrlDeferTable tab) {
  __shared__ float red[8][33];
  int j = 0;
  while (j + 1 < tab.njobs && blockIdx.x >= tab.job[j + 1].first_block) ++j;       // block-uniform: scalar loads from the argument segment
  const FrlDeferJob& job = tab.job[j];
  const float* __restrict__ slab = job.slab;
  const int nslab = job.nslab;
  const int64_t n = job.n;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = (int64_t)(blockIdx.x - job.first_block) * 32 + col;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  if (i < n) {
    int k = grp;
    for (; k + 56 < nslab; k += 64) {
      s0 += slab[(int64_t)k * n + i];
      s1 += slab[(int64_t)(k + 8) * n + i];
      s2 += slab[(int64_t)(k + 16) * n + i];
      s3 += slab[(int64_t)(k + 24) * n + i];
      s4 += slab[(int64_t)(k + 32) * n + i];
      s5 += slab[(int64_t)(k + 40) * n + i];
      s6 += slab[(int64_t)(k + 48) * n + i];
      s7 += slab[(int64_t)(k + 56) * n + i];
    }
    for (; k < nslab; k += 8) s0 += slab[(int64_t)k * n + i];
  }
  s0 += s4; s1 += s5; s2 += s6; s3 += s7;
  red[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < n) {
    float s = red[0][col];
#pragma unroll
    for (int g = 1; g < 8; ++g) s += red[g][col];
    switch (job.kind) {
      case FrlEpiKind<ThEpi>::id: defer_epi<ThEpi>(job.payload, i, s); break;
      case FrlEpiKind<C3Epi>::id: defer_epi<C3Epi>(job.payload, i, s); break;
      case FrlEpiKind<DecEpi>::id: defer_epi<DecEpi>(job.payload, i, s); break;
      case FrlEpiKind<EncEpi>::id: defer_epi<EncEpi>(job.payload, i, s); break;
      case FrlEpiKind<FilmEpi>::id: defer_epi<FilmEpi>(job.payload, i, s); break;
      case FrlEpiKind<ShEpi>::id: defer_epi<ShEpi>(job.payload, i, s); break;
      case FrlEpiKind<WgradEpi>::id: defer_epi<WgradEpi>(job.payload, i, s); break;
      case FrlEpiKind<CodeEpi>::id: defer_epi<CodeEpi>(job.payload, i, s); break;
      default: break;
    }
  }
}

extern "C" {

// Start deferring: every deferrable slab reduction issued (by any thread) until the flush is parked.  Returns -2 when already active.
int frl_defer_begin(void) {
  std::lock_guard<std::mutex> lk(g_defer_mu);
  if (g_defer_on) return frl_fail(-2, "frl_defer_begin: deferral is already active");
  g_defer_jobs.clear();
  g_defer_on = 1;
  return 0;
}

int frl_defer_pending(void) {
  std::lock_guard<std::mutex> lk(g_defer_mu);
  return (int)g_defer_jobs.size();
}

// Run every parked reduction in one launch on `stream` (which must be ordered behind the kernels that wrote the slabs) and stop deferring.
// Returns the number of jobs run, or a negative code.
int frl_defer_flush(hipStream_t stream) {
  FrlDeferTable tab;
  {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!g_defer_on) return frl_fail(-2, "frl_defer_flush: deferral is not active");
    g_defer_on = 0;
    tab.njobs = (int)g_defer_jobs.size();
    unsigned blocks = 0;
    for (int j = 0; j < tab.njobs; ++j) {
      tab.job[j] = g_defer_jobs[j];
      tab.job[j].first_block = blocks;
      blocks += (unsigned)((g_defer_jobs[j].n + 31) / 32);
    }
    g_defer_jobs.clear();
    if (tab.njobs == 0) return 0;
    FRL_LAUNCH(slab_reduce_jobs_kernel, dim3(blocks), dim3(256), 0, stream, tab);
  }
  const int rc = frl_check_launch("defer_flush");
  return rc ? rc : tab.njobs;
}

// Destination pointers of the parked jobs (every non-null gradient pointer of every epilogue), for the caller's check that they still are
// the tensors the optimizer will read.  Returns how many were written (at most `max`).
int frl_defer_destinations(void** out, int max) {
  std::lock_guard<std::mutex> lk(g_defer_mu);
  int n = 0;
  auto put = [&](const void* p) { if (p != nullptr && n < max) out[n++] = const_cast<void*>(p); };
  for (const FrlDeferJob& j : g_defer_jobs) {
    switch (j.kind) {
      case FrlEpiKind<ThEpi>::id: { ThEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dWc); put(e.dWg); put(e.dbc); put(e.dbg); put(e.dgam); put(e.dbet); break; }
      case FrlEpiKind<C3Epi>::id: { C3Epi e; memcpy(&e, j.payload, sizeof(e)); put(e.dW); put(e.dB); break; }
      case FrlEpiKind<DecEpi>::id: { DecEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dW2); put(e.dW1); put(e.db2); put(e.db1); break; }
      case FrlEpiKind<EncEpi>::id: { EncEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dw2); put(e.dw1); put(e.db2); put(e.dg2); put(e.db1); put(e.dg1); break; }
      case FrlEpiKind<FilmEpi>::id: { FilmEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dw1g); put(e.dw1b); put(e.dw2g); put(e.dw2b); put(e.db1g); put(e.db1b); put(e.db2g); put(e.db2b); break; }
      case FrlEpiKind<ShEpi>::id: { ShEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dwb); put(e.dwa); put(e.dbb); put(e.dba); break; }
      case FrlEpiKind<WgradEpi>::id: { WgradEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.dW); put(e.dB); break; }
      case FrlEpiKind<CodeEpi>::id: { CodeEpi e; memcpy(&e, j.payload, sizeof(e)); put(e.gE); break; }
      default: break;
    }
  }
  return n;
}

// Drop the parked jobs without running them (error paths: the gradients of the step are abandoned).
int frl_defer_abort(void) {
  std::lock_guard<std::mutex> lk(g_defer_mu);
  g_defer_on = 0;
  g_defer_jobs.clear();
  return 0;
}

}  // extern "C"
