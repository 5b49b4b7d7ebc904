// Fixed-order parallel reduction of per-workgroup slabs:  v[i] = sum_k slab[k * n + i],  epi(i, v[i]).
// 256 threads = 32 columns x 8 slab groups; each thread walks its group's slabs 8 at a time (independent loads in
// flight), the 8 groups are combined through LDS in a fixed order, so results are bit-reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frl_host.hpp"
#include "frl_epi.hpp"
#include <string.h>

template <typename V, class Epi>
__global__ __launch_bounds__(256) void slab_reduce_t(const V* __restrict__ slab, int nslab, int64_t n, Epi epi) {
  __shared__ V red[8][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + col;
  V s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  if (i < n) {
    int k = grp;
    for (; k + 56 < nslab; k += 64) {            // eight independent loads in flight per thread
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
    V s = red[0][col];
#pragma unroll
    for (int g = 1; g < 8; ++g) s += red[g][col];
    epi(i, s);
  }
}

template <typename V, class Epi>
static inline void launch_slab_reduce(const V* slab, int nslab, int64_t n, Epi epi, hipStream_t st) {
  const unsigned grid = (unsigned)((n + 31) / 32);
  FRL_LAUNCH((slab_reduce_t<V, Epi>), dim3(grid), dim3(256), 0, st, slab, nslab, n, epi);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Deferred reductions (defer.hip).  Between frl_defer_begin() and frl_defer_flush(stream) a deferrable reduction is not launched: its
// slab pointer, shape and epilogue are appended to a job list, and the flush runs every job in ONE launch.  The caller guarantees that
// the slabs (the workspaces of the deferred calls) stay untouched until the flush and that nothing reads the gradients before it.
#define FRL_DEFER_MAX_JOBS 24
#define FRL_DEFER_PAYLOAD 72
struct FrlDeferJob {
  const float* slab;
  int64_t n;
  int nslab, kind;
  unsigned first_block, pad_;
  unsigned char payload[FRL_DEFER_PAYLOAD];
};
int frl_defer_active_();                                  // 1 between begin and flush
int frl_defer_push_(const FrlDeferJob& job);              // 1 when the job was taken (0: list full -> the caller launches as usual)

template <typename V, class Epi>
static inline void launch_slab_reduce_deferrable(const V* slab, int nslab, int64_t n, Epi epi, hipStream_t st, bool allow = true) {
  static_assert(FrlEpiKind<Epi>::id != 0 && sizeof(Epi) <= FRL_DEFER_PAYLOAD && sizeof(V) == sizeof(float), "not a deferrable epilogue");
  if (allow && frl_defer_active_()) {
    FrlDeferJob job;
    memset(&job, 0, sizeof(job));
    job.slab = (const float*)slab; job.n = n; job.nslab = nslab; job.kind = FrlEpiKind<Epi>::id;
    memcpy(job.payload, &epi, sizeof(Epi));
    if (frl_defer_push_(job)) return;
  }
  launch_slab_reduce<V, Epi>(slab, nslab, n, epi, st);
}
