// Mutual k-nearest-neighbour pair mining for the contrastive objective:
//   pairs_mutual_knn_chunked  frl/losses/pairs.py:531-610 -- per query the k nearest anchors in feature space (L2), never itself,
//   never a same-patch anchor closer than pos_min_spatial pixels; (i, j) is a pair iff j is among i's neighbours AND i among j's.
// The reference walks chunk_size x N blocks of torch.cdist + topk; here one wave owns one query: its N squared distances are
// written to LDS once (direct differences, float32), then k rounds of a wave-wide arg-min by (distance, index) pick the
// neighbours in ascending order -- no N x N matrix in HBM, no sort.  A second kernel marks the mutual entries.
#include "frl_common.hpp"
#include "frl_host.hpp"

#define KNN_WAVES 4

// LDS: q [KNN_WAVES][D] | tile [64][D + 4] | dist [KNN_WAVES][N].  The workgroup's four waves (four queries) walk the targets in
// tiles of 64 rows that are fetched ONCE per workgroup with coalesced 16-byte loads (a lane reading "its own row" straight from
// global memory touches 64 cache lines per instruction and is L1-bound); lane j of every wave then reads row j of the tile (pitch
// D + 4 floats: conflict-free 16-byte LDS reads) against its query.  The next tile is prefetched into registers behind the arithmetic.
template <int D4PT>   // float4 pieces of a 64-row tile per thread = 64 * (D / 4) / 256
__global__ __launch_bounds__(64 * KNN_WAVES) void knn_kernel(const float* __restrict__ feat, int N, int D, const int* __restrict__ patch_id,
                                                             const float* __restrict__ coords, float min_spatial, int k,
                                                             int* __restrict__ knn_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int d4 = 4 * D4PT;                                                     // D / 4: the host dispatches on D = 16 * D4PT
  const int pitch = D + 4;
  float* qall = reinterpret_cast<float*>(smem);
  float* tile = qall + (size_t)KNN_WAVES * D;
  float* dist = tile + (size_t)64 * pitch + (size_t)wave * N;
  const int iq = blockIdx.x * KNN_WAVES + wave;
  const bool live = iq < N;
  const int i = live ? iq : N - 1;                                               // surplus waves shadow the last query (no stores)
  float* q = qall + (size_t)wave * D;
  for (int d = lane; d < D; d += 64) q[d] = feat[(size_t)i * D + d];
  const int pi = patch_id[i];
  const float ci0 = coords[2 * i], ci1 = coords[2 * i + 1];
  const float inf = __builtin_inff();
  const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
  const int ntiles = (N + 63) >> 6;
  f32x4 preA[D4PT], preB[D4PT];                                                  // two tiles in flight: with the distance rows in LDS only
                                                                                 // one wave per SIMD is resident, nobody else hides the latency
  auto fetch = [&](f32x4 (&pre)[D4PT], int tix) {                                // piece p of the tile: row p / d4, float4 column p % d4
#pragma unroll
    for (int u = 0; u < D4PT; ++u) {
      const int p = tid + 256 * u;
      int row = tix * 64 + p / d4;
      if (row >= N) row = N - 1;
      pre[u] = *reinterpret_cast<const f32x4*>(feat + (size_t)row * D + 4 * (p % d4));
    }
  };
  auto step = [&](f32x4 (&pre)[D4PT], int tix) {
    __syncthreads();                                                             // everyone is done with the previous tile (and q is written)
#pragma unroll
    for (int u = 0; u < D4PT; ++u) {
      const int p = tid + 256 * u;
      *reinterpret_cast<f32x4*>(tile + (size_t)(p / d4) * pitch + 4 * (p % d4)) = pre[u];
    }
    __syncthreads();
    if (tix + 2 < ntiles) fetch(pre, tix + 2);
    const int j = tix * 64 + lane;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(tile + (size_t)lane * pitch);
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < d4; ++d) {                                               // several LDS reads in flight per wave
      const f32x4 a = q4[d], b = x4[d];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = a[e] - b[e]; s = fmaf(t, t, s); }
    }
    if (j < N) {
      if (j == i) s = inf;                                                       // never itself
      else if (patch_id[j] == pi) {                                              // nor a same-patch anchor closer than min_spatial pixels
        const float d0 = ci0 - coords[2 * j], d1 = ci1 - coords[2 * j + 1];
        if (sqrtf(d0 * d0 + d1 * d1) < min_spatial) s = inf;
      }
      dist[j] = s;
    }
  };
  fetch(preA, 0);
  if (ntiles > 1) fetch(preB, 1);
  for (int tix = 0; tix < ntiles; tix += 2) {
    step(preA, tix);
    if (tix + 1 < ntiles) step(preB, tix + 1);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- k selection rounds.  Lane l owns the entries j = l, l + 64, ...; its running minimum (lbest, lj) is found once.  A round
  // takes the wave-wide arg-min by (distance, index), retires the winner, and only the winner's owner needs a new minimum -- which
  // all 64 lanes look for together in that owner's entries (two LDS reads each instead of a full rescan by every lane).
  auto wave_argmin = [&](float& v, int& jx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off, 64);
      const int oj = __shfl_xor(jx, off, 64);
      if (ov < v || (ov == v && oj < jx)) { v = ov; jx = oj; }
    }
  };
  float lbest = inf;
  int lj = 0x7fffffff;
  {
    int j = lane;
    for (; j + 7 * 64 < N; j += 8 * 64) {                                        // eight LDS reads in flight, then the ordered compares
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dist[j + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] < lbest) { lbest = v[u]; lj = j + 64 * u; }                     // ascending j per lane: first index wins ties
    }
    for (; j < N; j += 64) {
      const float v = dist[j];
      if (v < lbest) { lbest = v; lj = j; }
    }
  }
  for (int r = 0; r < k; ++r) {
    float best = lbest;
    int bj = lj;
    wave_argmin(best, bj);
    const bool found = best < inf;                                               // wave-uniform
    if (live && lane == 0) knn_idx[(size_t)i * k + r] = found ? bj : -1;
    if (!found) continue;                                                        // nothing admissible is left: the remaining slots are -1 too
    const int owner = bj & 63;
    if (lane == owner) dist[bj] = inf;                                           // retire the winner
    __builtin_amdgcn_wave_barrier();
    float cb = inf;
    int cj = 0x7fffffff;
    for (int j = owner + 64 * lane; j < N; j += 64 * 64) {                       // the owner's entries, spread over the wave
      const float v = dist[j];
      if (v < cb) { cb = v; cj = j; }
    }
    wave_argmin(cb, cj);
    if (lane == owner) { lbest = cb; lj = cb < inf ? cj : 0x7fffffff; }
  }
}

__global__ __launch_bounds__(256) void knn_mutual_kernel(const int* __restrict__ knn_idx, int N, int k, uint8_t* __restrict__ mutual) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)N * k) return;
  const int i = (int)(e / k);
  const int j = knn_idx[e];
  bool m = false;
  if (j >= 0)
    for (int r = 0; r < k; ++r) m |= knn_idx[(size_t)j * k + r] == i;
  mutual[e] = m ? 1 : 0;
}

static size_t knn_lds_bytes(int N, int D) {
  return ((size_t)KNN_WAVES * D + (size_t)64 * (D + 4) + (size_t)KNN_WAVES * N) * sizeof(float);
}

template <int D4PT>
static int knn_launch(const float* feat, int N, int D, const int32_t* patch_id, const float* coords, float pos_min_spatial, int k,
                      int32_t* knn_idx, size_t lds, hipStream_t stream) {
  auto kern = knn_kernel<D4PT>;
  if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH_AS("knn_kernel", kern, dim3((N + KNN_WAVES - 1) / KNN_WAVES), dim3(64 * KNN_WAVES), lds, stream, feat, N, D, patch_id, coords, pos_min_spatial,
             k, knn_idx);
  return 0;
}

extern "C" {

size_t frl_mutual_knn_max_points(int D) {
  const size_t lds = 160 * 1024, fixed = knn_lds_bytes(0, D);
  return lds > fixed ? (lds - fixed) / (KNN_WAVES * sizeof(float)) : 0;
}

// feat [N][D] float32 (D a multiple of 16, at most 256), patch_id [N] int32, coords [N][2] float32 (row, col), knn_idx [N][k] int32
// out (-1 = fewer than k valid neighbours), mutual [N][k] bytes out (1 where the pair (i, knn_idx[i][r]) is mutual).
int frl_mutual_knn(const float* feat, int N, int D, const int32_t* patch_id, const float* coords, float pos_min_spatial, int k,
                   int32_t* knn_idx, uint8_t* mutual, hipStream_t stream) {
  if (N <= 0 || D <= 0 || k <= 0) return frl_fail(-2, "mutual_knn: N, D and k must be positive");
  if ((D & 15) || D > 256) return frl_fail(-2, "mutual_knn: the feature width must be a multiple of 16, at most 256 (pad with zeros)");
  if ((size_t)N > frl_mutual_knn_max_points(D)) return frl_fail(-3, "mutual_knn: the distance rows of four queries do not fit the LDS");
  const size_t lds = knn_lds_bytes(N, D);
  int rc;
  switch (D / 16) {   // D4PT = 64 * (D / 4) / 256 = D / 16
    case 1: rc = knn_launch<1>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 2: rc = knn_launch<2>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 3: rc = knn_launch<3>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 4: rc = knn_launch<4>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 6: rc = knn_launch<6>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 8: rc = knn_launch<8>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    case 16: rc = knn_launch<16>(feat, N, D, patch_id, coords, pos_min_spatial, k, knn_idx, lds, stream); break;
    default: return frl_fail(-2, "mutual_knn: feature width must be 16, 32, 48, 64, 96, 128 or 256");
  }
  if (rc) return rc;
  const int64_t total = (int64_t)N * k;
  FRL_LAUNCH(knn_mutual_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, knn_idx, N, k, mutual);
  return frl_check_launch("mutual_knn");
}

}  // extern "C"
