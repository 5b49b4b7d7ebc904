// Sparse-location gather and the InfoNCE loss over mined pairs:
//   * extract_at_locations  (frl/utils/spatial.py:132-173; callers frl/training/representation/step.py:526,562,602): feature vectors at
//     (row, col) anchors of one raster, any (channel, row, col) strides -- the NHWC rows of this library and the reference's
//     [C, H, W] view are both zero-copy;
//   * contrastive_loss      (frl/losses/contrastive.py:29-212; callers step.py:563,787): per anchor
//         L_a = -log( sum_p w_p exp(s_ap / t) + eps ) + log( sum_{p,n} w exp(s / t) + eps )        (both sums taken relative to the
//     anchor's largest logit, eps = 1e-8 inside the logarithms exactly as the reference's stabilised form), mean over the anchors that
//     have a positive.
// The reference groups pairs with scatter_reduce / scatter_add (atomics: summation order varies from run to run).  Here the pairs
// arrive SORTED by anchor (stable sort done by the caller with torch.sort: index plumbing, as the reference's own torch.unique), so a
// segment is a contiguous run and one wave reduces it in a fixed order -> bit-reproducible loss and gradients.  The backward
// scatter (rows of the embedding matrix that occur in many pairs) is the same primitive: frl_segment_sum_rows over a sorted key.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include <math.h>

// ---------------------------------------------------------------------------------------------------------------------------
// out[n][c] = feat[c * sC + row_n * sH + col_n * sW]
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_locations_kernel(const T* __restrict__ feat, int64_t sC, int64_t sH, int64_t sW, int C, int H, int W,
                                                               const int64_t* __restrict__ coords, int64_t N, T* __restrict__ out) {
  const int64_t total = N * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    int64_t r = coords[2 * n], q = coords[2 * n + 1];
    if (r < 0) r += H;                                            // torch advanced indexing accepts negative indices
    if (q < 0) q += W;
    out[i] = feat[c * sC + r * sH + q * sW];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Segmented row sum over a key-sorted list:  out[key] (+)= sum over the run of equal keys of vals[order[i]][:]  (rows summed in list
// order = fixed order).  One thread per (run head, column block); threads that are not at the head of a run leave.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float* __restrict__ vals, const int64_t* __restrict__ order,
                                                               const int64_t* __restrict__ keys_sorted, int64_t M, int D,
                                                               float* __restrict__ out, int64_t out_stride, int accumulate) {
  const int64_t total = M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / D;
    const int c = (int)(i - p * D);
    const int64_t key = keys_sorted[p];
    if (p > 0 && keys_sorted[p - 1] == key) continue;             // not the head of its run
    float s = 0.f;
    for (int64_t j = p; j < M && keys_sorted[j] == key; ++j) s += vals[(order ? order[j] : j) * (int64_t)D + c];
    float* o = out + key * out_stride + c;
    *o = accumulate ? *o + s : s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// per pair: similarity (l2: -|a-b|^2 / D, cosine, dot) and logit = log(w) + sim / t.  One 16-lane group per pair.
// ---------------------------------------------------------------------------------------------------------------------------
enum { FRL_SIM_L2 = 0, FRL_SIM_COSINE = 1, FRL_SIM_DOT = 2 };

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}

__global__ __launch_bounds__(256) void pair_logits_kernel(const float* __restrict__ emb, int D, const int64_t* __restrict__ pairs,
                                                          const float* __restrict__ weights, int64_t T, float inv_t, int sim_kind,
                                                          float* __restrict__ sims, float* __restrict__ logits) {
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  for (int64_t p = (int64_t)blockIdx.x * 16 + g; p < T; p += (int64_t)gridDim.x * 16) {
    const float* a = emb + pairs[2 * p] * (int64_t)D;
    const float* b = emb + pairs[2 * p + 1] * (int64_t)D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;                           // l2: sum (a-b)^2 | else: a.b, a.a, b.b
    for (int j = l; j < D; j += 16) {
      const float av = a[j], bv = b[j];
      if (sim_kind == FRL_SIM_L2) { const float df = av - bv; s0 = fmaf(df, df, s0); }
      else { s0 = fmaf(av, bv, s0); s1 = fmaf(av, av, s1); s2 = fmaf(bv, bv, s2); }
    }
    s0 = group16_sum(s0);
    float sim;
    if (sim_kind == FRL_SIM_L2) sim = -s0 / (float)D;
    else if (sim_kind == FRL_SIM_DOT) sim = s0;
    else {                                                        // F.normalize: x / max(|x|, 1e-12)
      s1 = group16_sum(s1);
      s2 = group16_sum(s2);
      sim = s0 / (fmaxf(sqrtf(s1), 1e-12f) * fmaxf(sqrtf(s2), 1e-12f));
    }
    if (l == 0) {
      sims[p] = sim;
      logits[p] = logf(weights ? weights[p] : 1.f) + sim * inv_t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// per anchor segment [seg[a], seg[a+1]) of the anchor-sorted pair list (is_pos marks the positives): one wave per segment.
// loss_a = -log(pos_sum + eps) + log(all_sum + eps)  with sums of exp(logit - max); coef[i] = d loss_a / d logit_i (max included).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_segments_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ is_pos,
                                                               const int64_t* __restrict__ seg, int64_t nseg, float* __restrict__ loss_a,
                                                               float* __restrict__ coef) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t a = (int64_t)blockIdx.x * 4 + wave; a < nseg; a += (int64_t)gridDim.x * 4) {
    const int64_t lo = seg[a], hi = seg[a + 1];
    float m = -INFINITY;
    for (int64_t i = lo + lane; i < hi; i += 64) m = fmaxf(m, logits[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // fixed-order sums: lane-strided partials, then the butterfly (same association every run)
    float sa = 0.f, sp = 0.f;
    for (int64_t i = lo + lane; i < hi; i += 64) {
      const float e = expf(logits[i] - m);
      sa += e;
      if (is_pos[i]) sp += e;
    }
    sa = wave_sum(sa);
    sp = wave_sum(sp);
    const float eps = 1e-8f;
    if (lane == 0) loss_a[a] = -logf(sp + eps) + logf(sa + eps);
    if (coef != nullptr) {
      // d loss_a / d logit_i through the exponentials, plus the path through the maximum itself: the reference's amax is differentiable
      // and with eps inside the logarithms the +-m terms leave  sp / (sp + eps) - sa / (sa + eps)  on the arg-max logit(s) (shared
      // evenly between tied maxima, as torch's amax backward does) -- 1e-3 of the gradient when the positives sit far below the maximum
      const float ia = 1.f / (sa + eps), ip = 1.f / (sp + eps);
      int nmax = 0;
      for (int64_t i = lo + lane; i < hi; i += 64) nmax += logits[i] == m ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) nmax += __shfl_xor(nmax, o, 64);
      const float gmax = eps * (ia - ip) / (float)(nmax > 0 ? nmax : 1);      // = sp/(sp+eps) - sa/(sa+eps) without the cancellation
      for (int64_t i = lo + lane; i < hi; i += 64) {
        const float e = expf(logits[i] - m);
        coef[i] = e * ia - (is_pos[i] ? e * ip : 0.f) + (logits[i] == m ? gmax : 0.f);
      }
    }
  }
}

// mean of n floats in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = n > 0 ? (float)(red[0] / (double)n) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// per pair gradient rows: ga[p][:] = g_p * d sim / d a, gb[p][:] = g_p * d sim / d b, g_p = gscale * coef[p] / t
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_grad_rows_kernel(const float* __restrict__ emb, int D, const int64_t* __restrict__ pairs,
                                                             const float* __restrict__ sims, const float* __restrict__ coef,
                                                             const float* __restrict__ gscale, float scale, int64_t T, int sim_kind,
                                                             float* __restrict__ ga, float* __restrict__ gb) {
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  const float gs = scale * (gscale ? gscale[0] : 1.f);
  for (int64_t p = (int64_t)blockIdx.x * 16 + g; p < T; p += (int64_t)gridDim.x * 16) {
    const float* a = emb + pairs[2 * p] * (int64_t)D;
    const float* b = emb + pairs[2 * p + 1] * (int64_t)D;
    const float gp = gs * coef[p];
    float na = 1.f, nb = 1.f, sim = 0.f;
    if (sim_kind == FRL_SIM_COSINE) {
      float s1 = 0.f, s2 = 0.f;
      for (int j = l; j < D; j += 16) { s1 = fmaf(a[j], a[j], s1); s2 = fmaf(b[j], b[j], s2); }
      na = fmaxf(sqrtf(group16_sum(s1)), 1e-12f);
      nb = fmaxf(sqrtf(group16_sum(s2)), 1e-12f);
      sim = sims[p];
    }
    for (int j = l; j < D; j += 16) {
      const float av = a[j], bv = b[j];
      float da, db;
      if (sim_kind == FRL_SIM_L2) { da = -2.f * (av - bv) / (float)D; db = -da; }
      else if (sim_kind == FRL_SIM_DOT) { da = bv; db = av; }
      else { da = (bv / nb - sim * av / na) / na; db = (av / na - sim * bv / nb) / nb; }
      ga[p * (int64_t)D + j] = gp * da;
      gb[p * (int64_t)D + j] = gp * db;
    }
  }
}

static unsigned ct_grid(int64_t work, int per_block) {
  int64_t g = (work + per_block - 1) / per_block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" {

// feat: strided raster (element strides sC, sH, sW), coords [N][2] int64 (row, col; negative = from the end) -> out [N][C] contiguous
int frl_gather_locations_fwd(const void* feat, int64_t sC, int64_t sH, int64_t sW, int C, int H, int W, const int64_t* coords, int64_t N,
                             void* out, int dtype, hipStream_t stream) {
  if (N <= 0 || C <= 0) return 0;
  if (dtype == FRL_F32)
    FRL_LAUNCH((gather_locations_kernel<float>), dim3(ct_grid(N * C, 256)), dim3(256), 0, stream, (const float*)feat, sC, sH, sW, C, H, W, coords, N,
               (float*)out);
  else if (dtype == FRL_BF16)
    FRL_LAUNCH((gather_locations_kernel<bf16>), dim3(ct_grid(N * C, 256)), dim3(256), 0, stream, (const bf16*)feat, sC, sH, sW, C, H, W, coords, N,
               (bf16*)out);
  else return frl_fail(-2, "gather_locations: dtype must be FRL_F32 or FRL_BF16");
  return frl_check_launch("gather_locations");
}

// out[key][0..D) (+)= sum of vals[order[i]][:] over each run of equal keys_sorted[i] (order may be NULL = identity); rows of `out`
// whose key does not occur are left untouched (zero them first unless accumulate)
int frl_segment_sum_rows(const float* vals, const int64_t* order, const int64_t* keys_sorted, int64_t M, int D, float* out, int64_t out_stride,
                         int accumulate, hipStream_t stream) {
  if (M <= 0 || D <= 0) return 0;
  FRL_LAUNCH(segment_sum_rows_kernel, dim3(ct_grid(M * D, 256)), dim3(256), 0, stream, vals, order, keys_sorted, M, D, out, out_stride, accumulate);
  return frl_check_launch("segment_sum_rows");
}

// Forward over T pairs sorted by anchor: pairs [T][2] int64 (rows of emb [.][D] f32), weights [T] or NULL, is_pos [T] bytes, seg [nseg + 1]
// segment bounds.  Outputs: sims [T], logits [T], loss_a [nseg], coef [T] (d loss_a / d logit; NULL to skip), loss [1] = mean(loss_a).
int frl_infonce_fwd(const float* emb, int D, const int64_t* pairs, const float* weights, const unsigned char* is_pos, int64_t T,
                    const int64_t* seg, int64_t nseg, float temperature, int similarity, float* sims, float* logits, float* loss_a, float* coef,
                    float* loss, hipStream_t stream) {
  if (T <= 0 || nseg <= 0 || D <= 0) return frl_fail(-2, "infonce: empty input");
  if (!(temperature > 0.f)) return frl_fail(-2, "infonce: temperature must be positive");
  if (similarity < 0 || similarity > 2) return frl_fail(-2, "infonce: similarity must be 0 (l2), 1 (cosine) or 2 (dot)");
  FRL_LAUNCH(pair_logits_kernel, dim3(ct_grid(T, 16)), dim3(256), 0, stream, emb, D, pairs, weights, T, 1.f / temperature, similarity, sims, logits);
  FRL_LAUNCH(infonce_segments_kernel, dim3(ct_grid(nseg, 4)), dim3(256), 0, stream, (const float*)logits, is_pos, seg, nseg, loss_a, coef);
  FRL_LAUNCH(mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)loss_a, nseg, loss);
  return frl_check_launch("infonce_fwd");
}

// Gradient rows of every pair: ga / gb [T][D] = gscale[0] / (nseg * temperature) * coef[p] * d sim / d a (resp. d b); the caller folds them
// into the embedding gradient with frl_segment_sum_rows (keys: anchors, then targets).
int frl_infonce_pair_grads(const float* emb, int D, const int64_t* pairs, const float* sims, const float* coef, const float* gscale, int64_t T,
                           int64_t nseg, float temperature, int similarity, float* ga, float* gb, hipStream_t stream) {
  if (T <= 0 || nseg <= 0 || D <= 0) return frl_fail(-2, "infonce: empty input");
  FRL_LAUNCH(pair_grad_rows_kernel, dim3(ct_grid(T, 16)), dim3(256), 0, stream, emb, D, pairs, sims, coef, gscale,
             1.f / ((float)nseg * temperature), T, similarity, ga, gb);
  return frl_check_launch("infonce_pair_grads");
}

}  // extern "C"
