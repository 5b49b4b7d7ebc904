// Fused global-norm gradient clipping + AdamW over a table of (scattered) parameter tensors: two launches per train step
// instead of the ~45 small foreach / elementwise kernels of clip_grad_norm_ + torch.optim.AdamW.
// Reference step order: frl/training/representation/step.py:1081-1087 (clip_grad_norm_(max_norm=1.0) then optimizer.step()),
// optimizer wiring scripts/train_vqvae.py:221-228 (AdamW, codebook group without weight decay).
//   launch 1  frl_grad_sqnorm_kernel : per-workgroup float64 partial sums of g^2 (fixed chunk -> workgroup map, fixed order)
//   launch 2  frl_adamw_kernel       : every workgroup re-sums the partials in the same order (bit-identical clip factor
//                                      everywhere), then  g *= min(1, max_norm / (norm + 1e-6));  decoupled weight decay;
//                                      m, v update;  p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)   (torch.optim.AdamW)
// A chunk is (tensor, 4096-element window); the host builds the chunk table once per parameter set.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include <math.h>

struct FrlParamDesc {      // one per tensor, device-resident table
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float weight_decay;
  int lag;                 // updates this tensor has skipped (no gradient): torch.optim.AdamW counts steps per parameter
};

#define OPT_CHUNK 4096
#define OPT_BATCH 72       // parameter records per launch: 72 x 48 B = 3456 B of kernel arguments (tables travel BY VALUE, so a
                           // step whose gradient buffers moved needs no host->device copy and no synchronisation)
struct FrlParamBatch { FrlParamDesc d[OPT_BATCH]; };

// counters [2] (device, optional): {updates applied, updates skipped}.  With an `ok` flag the whole update is conditional ON THE DEVICE
// (the reference's isfinite guard, step.py:1057-1074, without a host synchronisation): ok[0] <= 0 -> parameters, moments and the update
// count stay untouched and counters[1] is incremented.
// (count_ok / counters: the FIRST squared-norm launch of a step also bumps the update counters -- applied or skipped -- so that the AdamW
// launches behind it read the new count and no separate one-thread launch is needed)
__global__ __launch_bounds__(256) void frl_grad_sqnorm_kernel(const FrlParamBatch tab, int tbase, const int2* __restrict__ chunks,
                                                              int nchunks, double* __restrict__ partial, const float* __restrict__ count_ok,
                                                              int* __restrict__ counters) {
  __shared__ double red[4];
  if (counters != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    const bool go = (count_ok == nullptr) || (count_ok[0] > 0.f);
    counters[go ? 0 : 1] += 1;
  }
  double s = 0.0;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int2 ck = chunks[c];
    const FrlParamDesc& d = tab.d[ck.x - tbase];
    const int64_t lo = (int64_t)ck.y * OPT_CHUNK;
    const int64_t hi = lo + OPT_CHUNK < d.n ? lo + OPT_CHUNK : d.n;
    float a = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float g = d.g[i]; a = fmaf(g, g, a); }
    s += (double)a;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void frl_adamw_kernel(const FrlParamBatch tab, int tbase, const int2* __restrict__ chunks, int nchunks,
                                                        const double* __restrict__ partial, int npartial, float max_norm, float lr,
                                                        double beta1d, double beta2d, float eps, int step_host,
                                                        float* __restrict__ norm_out, const float* __restrict__ ok, int* counters,
                                                        int last_batch, const float* __restrict__ lr_dev) {
  __shared__ float coef_s;
  if (lr_dev != nullptr) lr = lr_dev[0];                     // learning rate kept on the device (a captured graph replays with new values)
  const bool go = (ok == nullptr) || (ok[0] > 0.f);
  // update number: device counter + 1 when the caller keeps one (exact under skipped batches), else the host's count
  const int step = (counters != nullptr) ? counters[0] : step_host;   // (already bumped by the step's first frl_grad_sqnorm_kernel launch)
  (void)last_batch;
  if (!go) return;
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int i = threadIdx.x; i < npartial; i += 64) s += partial[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(s);
      float coef = max_norm / (norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
      if (max_norm <= 0.f) coef = 1.f;                     // clipping disabled
      coef_s = coef;
      if (blockIdx.x == 0 && norm_out != nullptr) norm_out[0] = norm;
    }
  }
  __syncthreads();
  const float coef = coef_s;
  const float beta1 = (float)beta1d, beta2 = (float)beta2d;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int2 ck = chunks[c];
    const FrlParamDesc& d = tab.d[ck.x - tbase];
    // bias corrections in float64 from this tensor's own update count, as the Python scalars of torch.optim.AdamW
    const double tstep = (double)(step - d.lag);
    const float step_size = lr / (float)(1.0 - pow(beta1d, tstep));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2d, tstep));
    const int64_t lo = (int64_t)ck.y * OPT_CHUNK;
    const int64_t hi = lo + OPT_CHUNK < d.n ? lo + OPT_CHUNK : d.n;
    const float decay = 1.f - lr * d.weight_decay;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const float g = d.g[i] * coef;
      float p = d.p[i] * decay;
      const float m = fmaf(beta1, d.m[i], (1.f - beta1) * g);       // lerp(m, g, 1 - beta1)
      const float v = fmaf(beta2, d.v[i], (1.f - beta2) * g * g);
      const float denom = sqrtf(v) / bc2_sqrt + eps;
      p -= step_size * (m / denom);
      d.m[i] = m;
      d.v[i] = v;
      d.p[i] = p;
    }
  }
}

// dst[i] = scale * src[i] for a table of (src, dst, n) tensors: flattens scattered gradients into an all-reduce bucket in ONE launch
struct FrlCopyDesc { const float* src; float* dst; int64_t n; };
#define OPT_CBATCH 144
struct FrlCopyBatch { FrlCopyDesc d[OPT_CBATCH]; };

__global__ __launch_bounds__(256) void frl_multi_copy_kernel(const FrlCopyBatch tab, int tbase, const int2* __restrict__ chunks, int nchunks,
                                                             float scale) {
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int2 ck = chunks[c];
    const FrlCopyDesc& d = tab.d[ck.x - tbase];
    const int64_t lo = (int64_t)ck.y * OPT_CHUNK;
    const int64_t hi = lo + OPT_CHUNK < d.n ? lo + OPT_CHUNK : d.n;
    if (d.src == nullptr) {
      for (int64_t i = lo + threadIdx.x; i < hi; i += 256) d.dst[i] = 0.f;      // parameter without a gradient this step
    } else {
      for (int64_t i = lo + threadIdx.x; i < hi; i += 256) d.dst[i] = scale * d.src[i];
    }
  }
}

// Splits the chunk table (sorted by tensor, host copy in chunk_tensor) into runs that reference at most `batch` tensors.
static int opt_batch_end(const int* chunk_tensor, int nchunks, int c0, int t0, int batch) {
  int c = c0;
  while (c < nchunks && chunk_tensor[c] < t0 + batch) ++c;
  return c;
}

extern "C" {

// desc [ntensors] = {src, dst, n} is a HOST table; chunks [nchunks] = {tensor index, 4096-element window} a DEVICE table sorted by
// tensor, chunk_tensor [nchunks] its tensor column on the host.
int frl_multi_tensor_scale_copy(const void* desc_host, int ntensors, const void* chunks, const int* chunk_tensor, int nchunks, float scale,
                                hipStream_t stream) {
  if (ntensors <= 0 || nchunks <= 0) return frl_fail(-2, "multi_tensor_scale_copy: empty table");
  const FrlCopyDesc* dh = (const FrlCopyDesc*)desc_host;
  for (int t0 = 0, c0 = 0; t0 < ntensors; t0 += OPT_CBATCH) {
    FrlCopyBatch tab;
    const int nt = (ntensors - t0) < OPT_CBATCH ? (ntensors - t0) : OPT_CBATCH;
    for (int i = 0; i < nt; ++i) tab.d[i] = dh[t0 + i];
    const int c1 = opt_batch_end(chunk_tensor, nchunks, c0, t0, OPT_CBATCH);
    if (c1 > c0) {
      const int grid = (c1 - c0) < 512 ? (c1 - c0) : 512;
      FRL_LAUNCH(frl_multi_copy_kernel, dim3(grid), dim3(256), 0, stream, tab, t0, (const int2*)chunks + c0, c1 - c0, scale);
    }
    c0 = c1;
  }
  return frl_check_launch("multi_tensor_scale_copy");
}

size_t frl_adamw_workspace_bytes(void) { return 4096 * sizeof(double); }

// desc [ntensors] is a HOST table of 48-byte records; chunks / chunk_tensor as above; step is the 1-based update count.
// norm_out (device, optional) receives the pre-clip global gradient norm.  max_norm <= 0 disables clipping.
int frl_adamw_clip_step(const void* desc_host, int ntensors, const void* chunks, const int* chunk_tensor, int nchunks, float max_norm,
                        float lr, double beta1, double beta2, float eps, int step, float* norm_out, const float* ok, int* counters,
                        const float* lr_dev, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (ntensors <= 0 || nchunks <= 0) return frl_fail(-2, "adamw: empty parameter table");
  if (step < 1 && counters == nullptr) return frl_fail(-2, "adamw: step must be >= 1");
  if (ws == nullptr || ws_bytes < frl_adamw_workspace_bytes()) return frl_fail(-4, "adamw: workspace too small");
  const int nbatch = (ntensors + OPT_BATCH - 1) / OPT_BATCH;
  if (nbatch * 512 > 4096) return frl_fail(-2, "adamw: more than 576 parameter tensors per call");
  const FrlParamDesc* dh = (const FrlParamDesc*)desc_host;
  double* partial = (double*)ws;
  int npartial = 0;
  for (int pass = 0; pass < 2; ++pass) {                      // pass 0: partial sums of g^2 of every batch; pass 1: updates
    int poff = 0;
    for (int t0 = 0, c0 = 0; t0 < ntensors; t0 += OPT_BATCH) {
      FrlParamBatch tab;
      const int nt = (ntensors - t0) < OPT_BATCH ? (ntensors - t0) : OPT_BATCH;
      for (int i = 0; i < nt; ++i) tab.d[i] = dh[t0 + i];
      const int c1 = opt_batch_end(chunk_tensor, nchunks, c0, t0, OPT_BATCH);
      if (c1 > c0) {
        const int grid = (c1 - c0) < 512 ? (c1 - c0) : 512;
        if (pass == 0) {
          FRL_LAUNCH(frl_grad_sqnorm_kernel, dim3(grid), dim3(256), 0, stream, tab, t0, (const int2*)chunks + c0, c1 - c0, partial + poff, ok,
                     poff == 0 ? counters : (int*)nullptr);
          poff += grid;
        } else {
          FRL_LAUNCH(frl_adamw_kernel, dim3(grid), dim3(256), 0, stream, tab, t0, (const int2*)chunks + c0, c1 - c0, (const double*)partial,
                     npartial, max_norm, lr, beta1, beta2, eps, step, norm_out, ok, counters, (t0 + OPT_BATCH >= ntensors) ? 1 : 0, lr_dev);
        }
      }
      c0 = c1;
    }
    if (pass == 0) npartial = poff;
  }
  return frl_check_launch("adamw_clip_step");
}

}  // extern "C"
