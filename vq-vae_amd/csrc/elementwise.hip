// Streaming (HBM-bound) elementwise / reduction kernels of the hot path:
//   * masked L2 reconstruction loss fwd/bwd  -- frl/losses/reconstruction.py:95-139 (loss_type 'l2', mean over valid)
//   * FiLM modulation z = gamma * h + beta broadcast over time -- frl/models/representation.py:369-372
//   * gate blend out = smoothed + clamp(gate) * residual       -- frl/models/spatial.py:333-335
//   * time mean of a (time,y,x,feature) tile (type-path input, SURVEY.md section 8 preamble)
// 16-byte accesses per lane, grid-stride, fixed-order reductions.
#include "frl_common.hpp"
#include "frl_host.hpp"

#define EW_GRID_MAX 2048

// ---------------------------------------------------------------- MSE
template <typename T, int V>
__global__ __launch_bounds__(256) void mse_partial_kernel(const T* __restrict__ pred, const T* __restrict__ tgt,
                                                          const uint8_t* __restrict__ mask, int64_t P, int C,
                                                          double* __restrict__ partial /*[grid][2]*/) {
  const int vpr = C / V;
  const int64_t total = P * vpr;
  float s = 0.f, cnt = 0.f;
  double sd = 0.0, cd = 0.0;
  int it = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / vpr;
    if (mask != nullptr && !mask[row]) continue;
    float a[V], b[V];
    if constexpr (V == 1) { a[0] = to_f32(pred[i]); b[0] = to_f32(tgt[i]); }
    else { Vec<T>::load(pred + i * V, a); Vec<T>::load(tgt + i * V, b); }
#pragma unroll
    for (int e = 0; e < V; ++e) { const float d = a[e] - b[e]; s = fmaf(d, d, s); }
    cnt += (float)V;
    if (++it == 64) { sd += (double)s; cd += (double)cnt; s = 0.f; cnt = 0.f; it = 0; }
  }
  sd += (double)s; cd += (double)cnt;
  sd = wave_sum_d(sd); cd = wave_sum_d(cd);
  __shared__ double red[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = sd; red[4 + wave] = cd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2 + 0] = red[0] + red[1] + red[2] + red[3];
    partial[blockIdx.x * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(256) void mse_finalize_kernel(const double* __restrict__ partial, int n, float* __restrict__ out /*{mean, n_valid}*/) {
  __shared__ double rs[256], rc[256];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  rs[threadIdx.x] = s; rc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = rc[0] > 0.0 ? (float)(rs[0] / rc[0]) : 0.f;
    out[1] = (float)rc[0];
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const T* __restrict__ pred, const T* __restrict__ tgt,
                                                      const uint8_t* __restrict__ mask, const float* __restrict__ gscale,
                                                      const float* __restrict__ stats, int64_t P, int C, T* __restrict__ dpred) {
  const int vpr = C / V;
  const int64_t total = P * vpr;
  const float nv = stats[1];
  const float k = nv > 0.f ? (gscale ? gscale[0] : 1.f) * 2.f / nv : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / vpr;
    const bool ok = (mask == nullptr) || mask[row];
    float a[V], b[V], o[V];
    if constexpr (V == 1) { a[0] = to_f32(pred[i]); b[0] = to_f32(tgt[i]); }
    else { Vec<T>::load(pred + i * V, a); Vec<T>::load(tgt + i * V, b); }
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = ok ? k * (a[e] - b[e]) : 0.f;
    if constexpr (V == 1) dpred[i] = from_f32<T>(o[0]); else Vec<T>::store(dpred + i * V, o);
  }
}

// ---------------------------------------------------------------- FiLM modulate: out[b,t,p,c] = g[b,p,c]*h[b,t,p,c] + be[b,p,c]
template <typename T, int V>
__global__ __launch_bounds__(256) void film_fwd_kernel(const T* __restrict__ h, const T* __restrict__ g, const T* __restrict__ be,
                                                       T* __restrict__ out, int64_t B, int Tn, int64_t HWC /*elements*/) {
  const int64_t nv = HWC / V, total = B * Tn * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t j = i % nv, bt = i / nv, b = bt / Tn;
    float hv[V], gv[V], bv[V], o[V];
    if constexpr (V == 1) { hv[0] = to_f32(h[i]); gv[0] = to_f32(g[b * nv + j]); bv[0] = to_f32(be[b * nv + j]); }
    else { Vec<T>::load(h + i * V, hv); Vec<T>::load(g + (b * nv + j) * V, gv); Vec<T>::load(be + (b * nv + j) * V, bv); }
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = fmaf(gv[e], hv[e], bv[e]);
    if constexpr (V == 1) out[i] = from_f32<T>(o[0]); else Vec<T>::store(out + i * V, o);
  }
}

// dh = dout * g ; dg = sum_t dout * h ; dbe = sum_t dout     (thread owns a (b, p, c-vector), loops over t)
template <typename T, int V>
__global__ __launch_bounds__(256) void film_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ h, const T* __restrict__ g,
                                                       T* __restrict__ dh, T* __restrict__ dg, T* __restrict__ dbe, int64_t B, int Tn,
                                                       int64_t HWC) {
  const int64_t nv = HWC / V, total = B * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t j = i % nv, b = i / nv;
    float gv[V], sg[V], sb[V];
    if constexpr (V == 1) gv[0] = to_f32(g[i]); else Vec<T>::load(g + i * V, gv);
#pragma unroll
    for (int e = 0; e < V; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
    for (int t = 0; t < Tn; ++t) {
      const int64_t k = (b * Tn + t) * nv + j;
      float dv[V], hv[V], o[V];
      if constexpr (V == 1) { dv[0] = to_f32(dout[k]); hv[0] = to_f32(h[k]); }
      else { Vec<T>::load(dout + k * V, dv); Vec<T>::load(h + k * V, hv); }
#pragma unroll
      for (int e = 0; e < V; ++e) { o[e] = dv[e] * gv[e]; sg[e] = fmaf(dv[e], hv[e], sg[e]); sb[e] += dv[e]; }
      if constexpr (V == 1) dh[k] = from_f32<T>(o[0]); else Vec<T>::store(dh + k * V, o);
    }
    if constexpr (V == 1) { dg[i] = from_f32<T>(sg[0]); dbe[i] = from_f32<T>(sb[0]); }
    else { Vec<T>::store(dg + i * V, sg); Vec<T>::store(dbe + i * V, sb); }
  }
}

// ---------------------------------------------------------------- gate blend
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_blend_fwd_kernel(const T* __restrict__ sm, const T* __restrict__ res, const T* __restrict__ graw,
                                                             float min_gate, T* __restrict__ out, T* __restrict__ gate, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float s[V], r[V], g[V], o[V];
    Vec<T>::load(sm + i * V, s); Vec<T>::load(res + i * V, r); Vec<T>::load(graw + i * V, g);
#pragma unroll
    for (int e = 0; e < V; ++e) { if (min_gate > 0.f && g[e] < min_gate) g[e] = min_gate; o[e] = fmaf(g[e], r[e], s[e]); }
    Vec<T>::store(out + i * V, o);
    Vec<T>::store(gate + i * V, g);
  }
}

// d_res = dout*gate ; d_graw = (dout*res + dgate_ext) * [graw >= min_gate or min_gate <= 0] ; (d_smoothed = dout, not written)
template <typename T, int V>
__global__ __launch_bounds__(256) void gate_blend_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ dgate_ext, const T* __restrict__ res,
                                                             const T* __restrict__ graw, float min_gate, T* __restrict__ dres,
                                                             T* __restrict__ dgraw, int64_t nvec, int sigmoid_mask) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float d[V], r[V], g[V], x[V], o1[V], o2[V];
    Vec<T>::load(dout + i * V, d); Vec<T>::load(res + i * V, r); Vec<T>::load(graw + i * V, g);
    if (dgate_ext != nullptr) Vec<T>::load(dgate_ext + i * V, x);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) x[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const bool clamped = (min_gate > 0.f && g[e] < min_gate);
      const float ge = clamped ? min_gate : g[e];
      o1[e] = d[e] * ge;
      o2[e] = clamped ? 0.f : fmaf(d[e], r[e], x[e]);
      if (sigmoid_mask) o2[e] *= g[e] * (1.f - g[e]);            // graw = sigmoid(.): hand on the gradient w.r.t. the pre-activation
    }
    Vec<T>::store(dres + i * V, o1);
    Vec<T>::store(dgraw + i * V, o2);
  }
}

// ---------------------------------------------------------------- time mean: out[b,p] = mean_t tile[b,t,p]
template <typename T, int V>
__global__ __launch_bounds__(256) void mean_time_kernel(const T* __restrict__ tile, T* __restrict__ out, int64_t B, int Tn, int64_t HWC) {
  const int64_t nv = HWC / V, total = B * nv;
  const float inv = 1.f / (float)Tn;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t j, b;
    if (total < ((int64_t)1 << 31)) { const unsigned u = (unsigned)i, q = u / (unsigned)nv; b = q; j = u - q * (unsigned)nv; }   // (one 32-bit division, not two 64-bit ones)
    else { j = i % nv; b = i / nv; }
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    for (int t = 0; t < Tn; ++t) {
      float v[V];
      Vec<T>::load(tile + ((b * Tn + t) * nv + j) * V, v);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] *= inv;
    Vec<T>::store(out + i * V, s);
  }
}

// out = a + scale_b * b (gradient accumulation / subtraction for multi-use tensors)
template <typename T, int V>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, float sb, T* __restrict__ out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float x[V], y[V];
    Vec<T>::load(a + i * V, x); Vec<T>::load(b + i * V, y);
#pragma unroll
    for (int e = 0; e < V; ++e) x[e] = fmaf(sb, y[e], x[e]);
    Vec<T>::store(out + i * V, x);
  }
}

static inline unsigned ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > EW_GRID_MAX) g = EW_GRID_MAX;
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define EW_DISPATCH(DTYPE, VECOK, CALL_F32V, CALL_F32S, CALL_BF16V, CALL_BF16S) \
  if ((DTYPE) == FRL_F32) { if (VECOK(4)) { CALL_F32V; } else { CALL_F32S; } }  \
  else if ((DTYPE) == FRL_BF16) { if (VECOK(8)) { CALL_BF16V; } else { CALL_BF16S; } } \
  else return frl_fail(-2, "bad dtype");

// m[i] (optional): a device scalar that multiplies coef[i] -- a weight that follows a per-step schedule (lambda_vq(step)) without being
// baked into a captured graph as a kernel argument
struct ScalarTerms { const float* p[8]; const float* m[8]; float c[8]; float a[8]; };
// aux_out (optional): a second, un-scheduled combination aux = sum_i a[i] * term_i of the same terms (a reported sub-total such as
// vq_loss = L_codebook + beta L_commit, which would otherwise be a launch of its own in the forward and one in the backward)
__global__ void scalar_combine_kernel(ScalarTerms t, int n, float* __restrict__ out, float* __restrict__ ok_out, float* __restrict__ aux_out) {
  float s = 0.f, a = 0.f;
  for (int i = 0; i < n; ++i) {
    const float v = *t.p[i];
    s = fmaf(t.m[i] != nullptr ? t.c[i] * *t.m[i] : t.c[i], v, s);
    a = fmaf(t.a[i], v, a);
  }
  out[0] = s;
  if (ok_out != nullptr) ok_out[0] = (s * 0.f == 0.f) ? 1.f : 0.f;   // x * 0 == 0 holds exactly for finite x
  if (aux_out != nullptr) aux_out[0] = a;
}
__global__ void scalar_fanout_kernel(const float* __restrict__ g, ScalarTerms t, int n, float* __restrict__ out) {
  if ((int)threadIdx.x < n) {
    const int i = threadIdx.x;
    out[i] = g[0] * (t.m[i] != nullptr ? t.c[i] * *t.m[i] : t.c[i]);
  }
}

extern "C" {
int frl_scalar_combine_dev(const float* const* terms_host, const float* coef_host, const float* const* mult_host, int n, float* out, float* ok_out,
                           hipStream_t stream);
int frl_scalar_combine_aux(const float* const* terms_host, const float* coef_host, const float* const* mult_host, const float* aux_coef_host, int n,
                           float* out, float* ok_out, float* aux_out, hipStream_t stream);
int frl_scalar_fanout_dev(const float* g, const float* coef_host, const float* const* mult_host, int n, float* out, hipStream_t stream);

size_t frl_mse_workspace_bytes(void) { return (size_t)EW_GRID_MAX * 2 * sizeof(double); }

// pred/target [P][C]; mask [P] uint8 (1 = valid) or null; out = {mean over valid elements, n_valid_elements}
int frl_mse_fwd(const void* pred, const void* target, const uint8_t* mask, int64_t P, int C, float* out, int dtype, void* ws,
                size_t ws_bytes, hipStream_t stream) {
  if (ws_bytes < frl_mse_workspace_bytes()) return frl_fail(-4, "mse: workspace too small");
  if (P <= 0) return frl_fail(-2, "mse: empty input");
  double* partial = (double*)ws;
#define VOK(v) (C % (v) == 0)
  unsigned grid;
  EW_DISPATCH(dtype, VOK,
    grid = ew_grid(P * (C / 4)); FRL_LAUNCH((mse_partial_kernel<float, 4>), dim3(grid), dim3(256), 0, stream, (const float*)pred, (const float*)target, mask, P, C, partial),
    grid = ew_grid(P * C); FRL_LAUNCH((mse_partial_kernel<float, 1>), dim3(grid), dim3(256), 0, stream, (const float*)pred, (const float*)target, mask, P, C, partial),
    grid = ew_grid(P * (C / 8)); FRL_LAUNCH((mse_partial_kernel<bf16, 8>), dim3(grid), dim3(256), 0, stream, (const bf16*)pred, (const bf16*)target, mask, P, C, partial),
    grid = ew_grid(P * C); FRL_LAUNCH((mse_partial_kernel<bf16, 1>), dim3(grid), dim3(256), 0, stream, (const bf16*)pred, (const bf16*)target, mask, P, C, partial))
  FRL_LAUNCH(mse_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double*)partial, (int)grid, out);
  return frl_check_launch("mse_fwd");
}

// dpred = gscale * 2 / n_valid * (pred - target) on valid rows; stats = frl_mse_fwd's out
int frl_mse_bwd(const void* pred, const void* target, const uint8_t* mask, const float* gscale, const float* stats, int64_t P,
                int C, void* dpred, int dtype, hipStream_t stream) {
  EW_DISPATCH(dtype, VOK,
    FRL_LAUNCH((mse_bwd_kernel<float, 4>), dim3(ew_grid(P * (C / 4))), dim3(256), 0, stream, (const float*)pred, (const float*)target, mask, gscale, stats, P, C, (float*)dpred),
    FRL_LAUNCH((mse_bwd_kernel<float, 1>), dim3(ew_grid(P * C)), dim3(256), 0, stream, (const float*)pred, (const float*)target, mask, gscale, stats, P, C, (float*)dpred),
    FRL_LAUNCH((mse_bwd_kernel<bf16, 8>), dim3(ew_grid(P * (C / 8))), dim3(256), 0, stream, (const bf16*)pred, (const bf16*)target, mask, gscale, stats, P, C, (bf16*)dpred),
    FRL_LAUNCH((mse_bwd_kernel<bf16, 1>), dim3(ew_grid(P * C)), dim3(256), 0, stream, (const bf16*)pred, (const bf16*)target, mask, gscale, stats, P, C, (bf16*)dpred))
#undef VOK
  return frl_check_launch("mse_bwd");
}

// h,out [B][T][HW][C]; gamma,beta [B][HW][C]
int frl_film_modulate_fwd(const void* h, const void* gamma, const void* beta, void* out, int64_t B, int T, int64_t HW, int C,
                          int dtype, hipStream_t stream) {
  const int64_t hwc = HW * C;
#define VOK(v) (hwc % (v) == 0)
  EW_DISPATCH(dtype, VOK,
    FRL_LAUNCH((film_fwd_kernel<float, 4>), dim3(ew_grid(B * T * hwc / 4)), dim3(256), 0, stream, (const float*)h, (const float*)gamma, (const float*)beta, (float*)out, B, T, hwc),
    FRL_LAUNCH((film_fwd_kernel<float, 1>), dim3(ew_grid(B * T * hwc)), dim3(256), 0, stream, (const float*)h, (const float*)gamma, (const float*)beta, (float*)out, B, T, hwc),
    FRL_LAUNCH((film_fwd_kernel<bf16, 8>), dim3(ew_grid(B * T * hwc / 8)), dim3(256), 0, stream, (const bf16*)h, (const bf16*)gamma, (const bf16*)beta, (bf16*)out, B, T, hwc),
    FRL_LAUNCH((film_fwd_kernel<bf16, 1>), dim3(ew_grid(B * T * hwc)), dim3(256), 0, stream, (const bf16*)h, (const bf16*)gamma, (const bf16*)beta, (bf16*)out, B, T, hwc))
  return frl_check_launch("film_modulate_fwd");
}

int frl_film_modulate_bwd(const void* dout, const void* h, const void* gamma, void* dh, void* dgamma, void* dbeta, int64_t B, int T,
                          int64_t HW, int C, int dtype, hipStream_t stream) {
  const int64_t hwc = HW * C;
  EW_DISPATCH(dtype, VOK,
    FRL_LAUNCH((film_bwd_kernel<float, 4>), dim3(ew_grid(B * hwc / 4)), dim3(256), 0, stream, (const float*)dout, (const float*)h, (const float*)gamma, (float*)dh, (float*)dgamma, (float*)dbeta, B, T, hwc),
    FRL_LAUNCH((film_bwd_kernel<float, 1>), dim3(ew_grid(B * hwc)), dim3(256), 0, stream, (const float*)dout, (const float*)h, (const float*)gamma, (float*)dh, (float*)dgamma, (float*)dbeta, B, T, hwc),
    FRL_LAUNCH((film_bwd_kernel<bf16, 8>), dim3(ew_grid(B * hwc / 8)), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)h, (const bf16*)gamma, (bf16*)dh, (bf16*)dgamma, (bf16*)dbeta, B, T, hwc),
    FRL_LAUNCH((film_bwd_kernel<bf16, 1>), dim3(ew_grid(B * hwc)), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)h, (const bf16*)gamma, (bf16*)dh, (bf16*)dgamma, (bf16*)dbeta, B, T, hwc))
#undef VOK
  return frl_check_launch("film_modulate_bwd");
}

// n elements, n % 8 == 0 (bf16) / % 4 (f32) required: these tensors are [P][C] with C a multiple of 8
int frl_gate_blend_fwd(const void* smoothed, const void* residual, const void* gate_raw, float min_gate, void* out, void* gate_out,
                       int64_t n, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && n % 4 == 0)
    FRL_LAUNCH((gate_blend_fwd_kernel<float, 4>), dim3(ew_grid(n / 4)), dim3(256), 0, stream, (const float*)smoothed, (const float*)residual, (const float*)gate_raw, min_gate, (float*)out, (float*)gate_out, n / 4);
  else if (dtype == FRL_BF16 && n % 8 == 0)
    FRL_LAUNCH((gate_blend_fwd_kernel<bf16, 8>), dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)smoothed, (const bf16*)residual, (const bf16*)gate_raw, min_gate, (bf16*)out, (bf16*)gate_out, n / 8);
  else return frl_fail(-2, "gate_blend: element count must be a multiple of the 16-byte vector width");
  return frl_check_launch("gate_blend_fwd");
}

// sigmoid_mask != 0: d_gate_raw is returned multiplied by gate_raw (1 - gate_raw) -- the derivative of the sigmoid that produced gate_raw
// (gate_net of spatial.py:266-272) -- so that the convolution's backward calls take it without their own activation mask
int frl_gate_blend_bwd_masked(const void* dout, const void* dgate_ext, const void* residual, const void* gate_raw, float min_gate,
                              void* d_residual, void* d_gate_raw, int64_t n, int dtype, int sigmoid_mask, hipStream_t stream) {
  if (dtype == FRL_F32 && n % 4 == 0)
    FRL_LAUNCH((gate_blend_bwd_kernel<float, 4>), dim3(ew_grid(n / 4)), dim3(256), 0, stream, (const float*)dout, (const float*)dgate_ext, (const float*)residual, (const float*)gate_raw, min_gate, (float*)d_residual, (float*)d_gate_raw, n / 4, sigmoid_mask);
  else if (dtype == FRL_BF16 && n % 8 == 0)
    FRL_LAUNCH((gate_blend_bwd_kernel<bf16, 8>), dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)dout, (const bf16*)dgate_ext, (const bf16*)residual, (const bf16*)gate_raw, min_gate, (bf16*)d_residual, (bf16*)d_gate_raw, n / 8, sigmoid_mask);
  else return frl_fail(-2, "gate_blend: element count must be a multiple of the 16-byte vector width");
  return frl_check_launch("gate_blend_bwd");
}
int frl_gate_blend_bwd(const void* dout, const void* dgate_ext, const void* residual, const void* gate_raw, float min_gate,
                       void* d_residual, void* d_gate_raw, int64_t n, int dtype, hipStream_t stream) {
  return frl_gate_blend_bwd_masked(dout, dgate_ext, residual, gate_raw, min_gate, d_residual, d_gate_raw, n, dtype, 0, stream);
}

// tile [B][T][HWC] -> out [B][HWC] (mean over time)
int frl_mean_time_fwd(const void* tile, void* out, int64_t B, int T, int64_t HWC, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && HWC % 4 == 0)
    FRL_LAUNCH((mean_time_kernel<float, 4>), dim3(ew_grid(B * HWC / 4)), dim3(256), 0, stream, (const float*)tile, (float*)out, B, T, HWC);
  else if (dtype == FRL_BF16 && HWC % 8 == 0)
    FRL_LAUNCH((mean_time_kernel<bf16, 8>), dim3(ew_grid(B * HWC / 8)), dim3(256), 0, stream, (const bf16*)tile, (bf16*)out, B, T, HWC);
  else return frl_fail(-2, "mean_time: H*W*C must be a multiple of the 16-byte vector width");
  return frl_check_launch("mean_time_fwd");
}

int frl_add(const void* a, const void* b, float scale_b, void* out, int64_t n, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && n % 4 == 0)
    FRL_LAUNCH((add_kernel<float, 4>), dim3(ew_grid(n / 4)), dim3(256), 0, stream, (const float*)a, (const float*)b, scale_b, (float*)out, n / 4);
  else if (dtype == FRL_BF16 && n % 8 == 0)
    FRL_LAUNCH((add_kernel<bf16, 8>), dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, scale_b, (bf16*)out, n / 8);
  else return frl_fail(-2, "add: element count must be a multiple of the 16-byte vector width");
  return frl_check_launch("add");
}


// Loss head of the train step in one launch: out[0] = sum_i coef[i] * *terms[i]; ok_out[0] (optional) = 1 when out[0] is finite, else 0
// (the device-side isfinite guard of the trainer: step.py:1057-1074).  Replaces the chain of scalar mul / add / compare kernels that
// `lambda_recon * L + lambda_vq * (L_cb + beta * L_cm) + ...` costs as separate launches.  n <= 8; coef are host values.
int frl_scalar_combine(const float* const* terms_host, const float* coef_host, int n, float* out, float* ok_out, hipStream_t stream) {
  return frl_scalar_combine_dev(terms_host, coef_host, nullptr, n, out, ok_out, stream);
}
// The same with optional device multipliers: out[0] = sum_i coef[i] * (mult[i] ? *mult[i] : 1) * *terms[i]  (mult_host: array of n device
// pointers, entries or the array itself may be NULL).  A weight that changes every step is then a device word, not a kernel argument.
int frl_scalar_combine_dev(const float* const* terms_host, const float* coef_host, const float* const* mult_host, int n, float* out, float* ok_out,
                           hipStream_t stream) {
  return frl_scalar_combine_aux(terms_host, coef_host, mult_host, nullptr, n, out, ok_out, nullptr, stream);
}
// The same plus aux_out[0] = sum_i aux_coef[i] * *terms[i] (aux_coef_host / aux_out may be NULL together).
int frl_scalar_combine_aux(const float* const* terms_host, const float* coef_host, const float* const* mult_host, const float* aux_coef_host, int n,
                           float* out, float* ok_out, float* aux_out, hipStream_t stream) {
  if (n < 1 || n > 8) return frl_fail(-2, "scalar_combine: 1..8 terms");
  if ((aux_coef_host == nullptr) != (aux_out == nullptr)) return frl_fail(-2, "scalar_combine: aux coefficients and aux output come as a pair");
  ScalarTerms t;
  for (int i = 0; i < 8; ++i) {
    t.p[i] = i < n ? terms_host[i] : nullptr;
    t.m[i] = (i < n && mult_host != nullptr) ? mult_host[i] : nullptr;
    t.c[i] = i < n ? coef_host[i] : 0.f;
    t.a[i] = (i < n && aux_coef_host != nullptr) ? aux_coef_host[i] : 0.f;
  }
  FRL_LAUNCH(scalar_combine_kernel, dim3(1), dim3(1), 0, stream, t, n, out, ok_out, aux_out);
  return frl_check_launch("scalar_combine");
}
// Backward of frl_scalar_combine: out[i] = g[0] * coef[i] (* *mult[i]).
int frl_scalar_fanout(const float* g, const float* coef_host, int n, float* out, hipStream_t stream) {
  return frl_scalar_fanout_dev(g, coef_host, nullptr, n, out, stream);
}
int frl_scalar_fanout_dev(const float* g, const float* coef_host, const float* const* mult_host, int n, float* out, hipStream_t stream) {
  if (n < 1 || n > 8) return frl_fail(-2, "scalar_fanout: 1..8 terms");
  ScalarTerms t;
  for (int i = 0; i < 8; ++i) {
    t.p[i] = nullptr;
    t.m[i] = (i < n && mult_host != nullptr) ? mult_host[i] : nullptr;
    t.c[i] = i < n ? coef_host[i] : 0.f;
    t.a[i] = 0.f;
  }
  FRL_LAUNCH(scalar_fanout_kernel, dim3(1), dim3(8), 0, stream, g, t, n, out);
  return frl_check_launch("scalar_fanout");
}

}  // extern "C"
