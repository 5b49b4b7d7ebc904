// GroupNorm over NHWC rows, statistics per (sample, group) across (C/G channels x HW pixels), biased
// variance, eps inside the sqrt, affine -- nn.GroupNorm semantics of the reference encoder
// (frl/models/conv2d_encoder.py:117; stack order conv -> GN -> ReLU, last layer stops after GN :119-125).
// HBM-bound streaming kernels: stats pass (read x once), apply pass (read x, write y, optional fused ReLU).
// All reductions are fixed-order (no float atomics) so results are bit-reproducible.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_reduce.hpp"

// One workgroup (NTH = 1024 threads: 16 waves keep a sample's 128-256 KB stream in flight) per sample.  Thread layout: vpr = C / V
// channel-vectors per row, rpi = NTH / vpr rows per
// iteration; each thread owns a fixed channel vector and strides over rows.  MODE 0: sums of x and x^2;
// MODE 1 (backward): sums of dyh and dyh * xhat with dyh = dy * relu'(xhat*gamma+beta).
template <typename T, int V, int MODE, int NTH>
__global__ __launch_bounds__(NTH) void gn_reduce_kernel(const T* __restrict__ X, const T* __restrict__ DY,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int HW, int C, int G, float eps, int relu,
                                                        float* __restrict__ out_a /*[B][G] or [B][C]*/,
                                                        float* __restrict__ out_b, float* __restrict__ grp /*[B][G][2]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [2][rpi][C]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int vpr = C / V;
  const int rpi = NTH / vpr > 0 ? NTH / vpr : 1;
  const int cg = C / G;
  const T* xb = X + (int64_t)b * HW * C;
  float s0[V], s1[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  const int cv = tid % vpr, r0 = tid / vpr;
  const bool active = tid < vpr * rpi;
  if (active) {
    const int c0 = cv * V;
    float ga[V], be[V], mu[V], rs[V];
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e];
        mu[e] = mean[b * G + (c0 + e) / cg]; rs[e] = rstd[b * G + (c0 + e) / cg];
      }
    }
    for (int r = r0; r < HW; r += rpi) {
      float xv[V];
      if constexpr (V == 1) xv[0] = to_f32(xb[(int64_t)r * C + c0]); else Vec<T>::load(xb + (int64_t)r * C + c0, xv);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) { s0[e] += xv[e]; s1[e] = fmaf(xv[e], xv[e], s1[e]); }
      } else {
        float dv[V];
        const T* dyb = DY + (int64_t)b * HW * C;
        if constexpr (V == 1) dv[0] = to_f32(dyb[(int64_t)r * C + c0]); else Vec<T>::load(dyb + (int64_t)r * C + c0, dv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float xh = (xv[e] - mu[e]) * rs[e];
          float d = dv[e];
          if (relu && !(fmaf(xh, ga[e], be[e]) > 0.f)) d = 0.f;
          s0[e] += d; s1[e] = fmaf(d, xh, s1[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[(0 * rpi + r0) * C + c0 + e] = s0[e];
      red[(1 * rpi + r0) * C + c0 + e] = s1[e];
    }
  }
  __syncthreads();
  // per-channel totals (fixed order over rows)
  for (int i = tid; i < 2 * C; i += NTH) {
    const int which = i / C, c = i % C;
    float s = 0.f;
    for (int r = 0; r < rpi; ++r) s += red[(which * rpi + r) * C + c];
    red[(which * rpi) * C + c] = s;
  }
  __syncthreads();
  if (MODE == 0) {
    if (tid < G) {
      double a = 0.0, q = 0.0;
      for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += (double)red[c]; q += (double)red[rpi * C + c]; }
      const double n = (double)cg * HW;
      const double m = a / n;
      double var = q / n - m * m;
      if (var < 0.0) var = 0.0;
      out_a[b * G + tid] = (float)m;
      out_b[b * G + tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
  } else {
    for (int c = tid; c < C; c += NTH) {
      out_a[(int64_t)b * C + c] = red[c];              // sum dyh          -> d beta contribution
      out_b[(int64_t)b * C + c] = red[rpi * C + c];    // sum dyh * xhat   -> d gamma contribution
    }
    if (tid < G) {
      float S1 = 0.f, S2 = 0.f;
      for (int c = tid * cg; c < (tid + 1) * cg; ++c) { S1 = fmaf(gamma[c], red[c], S1); S2 = fmaf(gamma[c], red[rpi * C + c], S2); }
      grp[(b * G + tid) * 2 + 0] = S1;
      grp[(b * G + tid) * 2 + 1] = S2;
    }
  }
}

// y = act(xhat * gamma + beta).  One workgroup = one row chunk of ONE sample; a thread keeps a fixed channel vector, so the per-channel
// coefficients (rstd*gamma, beta - mean*rstd*gamma) are formed once and the row loop is load -> V FMAs -> store.
template <typename T, int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ X, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, T* __restrict__ Y, int rows_per_wg,
                                                       int HW, int C, int G, int relu) {
  const int vpr = C / V, cg = C / G;
  const int rpi = 256 / vpr;                       // rows per iteration (launcher guarantees vpr <= 256)
  const int b = blockIdx.y, cv = threadIdx.x % vpr, r0 = threadIdx.x / vpr;
  if (r0 >= rpi) return;
  const int c0 = cv * V;
  float a[V], o[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const int c = c0 + e, g = c / cg;
    a[e] = rstd[b * G + g] * gamma[c];
    o[e] = fmaf(-mean[b * G + g], a[e], beta[c]);
  }
  const int row_lo = blockIdx.x * rows_per_wg;
  const int row_hi = (row_lo + rows_per_wg) < HW ? (row_lo + rows_per_wg) : HW;
  const T* xb = X + (int64_t)b * HW * C + c0;
  T* yb = Y + (int64_t)b * HW * C + c0;
  for (int r = row_lo + r0; r < row_hi; r += rpi) {
    float xv[V], yv[V];
    if constexpr (V == 1) xv[0] = to_f32(xb[(int64_t)r * C]); else Vec<T>::load(xb + (int64_t)r * C, xv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = fmaf(xv[e], a[e], o[e]);
      if (relu) v = v > 0.f ? v : 0.f;
      yv[e] = v;
    }
    if constexpr (V == 1) yb[(int64_t)r * C] = from_f32<T>(yv[0]); else Vec<T>::store(yb + (int64_t)r * C, yv);
  }
}

// dx = rstd * (dyh * gamma - (S1 + xhat * S2) / n)  =  A * dyh + E * x + F  with per-(sample, channel) constants
template <typename T, int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ X, const T* __restrict__ DY,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ grp, T* __restrict__ DX, int rows_per_wg,
                                                           int HW, int C, int G, int relu) {
  const int vpr = C / V, cg = C / G;
  const int rpi = 256 / vpr;
  const int b = blockIdx.y, cv = threadIdx.x % vpr, r0 = threadIdx.x / vpr;
  if (r0 >= rpi) return;
  const int c0 = cv * V;
  const float inv_n = 1.f / ((float)cg * (float)HW);
  float A[V], E[V], F[V], ma[V], mo[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const int c = c0 + e, g = c / cg;
    const float rs = rstd[b * G + g], mu = mean[b * G + g];
    const float S1 = grp[(b * G + g) * 2], S2 = grp[(b * G + g) * 2 + 1];
    A[e] = rs * gamma[c];
    E[e] = -rs * rs * S2 * inv_n;                  // coefficient of x:      -rstd * S2/n * xhat, xhat = (x - mean) * rstd
    F[e] = -rs * S1 * inv_n - E[e] * mu;
    ma[e] = A[e];                                  // relu mask: xhat*gamma + beta = x * ma + mo
    mo[e] = fmaf(-mu, A[e], beta[c]);
  }
  const int row_lo = blockIdx.x * rows_per_wg;
  const int row_hi = (row_lo + rows_per_wg) < HW ? (row_lo + rows_per_wg) : HW;
  const T* xb = X + (int64_t)b * HW * C + c0;
  const T* db = DY + (int64_t)b * HW * C + c0;
  T* ob = DX + (int64_t)b * HW * C + c0;
  for (int r = row_lo + r0; r < row_hi; r += rpi) {
    float xv[V], dv[V], ov[V];
    if constexpr (V == 1) { xv[0] = to_f32(xb[(int64_t)r * C]); dv[0] = to_f32(db[(int64_t)r * C]); }
    else { Vec<T>::load(xb + (int64_t)r * C, xv); Vec<T>::load(db + (int64_t)r * C, dv); }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float d = dv[e];
      if (relu && !(fmaf(xv[e], ma[e], mo[e]) > 0.f)) d = 0.f;
      ov[e] = fmaf(A[e], d, fmaf(E[e], xv[e], F[e]));
    }
    if constexpr (V == 1) ob[(int64_t)r * C] = from_f32<T>(ov[0]); else Vec<T>::store(ob + (int64_t)r * C, ov);
  }
}

// rows per workgroup of the apply kernels: a multiple of the rows per iteration, sized for >= ~8 workgroups per CU overall
static int gn_rows_per_wg(int B, int HW, int rpi) {
  int chunks = (2048 + B - 1) / B;                 // workgroups per sample
  if (chunks < 1) chunks = 1;
  int rpw = (HW + chunks - 1) / chunks;
  rpw = (rpw + rpi - 1) / rpi * rpi;
  if (rpw < 4 * rpi) rpw = 4 * rpi;                // at least four iterations to amortise the coefficient set-up
  return rpw;
}

struct StoreEpi {
  float* out;
  __device__ void operator()(int64_t i, float s) const { out[i] = s; }
};

template <typename T, int V>
static int gn_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B,
                       int HW, int C, int G, float eps, int relu, hipStream_t st) {
  const int vpr = C / V;
  const int rpi = 256 / vpr > 0 ? 256 / vpr : 1;
  constexpr int NTH = 1024;
  const size_t lds = (size_t)2 * (NTH / vpr > 0 ? NTH / vpr : 1) * C * sizeof(float);
  {
    auto kern = gn_reduce_kernel<T, V, 0, NTH>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("gn_reduce_kernel(fwd)", kern, dim3(B), dim3(NTH), lds, st, (const T*)x, (const T*)nullptr, gamma, beta,
               (const float*)nullptr, (const float*)nullptr, HW, C, G, eps, 0, mean, rstd, (float*)nullptr);
  }
  const int rpw = gn_rows_per_wg(B, HW, rpi);
  FRL_LAUNCH((gn_apply_kernel<T, V>), dim3((unsigned)((HW + rpw - 1) / rpw), (unsigned)B), dim3(256), 0, st, (const T*)x, gamma, beta,
                     (const float*)mean, (const float*)rstd, (T*)y, rpw, HW, C, G, relu);
  return frl_check_launch("groupnorm_fwd");
}

template <typename T, int V>
static int gn_bwd_impl(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int G, int relu,
                       float* ws, hipStream_t st) {
  const int vpr = C / V;
  const int rpi = 256 / vpr > 0 ? 256 / vpr : 1;
  constexpr int NTH = 1024;
  const size_t lds = (size_t)2 * (NTH / vpr > 0 ? NTH / vpr : 1) * C * sizeof(float);
  float* sdy = ws;                         // [B][C]
  float* sdyx = ws + (size_t)B * C;        // [B][C]
  float* grp = ws + (size_t)2 * B * C;     // [B][G][2]
  {
    auto kern = gn_reduce_kernel<T, V, 1, NTH>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("gn_reduce_kernel(bwd)", kern, dim3(B), dim3(NTH), lds, st, (const T*)x, (const T*)dy, gamma, beta, mean, rstd, HW, C, G, 0.f, relu, sdy, sdyx, grp);
  }
  const int rpw = gn_rows_per_wg(B, HW, rpi);
  FRL_LAUNCH((gn_bwd_apply_kernel<T, V>), dim3((unsigned)((HW + rpw - 1) / rpw), (unsigned)B), dim3(256), 0, st, (const T*)x, (const T*)dy,
                     gamma, beta, mean, rstd, (const float*)grp, (T*)dx, rpw, HW, C, G, relu);
  launch_slab_reduce<float, StoreEpi>((const float*)sdy, B, C, StoreEpi{dbeta}, st);
  launch_slab_reduce<float, StoreEpi>((const float*)sdyx, B, C, StoreEpi{dgamma}, st);
  return frl_check_launch("groupnorm_bwd");
}

extern "C" {

size_t frl_groupnorm_bwd_workspace_bytes(int B, int C, int G) { return ((size_t)2 * B * C + (size_t)2 * B * G) * sizeof(float); }

// x,y [B][HW][C]; gamma,beta [C] f32; mean,rstd [B][G] f32 outputs (saved for backward)
int frl_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int B, int HW,
                      int C, int G, float eps, int relu, int dtype, hipStream_t stream) {
  if (B <= 0 || HW <= 0) return frl_fail(-2, "groupnorm: empty input");
  if (C % G != 0 || C > 1024 || G > 256) return frl_fail(-2, "groupnorm: need C % G == 0, C <= 1024, G <= 256");
  if (dtype == FRL_F32) {
    if (C % 4 == 0 && C / 4 <= 256) return gn_fwd_impl<float, 4>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, relu, stream);
    if (C <= 256) return gn_fwd_impl<float, 1>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, relu, stream);
  } else if (dtype == FRL_BF16) {
    if (C % 8 == 0 && C / 8 <= 256) return gn_fwd_impl<bf16, 8>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, relu, stream);
    if (C <= 256) return gn_fwd_impl<bf16, 1>(x, gamma, beta, y, mean, rstd, B, HW, C, G, eps, relu, stream);
  }
  return frl_fail(-2, "groupnorm: unsupported dtype / channel count");
}

int frl_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int G, int relu,
                      int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (ws_bytes < frl_groupnorm_bwd_workspace_bytes(B, C, G)) return frl_fail(-4, "groupnorm_bwd: workspace too small");
  if (C % G != 0 || C > 1024 || G > 256) return frl_fail(-2, "groupnorm: need C % G == 0, C <= 1024, G <= 256");
  if (dtype == FRL_F32) {
    if (C % 4 == 0 && C / 4 <= 256) return gn_bwd_impl<float, 4>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, B, HW, C, G, relu, (float*)ws, stream);
    if (C <= 256) return gn_bwd_impl<float, 1>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, B, HW, C, G, relu, (float*)ws, stream);
  } else if (dtype == FRL_BF16) {
    if (C % 8 == 0 && C / 8 <= 256) return gn_bwd_impl<bf16, 8>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, B, HW, C, G, relu, (float*)ws, stream);
    if (C <= 256) return gn_bwd_impl<bf16, 1>(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, B, HW, C, G, relu, (float*)ws, stream);
  }
  return frl_fail(-2, "groupnorm_bwd: unsupported dtype / channel count");
}

}  // extern "C"
