// Shared device code of the fused TCN (GatedResidualBlock) kernels.
// Reference: frl/models/tcn.py:78-111 (block), :220-302 (5-D <-> [B*H*W, C, T] flattening).
//
// Data layout: x [B][T][HW][C] (the tile's own (time, y, x, feature) order).  A wave owns 16 pixels for ALL time
// steps: lane (px = l & 15, kc = l >> 4) holds the lane-quarter of its pixel's channels, so
//   * the 3 temporal taps are the SAME lane at other time steps (no cross-lane traffic),
//   * the GroupNorm over (C/G channels x T) of one pixel is an in-lane reduction,
//   * the accumulator image of conv / gate GEMMs is again a lane-quarter image (see frl_common.hpp), so the
//     normalised features feed the gate GEMM and the block output feeds HBM with no LDS transpose.
// With TP > 0 the pixel's whole time series (TP time steps) is loaded ONCE per tile into registers in a single batch
// of 16-byte loads (one memory latency per tile instead of one per tap), and every conv pass reuses it.
#pragma once
#include "frl_common.hpp"

struct TcnArgs {
  int64_t npix;       // B * HW
  int HW, Tn, dil;
  int Cin, Cout, G;
  float eps;
};

// register cache of one pixel's time series (lane-quarter images for t = 0..TP-1)
template <typename T, int NFI, int TP> struct XCache { LQTile<T, NFI> xs[TP > 0 ? TP : 1]; };

template <typename T, int NFI, int TP>
__device__ __forceinline__ void xcache_load(XCache<T, NFI, TP>& xc, const T* __restrict__ X, int64_t row0, const TcnArgs& a, int C,
                                            int kc, bool fast) {
  if constexpr (TP > 0) {
#pragma unroll
    for (int u = 0; u < TP; ++u)
      if (u < a.Tn) lq_load<T, NFI>(xc.xs[u], X, row0 + (int64_t)u * a.HW, C, kc, fast);
  }
}

// lane-quarter image of time step t (t is wave-uniform): from the register cache or straight from memory
template <typename T, int NFI, int TP>
__device__ __forceinline__ void xcache_get(LQTile<T, NFI>& o, const XCache<T, NFI, TP>& xc, const T* __restrict__ X, int64_t row0,
                                           const TcnArgs& a, int t, int C, int kc, bool fast) {
  if constexpr (TP > 0) {
#pragma unroll
    for (int u = 0; u < TP; ++u)
      if (u == t) o = xc.xs[u];
  } else {
    lq_load<T, NFI>(o, X, row0 + (int64_t)t * a.HW, C, kc, fast);
  }
}

// conv output (no bias) for time t of this lane's pixel: 3 taps at t-d, t, t+d (zero padded in time)
template <typename T, int NFI, int MBO, int TP>
__device__ __forceinline__ void tconv_at(f32x4 (&acc)[MBO], const XCache<T, NFI, TP>& xc, const T* __restrict__ X, int64_t row0,
                                         const TcnArgs& a, int t, int C, int kc, bool fast,
                                         const typename DT<T>::frag_t* __restrict__ wl, int lane) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tin = t + (k - 1) * a.dil;
    if (tin < 0 || tin >= a.Tn) continue;
    if constexpr (TP > 0) {
#pragma unroll
      for (int u = 0; u < TP; ++u) {
        if (u == tin) {
#pragma unroll
          for (int m = 0; m < MBO; ++m)
#pragma unroll
            for (int s = 0; s < NFI; ++s) acc[m] = mfma16(wl[((k * MBO + m) * NFI + s) * 64 + lane], xc.xs[u].f[s], acc[m]);
        }
      }
    } else {
      LQTile<T, NFI> xt;
      lq_load<T, NFI>(xt, X, row0 + (int64_t)tin * a.HW, C, kc, fast);
#pragma unroll
      for (int m = 0; m < MBO; ++m)
#pragma unroll
        for (int s = 0; s < NFI; ++s) acc[m] = mfma16(wl[((k * MBO + m) * NFI + s) * 64 + lane], xt.f[s], acc[m]);
    }
  }
}

// pointwise GEMM on a lane-quarter tile: acc[m] += W (packed [MBO][NF][64]) * tile
template <typename T, int NF, int MBO>
__device__ __forceinline__ void pw_at(f32x4 (&acc)[MBO], const LQTile<T, NF>& xt, const typename DT<T>::frag_t* __restrict__ wl, int lane) {
#pragma unroll
  for (int m = 0; m < MBO; ++m)
#pragma unroll
    for (int s = 0; s < NF; ++s) acc[m] = mfma16(wl[(m * NF + s) * 64 + lane], xt.f[s], acc[m]);
}

// accumulator image (channel j = 4*m + r of the lane quarter) -> lane-quarter tile of dtype T
template <typename T, int MBO>
__device__ __forceinline__ void acc_to_tile(LQTile<T, (4 * MBO) / DT<T>::FE>& o, const float (&v)[4 * MBO]) {
  constexpr int FE = DT<T>::FE;
#pragma unroll
  for (int j = 0; j < 4 * MBO; ++j) lq_set<T, (4 * MBO) / FE>(o, j / FE, j % FE, v[j]);
}

template <typename T, int NF>
__device__ __forceinline__ void lq_store(const LQTile<T, NF>& t, T* __restrict__ Y, int64_t row, int C, int kc, bool fast) {
  constexpr int FE = DT<T>::FE;
  const int q = NF * FE;
  T* p = Y + row * (int64_t)C + q * kc;
  if (fast) {
    if constexpr (FE == 8) {
#pragma unroll
      for (int s = 0; s < NF; ++s) *reinterpret_cast<bf16x8*>(p + 8 * s) = t.f[s];
    } else {
#pragma unroll
      for (int s = 0; s < NF; s += 4) *reinterpret_cast<f32x4*>(p + s) = f32x4{t.f[s], t.f[s + 1], t.f[s + 2], t.f[s + 3]};
    }
  } else {
#pragma unroll
    for (int s = 0; s < NF; ++s)
#pragma unroll
      for (int e = 0; e < FE; ++e) {
        const int c = q * kc + s * FE + e;
        if (c < C) p[s * FE + e] = from_f32<T>(lq_get<T, NF>(t, s, e));
      }
  }
}

// static-group-size versions (cg known at compile time: a handful of adds instead of Q*Q predicated ones)
template <int Q, int CG>
__device__ __forceinline__ void group_combine_s(float (&out)[Q], const float (&in)[Q]) {
#pragma unroll
  for (int g = 0; g < Q / CG; ++g) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CG; ++j) s += in[g * CG + j];
#pragma unroll
    for (int j = 0; j < CG; ++j) out[g * CG + j] = s;
  }
}
template <int Q, int CG>
__device__ __forceinline__ void group_first_s(float (&out)[Q], const float (&in)[Q]) {
#pragma unroll
  for (int j = 0; j < Q; ++j) out[j] = in[(j / CG) * CG];
}

// In-lane group combine: out[j] = sum over j2 in the same group (j2 / cg == j / cg) of in[j2]
template <int Q>
__device__ __forceinline__ void group_combine(float (&out)[Q], const float (&in)[Q], int cg) {
  if constexpr (Q % 8 == 0) { if (cg == 8) { group_combine_s<Q, 8>(out, in); return; } }
  if constexpr (Q % 4 == 0) { if (cg == 4) { group_combine_s<Q, 4>(out, in); return; } }
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    float s = 0.f;
    const int gj = j / cg;
#pragma unroll
    for (int j2 = 0; j2 < Q; ++j2) s += (j2 / cg == gj) ? in[j2] : 0.f;
    out[j] = s;
  }
}
// out[j] = in[first channel of j's group]
template <int Q>
__device__ __forceinline__ void group_first(float (&out)[Q], const float (&in)[Q], int cg) {
  if constexpr (Q % 8 == 0) { if (cg == 8) { group_first_s<Q, 8>(out, in); return; } }
  if constexpr (Q % 4 == 0) { if (cg == 4) { group_first_s<Q, 4>(out, in); return; } }
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    float s = 0.f;
    const int first = (j / cg) * cg;
#pragma unroll
    for (int j2 = 0; j2 < Q; ++j2) s = (j2 == first) ? in[j2] : s;
    out[j] = s;
  }
}

// Per-lane GroupNorm statistics of the conv output c = acc + bias over (group channels x T), shifted sums for accuracy.
// Produces for the lane's Q = 4*MBO channels:  scale[j] = rstd,  shiftv[j] = (bias_j - mean_group) * rstd  so that
//   xhat = acc * scale + shiftv   (acc = bias-free MFMA accumulator).
template <typename T, int NFI, int MBO, int TP>
__device__ __forceinline__ void tcn_stats(float (&scale)[4 * MBO], float (&shiftv)[4 * MBO], const XCache<T, NFI, TP>& xc,
                                          const T* __restrict__ X, int64_t row0, const TcnArgs& a, int kc, bool fast,
                                          const typename DT<T>::frag_t* __restrict__ wl_conv, const float* __restrict__ bias,
                                          int lane) {
  constexpr int Q = 4 * MBO;
  const int cg = a.Cout / a.G;
  float shift[Q], s1[Q], s2[Q], cb[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) cb[j] = (Q * kc + j < a.Cout) ? bias[Q * kc + j] : 0.f;
#pragma unroll
  for (int j = 0; j < Q; ++j) { shift[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  for (int t = 0; t < a.Tn; ++t) {
    f32x4 acc[MBO];
#pragma unroll
    for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    tconv_at<T, NFI, MBO, TP>(acc, xc, X, row0, a, t, a.Cin, kc, fast, wl_conv, lane);
    float v[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) v[j] = acc[j >> 2][j & 3] + cb[j];
    if (t == 0) group_first<Q>(shift, v, cg);
#pragma unroll
    for (int j = 0; j < Q; ++j) { const float d = v[j] - shift[j]; s1[j] += d; s2[j] = fmaf(d, d, s2[j]); }
  }
  float g1[Q], g2[Q];
  group_combine<Q>(g1, s1, cg);
  group_combine<Q>(g2, s2, cg);
  const float inv_n = 1.f / ((float)cg * (float)a.Tn);
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const float md = g1[j] * inv_n;
    float var = g2[j] * inv_n - md * md;
    var = var > 0.f ? var : 0.f;
    const float rs = 1.f / sqrtf(var + a.eps);
    scale[j] = rs;
    shiftv[j] = (cb[j] - (shift[j] + md)) * rs;
  }
}

// One launch writes every packed weight image a TCN kernel needs (conv taps, gate, gate^T, projection) to the workspace.
// mode 0: forward (conv, gate, proj) ; mode 1: backward (conv, gate, gate^T, proj) ; mode 2: tconv3 (conv [rev], extra pointwise)
template <typename T, int NFI, int MBO, int NFP>
__global__ void tcn_pack_kernel(typename DT<T>::frag_t* __restrict__ dst, int mode, const float* __restrict__ Wc, int64_t so, int64_t si,
                                int rev, const float* __restrict__ Wg, const float* __restrict__ Wp, int64_t pso, int64_t psi, int Cp,
                                int Cin, int Cout) {
  constexpr int NFO = 4 * MBO / DT<T>::FE;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int k = 0; k < 3; ++k)
    pack_weights_lds<T, NFI>(dst + k * MBO * NFI * 64, Wc + (rev ? 2 - k : k), Cout, Cin, MBO, so, si, tid, nt);
  typename DT<T>::frag_t* p = dst + 3 * MBO * NFI * 64;
  if (mode != 2) {
    pack_weights_lds<T, NFO>(p, Wg, Cout, Cout, MBO, Cout, 1, tid, nt);
    p += MBO * NFO * 64;
    if (mode == 1) { pack_weights_lds<T, NFO>(p, Wg, Cout, Cout, MBO, 1, Cout, tid, nt); p += MBO * NFO * 64; }
  }
  if (Wp != nullptr) pack_weights_lds<T, NFP>(p, Wp, Cout, Cp, MBO, pso, psi, tid, nt);
}

// per-channel constants of a lane quarter live in LDS ([4*Q] floats each) and are re-read when needed instead of
// occupying Q registers each for the whole kernel
__device__ __forceinline__ float lds_chan(const float* __restrict__ tab, int Q, int kc, int j) { return tab[Q * kc + j]; }
