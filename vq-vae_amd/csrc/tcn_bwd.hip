// Fused GatedResidualBlock backward, main kernel (frl/models/tcn.py:78-111 differentiated by hand).
// Recomputes conv -> GroupNorm -> gate from x (nothing but x is saved by the forward), then per pixel:
//   dg = dy (relu(n) - res) ; dgpre = dg g (1-g) ; dres = dy (1-g) ; dn = dy g [n>0] + Wg^T dgpre
//   GroupNorm backward over (group channels x T) in-lane ; dconv = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
// Outputs feed the MFMA weight-gradient kernels (pw_wgrad.hip): dconv (3 temporal taps against x), dgpre against the
// normalised features `normed`, dres against x (projection only); d gamma / d beta are reduced in fixed order.
// dx itself is produced by frl_tcn_block_bwd_data (conv^T over dconv + residual path).
// The pixel's time series is cached in registers (TP time steps, one batch of loads per tile); per-channel constants
// live in an LDS table.
#include "tcn_common.hpp"
#include "frl_host.hpp"
#include "frl_reduce.hpp"

template <typename T, int NFI, int TP> constexpr bool tcn_tp_ok_b() {
  return TP == 0 || (NFI * DT<T>::FE * 4 == 64 && TP * NFI * (int)sizeof(typename DT<T>::frag_t) / 4 <= 80);
}

template <typename T, int NFI, int MBO, int TP, int DIL>
__global__ __launch_bounds__(256) void tcn_block_bwd_kernel(const T* __restrict__ X, const T* __restrict__ DY,
                                                            const typename DT<T>::frag_t* __restrict__ Wpk,
                                                            const float* __restrict__ bc, const float* __restrict__ gn_w,
                                                            const float* __restrict__ gn_b,
                                                            const float* __restrict__ bg, int has_proj,
                                                            const float* __restrict__ bp, T* DCONV, T* __restrict__ DGPRE,
                                                            T* __restrict__ NORMED, T* __restrict__ DRES, float* __restrict__ slab, TcnArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int Q = 4 * MBO;
  constexpr int NFO = Q / FE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag_t* wl_conv = reinterpret_cast<frag_t*>(smem);                // [3][MBO][NFI][64]
  frag_t* wl_gate = wl_conv + 3 * MBO * NFI * 64;                   // [MBO][NFO][64]
  frag_t* wl_gateT = wl_gate + MBO * NFO * 64;                      // [MBO][NFO][64]  (Wg^T)
  frag_t* wl_proj = wl_gateT + MBO * NFO * 64;                      // [MBO][NFI][64]
  float* tab = reinterpret_cast<float*>(wl_proj + (has_proj ? MBO * NFI * 64 : 0));   // gw | gb | gbias | pb, each [4*Q]
  float* gacc_lds = tab + 16 * Q;                         // [4 waves][2][4*Q]: per-wave d gamma / d beta accumulators
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<T>(wl_conv, Wpk, (3 * MBO * NFI + 2 * MBO * NFO + (has_proj ? MBO * NFI : 0)) * 64, tid, 256);
  for (int i = tid; i < 4 * Q; i += 256) {
    const bool ok = i < a.Cout;
    tab[i] = ok ? gn_w[i] : 0.f;
    tab[4 * Q + i] = ok ? gn_b[i] : 0.f;
    tab[8 * Q + i] = ok ? bg[i] : 0.f;
    tab[12 * Q + i] = (ok && has_proj) ? bp[i] : 0.f;
  }
  for (int i = tid; i < 4 * 2 * 4 * Q; i += 256) gacc_lds[i] = 0.f;
  __syncthreads();
  float* my_dg = gacc_lds + (wave * 2 + 0) * 4 * Q + Q * kc;
  float* my_db = gacc_lds + (wave * 2 + 1) * 4 * Q + Q * kc;
  const float* tgw = tab + Q * kc;
  const float* tgb = tab + 4 * Q + Q * kc;
  const float* tbg = tab + 8 * Q + Q * kc;
  const float* tpb = tab + 12 * Q + Q * kc;

  const bool fast_in = (a.Cin == 4 * NFI * FE), fast_out = (a.Cout == 4 * Q);
  const int cg = a.Cout / a.G;
  const float inv_n = 1.f / ((float)cg * (float)a.Tn);
  const int64_t ntile = (a.npix + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    int64_t pidx = tile * 16 + px;
    const bool valid = pidx < a.npix;
    const float vf = valid ? 1.f : 0.f;
    if (!valid) pidx = a.npix - 1;
    const int64_t b = pidx / a.HW, hw = pidx % a.HW;
    const int64_t row0 = b * a.Tn * a.HW + hw;
    XCache<T, NFI, TP> xc;
    xcache_load<T, NFI, TP>(xc, X, row0, a, a.Cin, kc, fast_in);
    float rs[Q], sh[Q];                                   // xhat = acc * rs + sh
    tcn_stats<T, NFI, MBO, TP>(rs, sh, xc, X, row0, a, kc, fast_in, wl_conv, bc, lane);
    float S1[Q], S2[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) { S1[j] = 0.f; S2[j] = 0.f; }
    // ---------------- pass 2: dn, side outputs, group sums ----------------
    for (int t = 0; t < a.Tn; ++t) {
      const int64_t row = row0 + (int64_t)t * a.HW;
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<T, NFI, MBO, TP>(acc, xc, X, row0, a, t, a.Cin, kc, fast_in, wl_conv, lane);
      float xh[Q], n[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) { xh[j] = fmaf(acc[j >> 2][j & 3], rs[j], sh[j]); n[j] = fmaf(xh[j], tgw[j], tgb[j]); }
      LQTile<T, NFO> nt;
      acc_to_tile<T, MBO>(nt, n);
      f32x4 gacc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pw_at<T, NFO, MBO>(gacc, nt, wl_gate, lane);
      LQTile<T, NFI> xt;
      xcache_get<T, NFI, TP>(xt, xc, X, row0, a, t, a.Cin, kc, fast_in);
      float res[Q];
      if (has_proj) {
        f32x4 pacc[MBO];
#pragma unroll
        for (int m = 0; m < MBO; ++m) pacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        pw_at<T, NFI, MBO>(pacc, xt, wl_proj, lane);
#pragma unroll
        for (int j = 0; j < Q; ++j) res[j] = pacc[j >> 2][j & 3] + tpb[j];
      } else {
        if constexpr (NFI * FE == Q) {
#pragma unroll
          for (int j = 0; j < Q; ++j) res[j] = lq_get<T, NFI>(xt, j / FE, j % FE);
        } else {
#pragma unroll
          for (int j = 0; j < Q; ++j) res[j] = 0.f;
        }
      }
      LQTile<T, NFO> dyt;
      lq_load<T, NFO>(dyt, DY, row, a.Cout, kc, fast_out);
      float dgp[Q], dn[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const float dy = lq_get<T, NFO>(dyt, j / FE, j % FE) * vf;
        const float g = sigmoid_t<T>(gacc[j >> 2][j & 3] + tbg[j]);
        const float o = n[j] > 0.f ? n[j] : 0.f;
        dgp[j] = dy * (o - res[j]) * g * (1.f - g);
        res[j] = dy * (1.f - g);                      // reuse as d res
        dn[j] = n[j] > 0.f ? dy * g : 0.f;
      }
      LQTile<T, NFO> gt;
      acc_to_tile<T, MBO>(gt, dgp);
      f32x4 bacc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) bacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pw_at<T, NFO, MBO>(bacc, gt, wl_gateT, lane);
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        dn[j] += bacc[j >> 2][j & 3];
        const float dxh = dn[j] * tgw[j];
        S1[j] += dxh;
        S2[j] = fmaf(dxh, xh[j], S2[j]);
      }
      if (valid) {
        lq_store<T, NFO>(nt, NORMED, row, a.Cout, kc, fast_out);
        lq_store<T, NFO>(gt, DGPRE, row, a.Cout, kc, fast_out);
        LQTile<T, NFO> tt;
        acc_to_tile<T, MBO>(tt, res);
        lq_store<T, NFO>(tt, DRES, row, a.Cout, kc, fast_out);
        acc_to_tile<T, MBO>(tt, dn);
        lq_store<T, NFO>(tt, DCONV, row, a.Cout, kc, fast_out);      // temporary: dn, rewritten below
      }
    }
    float m1[Q], m2[Q];
    group_combine<Q>(m1, S1, cg);
    group_combine<Q>(m2, S2, cg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own dn stores have landed before they are re-read
    // ---------------- pass 3: GroupNorm backward -> dconv ; d gamma / d beta of this tile ----------------
    float dgam[Q], dbet[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) { dgam[j] = 0.f; dbet[j] = 0.f; }
    for (int t = 0; t < a.Tn; ++t) {
      const int64_t row = row0 + (int64_t)t * a.HW;
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<T, NFI, MBO, TP>(acc, xc, X, row0, a, t, a.Cin, kc, fast_in, wl_conv, lane);
      LQTile<T, NFO> dnt;
      lq_load<T, NFO>(dnt, DCONV, row, a.Cout, kc, fast_out);
      float dc[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const float xh = fmaf(acc[j >> 2][j & 3], rs[j], sh[j]);
        const float dnv = lq_get<T, NFO>(dnt, j / FE, j % FE) * vf;
        const float dxh = dnv * tgw[j];
        dc[j] = rs[j] * (dxh - m1[j] * inv_n - xh * m2[j] * inv_n);
        dgam[j] = fmaf(dnv, xh, dgam[j]);
        dbet[j] += dnv;
      }
      if (valid) {
        LQTile<T, NFO> tt;
        acc_to_tile<T, MBO>(tt, dc);
        lq_store<T, NFO>(tt, DCONV, row, a.Cout, kc, fast_out);
      }
    }
    // reduce this tile's d gamma / d beta over the 16 pixel lanes and accumulate in the wave's private LDS rows
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { dgam[j] += __shfl_xor(dgam[j], off, 64); dbet[j] += __shfl_xor(dbet[j], off, 64); }
    }
    if (px == 0) {
#pragma unroll
      for (int j = 0; j < Q; ++j) { my_dg[j] += dgam[j]; my_db[j] += dbet[j]; }
    }
  }
  // ---------------- d gamma / d beta: sum the 4 per-wave LDS accumulators, write the workgroup slab ----------------
  __syncthreads();
  for (int i = tid; i < 2 * a.Cout; i += 256) {
    const int which = i / a.Cout, c = i % a.Cout;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += gacc_lds[(w * 2 + which) * 4 * Q + c];
    slab[(int64_t)blockIdx.x * 2 * a.Cout + i] = s;
  }
}

static unsigned tcn_bwd_grid(int64_t npix) {
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (unsigned)g;
}

struct TcnGbEpi {
  int n; float* dgamma; float* dbeta;
  __device__ void operator()(int64_t i, float s) const { if (i < n) dgamma[i] = s; else dbeta[i - n] = s; }
};

template <typename T, int NFI, int MBO, int TP, int DIL>
static int launch_tcn_bwd_tp(const void* x, const void* dy, const void* pk, const float* bc, const float* gw, const float* gb,
                             const float* bg, const float* wp, const float* bp, void* dconv, void* dgpre, void* normed, void* dres,
                             float* ws, const TcnArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  if constexpr (tcn_tp_ok_b<T, NFI, TP>()) {
    auto kern = tcn_block_bwd_kernel<T, NFI, MBO, TP, DIL>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("tcn_bwd_kernel", kern, dim3(grid), dim3(256), lds, st, (const T*)x, (const T*)dy, (const frag_t*)pk, bc, gw, gb, bg, wp ? 1 : 0, bp,
               (T*)dconv, (T*)dgpre, (T*)normed, (T*)dres, ws, a);
    return 0;
  } else {
    return launch_tcn_bwd_tp<T, NFI, MBO, 0, 0>(x, dy, pk, bc, gw, gb, bg, wp, bp, dconv, dgpre, normed, dres, ws, a, grid, lds, st);
  }
}

template <typename T, int NFI, int MBO>
static int launch_tcn_bwd(const void* x, const void* dy, const float* wc, const float* bc, const float* gw, const float* gb, const float* wg,
                          const float* bg, const float* wp, const float* bp, void* dconv, void* dgpre, void* normed, void* dres,
                          float* dgamma, float* dbeta, float* ws, const TcnArgs& a, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int NFO = 4 * MBO / DT<T>::FE;
  size_t lds = (size_t)(3 * MBO * NFI + 2 * MBO * NFO + (wp ? MBO * NFI : 0)) * 64 * sizeof(frag_t) + (size_t)(16 + 32) * 4 * MBO * sizeof(float);
  if (lds > 160 * 1024) return frl_fail(-3, "tcn_block_bwd: weights exceed LDS");
  const unsigned grid = tcn_bwd_grid(a.npix);
  frag_t* pk = reinterpret_cast<frag_t*>(reinterpret_cast<char*>(ws) + (((size_t)grid * 2 * a.Cout * sizeof(float) + 255) / 256) * 256);
  FRL_LAUNCH((tcn_pack_kernel<T, NFI, MBO, NFI>), dim3(32), dim3(256), 0, st, pk, 1, wc, (int64_t)a.Cin * 3, (int64_t)3, 0, wg, wp,
             (int64_t)a.Cin, (int64_t)1, a.Cin, a.Cin, a.Cout);
  int rc;
#define BWD_TP(D) launch_tcn_bwd_tp<T, NFI, MBO, 5, D>(x, dy, pk, bc, gw, gb, bg, wp, bp, dconv, dgpre, normed, dres, ws, a, grid, lds, st)
  if (a.Tn <= 5) rc = BWD_TP(0);
  else rc = launch_tcn_bwd_tp<T, NFI, MBO, 0, 0>(x, dy, pk, bc, gw, gb, bg, wp, bp, dconv, dgpre, normed, dres, ws, a, grid, lds, st);
#undef BWD_TP
  if (rc) return rc;
  launch_slab_reduce<float, TcnGbEpi>((const float*)ws, (int)grid, 2 * a.Cout, TcnGbEpi{a.Cout, dgamma, dbeta}, st);
  return frl_check_launch("tcn_block_bwd");
}

static int pad_class_b(int C, int dtype) {
  if (dtype == FRL_F32) return C <= 16 ? 16 : C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : -1;
  return C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : -1;
}
int frl_tcn_check(int Cin, int Cout, int G, int dtype);

#define TCN_SWITCH_F32(PI, PO, CALL)                                   \
  switch ((PI) * 1000 + (PO)) {                                        \
    case 16016: { CALL(float, 4, 1) } case 16032: { CALL(float, 4, 2) } case 16064: { CALL(float, 4, 4) } case 16128: { CALL(float, 4, 8) } \
    case 32016: { CALL(float, 8, 1) } case 32032: { CALL(float, 8, 2) } case 32064: { CALL(float, 8, 4) } case 32128: { CALL(float, 8, 8) } \
    case 64016: { CALL(float, 16, 1) } case 64032: { CALL(float, 16, 2) } case 64064: { CALL(float, 16, 4) } case 64128: { CALL(float, 16, 8) } \
    case 128016: { CALL(float, 32, 1) } case 128032: { CALL(float, 32, 2) } case 128064: { CALL(float, 32, 4) } case 128128: { CALL(float, 32, 8) } \
    default: break; }
#define TCN_SWITCH_BF16(PI, PO, CALL)                                  \
  switch ((PI) * 1000 + (PO)) {                                        \
    case 32032: { CALL(bf16, 1, 2) } case 32064: { CALL(bf16, 1, 4) } case 32128: { CALL(bf16, 1, 8) }     \
    case 64032: { CALL(bf16, 2, 2) } case 64064: { CALL(bf16, 2, 4) } case 64128: { CALL(bf16, 2, 8) }     \
    case 128032: { CALL(bf16, 4, 2) } case 128064: { CALL(bf16, 4, 4) } case 128128: { CALL(bf16, 4, 8) }  \
    default: break; }

extern "C" {

size_t frl_tcn_block_bwd_workspace_bytes(int64_t npix, int Cout) {
  return (size_t)tcn_bwd_grid(npix) * 2 * Cout * sizeof(float) + 256 + (size_t)6 * 128 * 128 * sizeof(float);   // slabs + packed weights
}

// Side outputs (all [B][T][HW][Cout], dtype): dconv, dgpre, normed, dres; dgamma/dbeta [Cout] f32.
int frl_tcn_block_bwd(const void* x, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                      const float* gate_w, const float* gate_b, const float* proj_w, const float* proj_b, void* dconv, void* dgpre,
                      void* normed, void* dres, float* dgamma, float* dbeta, int64_t npix, int HW, int T, int Cin, int Cout,
                      int dilation, int G, float eps, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (npix <= 0 || T <= 0) return frl_fail(-2, "tcn_block_bwd: empty input");
  if (ws_bytes < frl_tcn_block_bwd_workspace_bytes(npix, Cout)) return frl_fail(-4, "tcn_block_bwd: workspace too small");
  if (proj_w == nullptr && Cin != Cout) return frl_fail(-2, "tcn_block_bwd: identity residual needs Cin == Cout");
  int rc = frl_tcn_check(Cin, Cout, G, dtype);
  if (rc) return rc;
  TcnArgs a{npix, HW, T, dilation, Cin, Cout, G, eps};
  const int pi = pad_class_b(Cin, dtype), po = pad_class_b(Cout, dtype);
#define CALL(TT, NFI, MBO) return launch_tcn_bwd<TT, NFI, MBO>(x, dy, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b, dconv, dgpre, normed, dres, dgamma, dbeta, (float*)ws, a, stream);
  if (dtype == FRL_F32) TCN_SWITCH_F32(pi, po, CALL)
  else if (dtype == FRL_BF16) TCN_SWITCH_BF16(pi, po, CALL)
#undef CALL
  return frl_fail(-2, "tcn_block_bwd: unsupported dtype / widths");
}

}  // extern "C"
