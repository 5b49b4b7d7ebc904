// Fused GatedResidualBlock forward (frl/models/tcn.py:78-111), one launch per block:
//   res = proj(x) | x ;  c = conv1d_k3_dil(x) + b ;  n = GroupNorm_(8ch x T per pixel)(c) ;  g = sigmoid(Wg n + bg) ;
//   y = g * relu(n) + (1 - g) * res
// plus the generic 3-tap temporal convolution used by the backward pass (dx = conv^T(dconv) [+ residual]).
// MFMA work per (pixel, t): 3 taps x Cin x Cout (computed twice: statistics pass + apply pass) + Cout x Cout gate.
// HBM traffic: x read (taps hit L1/L2), y written once -- the GroupNorm / gate / blend intermediates never leave the CU.
#include "tcn_common.hpp"
#include "frl_host.hpp"

// TP = number of time steps cached in registers per pixel (0 = fetch taps from memory on demand, any T)
template <typename T, int NFI, int TP> constexpr bool tcn_tp_ok() {
  return TP == 0 || (NFI * DT<T>::FE * 4 == 64 && TP * NFI * (int)sizeof(typename DT<T>::frag_t) / 4 <= 80);
}

template <typename T, int NFI, int MBO, int TP, int DIL>
__global__ __launch_bounds__(256) void tcn_block_fwd_kernel(const T* __restrict__ X, const typename DT<T>::frag_t* __restrict__ Wpk,
                                                            const float* __restrict__ bc,
                                                            const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                            const float* __restrict__ bg, int has_proj, const float* __restrict__ bp,
                                                            T* __restrict__ Y, TcnArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int Q = 4 * MBO;            // output channels per lane quarter
  constexpr int NFO = Q / FE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag_t* wl_conv = reinterpret_cast<frag_t*>(smem);                // [3][MBO][NFI][64]
  frag_t* wl_gate = wl_conv + 3 * MBO * NFI * 64;                   // [MBO][NFO][64]
  frag_t* wl_proj = wl_gate + MBO * NFO * 64;                       // [MBO][NFI][64] (only if has_proj)
  float* tab = reinterpret_cast<float*>(wl_proj + (has_proj ? MBO * NFI * 64 : 0));   // gw | gb | gbias | pb, each [4*Q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<T>(wl_conv, Wpk, (3 * MBO * NFI + MBO * NFO + (has_proj ? MBO * NFI : 0)) * 64, tid, 256);
  for (int i = tid; i < 4 * Q; i += 256) {
    const bool ok = i < a.Cout;
    tab[i] = ok ? gn_w[i] : 0.f;
    tab[4 * Q + i] = ok ? gn_b[i] : 0.f;
    tab[8 * Q + i] = ok ? bg[i] : 0.f;
    tab[12 * Q + i] = (ok && has_proj) ? bp[i] : 0.f;
  }
  __syncthreads();
  const float* tgw = tab + Q * kc;
  const float* tgb = tab + 4 * Q + Q * kc;
  const float* tbg = tab + 8 * Q + Q * kc;
  const float* tpb = tab + 12 * Q + Q * kc;

  const bool fast_in = (a.Cin == 4 * NFI * FE), fast_out = (a.Cout == 4 * Q);
  const int64_t ntile = (a.npix + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    int64_t pidx = tile * 16 + px;
    const bool valid = pidx < a.npix;
    if (!valid) pidx = a.npix - 1;
    const int64_t b = pidx / a.HW, hw = pidx % a.HW;
    const int64_t row0 = b * a.Tn * a.HW + hw;
    XCache<T, NFI, TP> xc;
    xcache_load<T, NFI, TP>(xc, X, row0, a, a.Cin, kc, fast_in);
    float A[Q], Bc[Q];
    tcn_stats<T, NFI, MBO, TP>(A, Bc, xc, X, row0, a, kc, fast_in, wl_conv, bc, lane);
#pragma unroll
    for (int j = 0; j < Q; ++j) { const float gw = tgw[j]; Bc[j] = fmaf(Bc[j], gw, tgb[j]); A[j] *= gw; }   // n = acc*A + Bc
    for (int t = 0; t < a.Tn; ++t) {
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<T, NFI, MBO, TP>(acc, xc, X, row0, a, t, a.Cin, kc, fast_in, wl_conv, lane);
      float n[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) n[j] = fmaf(acc[j >> 2][j & 3], A[j], Bc[j]);
      LQTile<T, NFO> nt;
      acc_to_tile<T, MBO>(nt, n);
      f32x4 gacc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pw_at<T, NFO, MBO>(gacc, nt, wl_gate, lane);
      LQTile<T, NFI> xt;
      xcache_get<T, NFI, TP>(xt, xc, X, row0, a, t, a.Cin, kc, fast_in);
      float y[Q];
      if (has_proj) {
        f32x4 pacc[MBO];
#pragma unroll
        for (int m = 0; m < MBO; ++m) pacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        pw_at<T, NFI, MBO>(pacc, xt, wl_proj, lane);
#pragma unroll
        for (int j = 0; j < Q; ++j) y[j] = pacc[j >> 2][j & 3] + tpb[j];
      } else {
        if constexpr (NFI * FE == Q) {      // identity residual: x tile is already the lane-quarter image
#pragma unroll
          for (int j = 0; j < Q; ++j) y[j] = lq_get<T, NFI>(xt, j / FE, j % FE);
        } else {
#pragma unroll
          for (int j = 0; j < Q; ++j) y[j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const float g = sigmoid_t<T>(gacc[j >> 2][j & 3] + tbg[j]);
        const float o = n[j] > 0.f ? n[j] : 0.f;
        y[j] = g * o + (1.f - g) * y[j];
      }
      if (valid) {
        LQTile<T, NFO> yt;
        acc_to_tile<T, MBO>(yt, y);
        lq_store<T, NFO>(yt, Y, row0 + (int64_t)t * a.HW, a.Cout, kc, fast_out);
      }
    }
  }
}

// Generic 3-tap temporal convolution with optional extra pointwise term:
//   Y[t] = sum_k W_k X[t + (k-1) dil]  (+ R[t])  (+ Wp R2[t]);  used as conv^T (+ residual path) in the TCN backward.
template <typename T, int NFI, int MBO, int NFP, int TP, int DIL>
__global__ __launch_bounds__(256) void tconv3_kernel(const T* __restrict__ X, const typename DT<T>::frag_t* __restrict__ Wpk,
                                                     const T* __restrict__ R, const T* __restrict__ R2, int has_p,
                                                     int Cp, T* __restrict__ Y, TcnArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int Q = 4 * MBO, NFO = Q / FE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag_t* wl_conv = reinterpret_cast<frag_t*>(smem);
  frag_t* wl_p = wl_conv + 3 * MBO * NFI * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<T>(wl_conv, Wpk, (3 * MBO * NFI + (has_p ? MBO * NFP : 0)) * 64, tid, 256);
  __syncthreads();
  const bool fast_in = (a.Cin == 4 * NFI * FE), fast_out = (a.Cout == 4 * Q), fast_p = (Cp == 4 * NFP * FE);
  const int64_t ntile = (a.npix + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    int64_t pidx = tile * 16 + px;
    const bool valid = pidx < a.npix;
    if (!valid) pidx = a.npix - 1;
    const int64_t b = pidx / a.HW, hw = pidx % a.HW;
    const int64_t row0 = b * a.Tn * a.HW + hw;
    XCache<T, NFI, TP> xc;
    xcache_load<T, NFI, TP>(xc, X, row0, a, a.Cin, kc, fast_in);
    for (int t = 0; t < a.Tn; ++t) {
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<T, NFI, MBO, TP>(acc, xc, X, row0, a, t, a.Cin, kc, fast_in, wl_conv, lane);
      if (has_p) {
        LQTile<T, NFP> rt;
        lq_load<T, NFP>(rt, R2, row0 + (int64_t)t * a.HW, Cp, kc, fast_p);
        pw_at<T, NFP, MBO>(acc, rt, wl_p, lane);
      }
      float y[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) y[j] = acc[j >> 2][j & 3];
      if (R != nullptr) {
        LQTile<T, NFO> rt;
        lq_load<T, NFO>(rt, R, row0 + (int64_t)t * a.HW, a.Cout, kc, fast_out);
#pragma unroll
        for (int j = 0; j < Q; ++j) y[j] += lq_get<T, NFO>(rt, j / FE, j % FE);
      }
      if (valid) {
        LQTile<T, NFO> yt;
        acc_to_tile<T, MBO>(yt, y);
        lq_store<T, NFO>(yt, Y, row0 + (int64_t)t * a.HW, a.Cout, kc, fast_out);
      }
    }
  }
}

static unsigned tcn_grid(int64_t npix) {
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <typename T, int NFI, int MBO, int TP, int DIL>
static int launch_tcn_fwd_tp(const void* x, const float* bc, const float* gw, const float* gb, const float* bg, const float* wp,
                             const float* bp, void* y, const TcnArgs& a, void* ws, size_t lds, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  if constexpr (tcn_tp_ok<T, NFI, TP>()) {
    auto kern = tcn_block_fwd_kernel<T, NFI, MBO, TP, DIL>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("tcn_fwd_kernel", kern, dim3(tcn_grid(a.npix)), dim3(256), lds, st, (const T*)x, (const frag_t*)ws, bc, gw, gb, bg, wp ? 1 : 0, bp, (T*)y, a);
    return frl_check_launch("tcn_block_fwd");
  } else {
    return launch_tcn_fwd_tp<T, NFI, MBO, 0, 0>(x, bc, gw, gb, bg, wp, bp, y, a, ws, lds, st);
  }
}

template <typename T, int NFI, int MBO>
static int launch_tcn_fwd(const void* x, const float* wc, const float* bc, const float* gw, const float* gb, const float* wg,
                          const float* bg, const float* wp, const float* bp, void* y, const TcnArgs& a, void* ws, size_t ws_bytes,
                          hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int NFO = 4 * MBO / DT<T>::FE;
  const size_t wbytes = (size_t)(3 * MBO * NFI + MBO * NFO + (wp ? MBO * NFI : 0)) * 64 * sizeof(frag_t);
  const size_t lds = wbytes + (size_t)16 * 4 * MBO * sizeof(float);
  if (lds > 160 * 1024) return frl_fail(-3, "tcn_block_fwd: weights exceed LDS");
  if (ws == nullptr || ws_bytes < wbytes) return frl_fail(-4, "tcn_block_fwd: workspace too small for the packed weights");
  FRL_LAUNCH((tcn_pack_kernel<T, NFI, MBO, NFI>), dim3(32), dim3(256), 0, st, (frag_t*)ws, 0, wc, (int64_t)a.Cin * 3, (int64_t)3, 0, wg, wp,
             (int64_t)a.Cin, (int64_t)1, a.Cin, a.Cin, a.Cout);
  if (a.Tn <= 5) return launch_tcn_fwd_tp<T, NFI, MBO, 5, 0>(x, bc, gw, gb, bg, wp, bp, y, a, ws, lds, st);
  return launch_tcn_fwd_tp<T, NFI, MBO, 0, 0>(x, bc, gw, gb, bg, wp, bp, y, a, ws, lds, st);
}

template <typename T, int NFI, int MBO, int NFP, int TP, int DIL>
static int launch_tconv3_tp(const void* x, const void* r, const void* r2, const float* wp, int Cp, void* y, const TcnArgs& a, void* ws,
                            size_t lds, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  if constexpr (tcn_tp_ok<T, NFI, TP>()) {
    auto kern = tconv3_kernel<T, NFI, MBO, NFP, TP, DIL>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("tconv3_kernel", kern, dim3(tcn_grid(a.npix)), dim3(256), lds, st, (const T*)x, (const frag_t*)ws, (const T*)r, (const T*)r2, wp ? 1 : 0,
               Cp, (T*)y, a);
    return frl_check_launch("tconv3");
  } else {
    return launch_tconv3_tp<T, NFI, MBO, NFP, 0, 0>(x, r, r2, wp, Cp, y, a, ws, lds, st);
  }
}

template <typename T, int NFI, int MBO, int NFP>
static int launch_tconv3(const void* x, const float* w, int64_t so, int64_t si, int rev, const void* r, const void* r2,
                         const float* wp, int64_t pso, int64_t psi, int Cp, void* y, const TcnArgs& a, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  const size_t lds = (size_t)(3 * MBO * NFI + (wp ? MBO * NFP : 0)) * 64 * sizeof(frag_t);
  if (lds > 160 * 1024) return frl_fail(-3, "tconv3: weights exceed LDS");
  if (ws == nullptr || ws_bytes < lds) return frl_fail(-4, "tconv3: workspace too small for the packed weights");
  FRL_LAUNCH((tcn_pack_kernel<T, NFI, MBO, NFP>), dim3(32), dim3(256), 0, st, (frag_t*)ws, 2, w, so, si, rev, (const float*)nullptr, wp,
             pso, psi, Cp, a.Cin, a.Cout);
  if (a.Tn <= 5) return launch_tconv3_tp<T, NFI, MBO, NFP, 5, 0>(x, r, r2, wp, Cp, y, a, ws, lds, st);
  return launch_tconv3_tp<T, NFI, MBO, NFP, 0, 0>(x, r, r2, wp, Cp, y, a, ws, lds, st);
}

// padded-width class: f32 -> {16,32,64,128} ; bf16 -> {32,64,128}
static int pad_class(int C, int dtype) {
  if (dtype == FRL_F32) return C <= 16 ? 16 : C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : -1;
  return C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : -1;
}

int frl_tcn_check(int Cin, int Cout, int G, int dtype) {
  const int pi = pad_class(Cin, dtype), po = pad_class(Cout, dtype);
  if (pi < 0 || po < 0) return frl_fail(-2, "tcn: channel counts above 128 unsupported");
  if (G <= 0 || Cout % G != 0) return frl_fail(-2, "tcn: Cout must be divisible by num_groups");
  const int cg = Cout / G;
  if ((po / 4) % cg != 0) return frl_fail(-2, "tcn: GroupNorm group must fit inside a lane quarter (Cout_pad/4 % (Cout/G) == 0)");
  return 0;
}

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

// x [B][T][HW][Cin] -> y [B][T][HW][Cout].  conv_w [Cout][Cin][3], gate_w [Cout][Cout], proj_w [Cout][Cin] or null
// (null <=> identity residual, requires Cin == Cout).  npix = B*HW.
int frl_tcn_block_fwd(const void* x, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                      const float* gate_w, const float* gate_b, const float* proj_w, const float* proj_b, void* y, int64_t npix,
                      int HW, int T, int Cin, int Cout, int dilation, int G, float eps, int dtype, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
  if (npix <= 0 || T <= 0) return frl_fail(-2, "tcn_block_fwd: empty input");
  if (proj_w == nullptr && Cin != Cout) return frl_fail(-2, "tcn_block_fwd: identity residual needs Cin == Cout");
  int rc = frl_tcn_check(Cin, Cout, G, dtype);
  if (rc) return rc;
  TcnArgs a{npix, HW, T, dilation, Cin, Cout, G, eps};
  const int pi = pad_class(Cin, dtype), po = pad_class(Cout, dtype);
#define CALL(TT, NFI, MBO) return launch_tcn_fwd<TT, NFI, MBO>(x, conv_w, conv_b, gn_w, gn_b, gate_w, gate_b, proj_w, proj_b, y, a, ws, ws_bytes, stream);
  if (dtype == FRL_F32) TCN_SWITCH_F32(pi, po, CALL)
  else if (dtype == FRL_BF16) TCN_SWITCH_BF16(pi, po, CALL)
#undef CALL
  return frl_fail(-2, "tcn_block_fwd: unsupported dtype / widths");
}

// dx [.. Cin] = conv^T(dconv [.. Cout]) + (proj_w ? proj_w^T dres : dres)
int frl_tcn_block_bwd_data(const void* dconv, const void* dres, const float* conv_w, const float* proj_w, void* dx, int64_t npix, int HW,
                           int T, int Cin, int Cout, int dilation, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  TcnArgs a{npix, HW, T, dilation, Cout, Cin, 1, 0.f};   // roles swapped: input width Cout, output width Cin
  const int pi = pad_class(Cout, dtype), po = pad_class(Cin, dtype);
  if (pi < 0 || po < 0) return frl_fail(-2, "tcn: channel counts above 128 unsupported");
  // Weff_k[o=ci][i=co] = conv_w[co][ci][2-k] -> so = 3, si = Cin*3, rev
  const void* r = proj_w ? nullptr : dres;
  const void* r2 = proj_w ? dres : nullptr;
#define CALL(TT, NFI, MBO) return launch_tconv3<TT, NFI, MBO, NFI>(dconv, conv_w, 3, (int64_t)Cin * 3, 1, r, r2, proj_w, 1, Cin, Cout, dx, a, ws, ws_bytes, stream);
  if (dtype == FRL_F32) TCN_SWITCH_F32(pi, po, CALL)
  else if (dtype == FRL_BF16) TCN_SWITCH_BF16(pi, po, CALL)
#undef CALL
  return frl_fail(-2, "tcn_block_bwd_data: unsupported dtype / widths");
}

}  // extern "C"
