// Fully fused GatedResidualBlock backward for the hot configuration (bf16, Cin = Cout = 64, T <= 5, identity residual):
// ONE launch reads x and dy and writes dx plus every parameter gradient of the block.  Nothing else touches HBM: the
// intermediates (normalised features, gate pre-activation gradient, dconv) that the unfused path (tcn_bwd.hip) round-trips
// through HBM for the weight-gradient GEMMs (~1 GB per launch by PMC) stay in LDS.
//
// Workgroup = 4 waves = 64 pixels (all T).  Each wave runs the per-pixel backward of its 16 pixels exactly as
// tcn_block_bwd_kernel does (register time-series cache, lane-quarter images, in-lane GroupNorm backward) and publishes
// 16-pixel tiles into LDS; the weight gradients contract over pixels, so wave w owns output rows [16w, 16w+16) of the four
// 64x64 gradient matrices (3 conv taps + gate) and accumulates them over the whole kernel in registers, fetching the
// k-strided operands with ds_read_b64_tr_b16.  Bias gradients ride along as an extra MFMA against a synthetic "ones" column.
// dx = conv^T(dconv) + dres is produced from the LDS-resident dconv tiles of the same workgroup.
//
// Reference math: frl/models/tcn.py:78-111 (differentiated by hand; see tcn_bwd.hip for the formulas).
#include "tcn_common.hpp"
#include "frl_host.hpp"
#include "frl_reduce.hpp"

#define TF_T 5            // max time steps cached
#define TF_C 64
#define TF_PITCH 72       // bf16 elements per pixel row in LDS tiles (64 + 8: 16-byte skew)

typedef bf16 TT;
typedef bf16x8 frag8;

// k-strided MFMA fragment (8 consecutive pixels of one channel) from a [pixel][TF_PITCH] LDS tile
__device__ __forceinline__ bf16x8 tr_frag(const TT* tile, int pix0, int ch0, int r16) {
  const TT* a0 = tile + (pix0 + (r16 >> 2)) * TF_PITCH + ch0 + 4 * (r16 & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * TF_PITCH));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// acc[i] += A(rows 16*wave.. of `atile`)^T-contraction with B blocks of `btile` over 64 pixels; accb += row sums (bias)
__device__ __forceinline__ void wgrad_tile(f32x4 (&acc)[4], f32x4& accb, bool with_bias, const TT* atile, const TT* btile, int wave,
                                           int r16, int kc) {
  const bf16x8 ones = (r16 == 0) ? bf16x8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f}
                                 : bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pix0 = ks * 32 + 8 * kc;
    const bf16x8 af = tr_frag(atile, pix0, wave * 16, r16);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = mfma16(af, tr_frag(btile, pix0, i * 16, r16), acc[i]);
    if (with_bias) accb = mfma16(af, ones, accb);
  }
}

// stores a lane-quarter tile (16 channels of pixel px) into row `prow` of an LDS tile
__device__ __forceinline__ void lds_put(TT* tile, int prow, int kc, const LQTile<TT, 2>& t) {
  bf16x8* p = reinterpret_cast<bf16x8*>(tile + prow * TF_PITCH + 16 * kc);
  p[0] = t.f[0];
  p[1] = t.f[1];
}
__device__ __forceinline__ void lds_get(LQTile<TT, 2>& t, const TT* tile, int prow, int kc) {
  const bf16x8* p = reinterpret_cast<const bf16x8*>(tile + prow * TF_PITCH + 16 * kc);
  t.f[0] = p[0];
  t.f[1] = p[1];
}

// slab layout per workgroup (floats): [3][64][64] conv taps | [64][64] gate | [64] dbc | [64] dbg | [64] dgamma | [64] dbeta
#define TF_SLAB (4 * 64 * 64 + 4 * 64)

__global__ __launch_bounds__(256) void tcn_fused_bwd_kernel(const TT* __restrict__ X, const TT* __restrict__ DY,
                                                            const frag8* __restrict__ Wpk, const float* __restrict__ bc,
                                                            const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                            const float* __restrict__ bg, TT* DX, float* __restrict__ slab, TcnArgs a) {
  constexpr int NFI = 2, MBO = 4, Q = 16, NFO = 2, FE = 8, TP = TF_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wl_conv = reinterpret_cast<frag8*>(smem);               // [3][4][2][64]
  frag8* wl_gate = wl_conv + 3 * MBO * NFI * 64;                 // [4][2][64]
  frag8* wl_gateT = wl_gate + MBO * NFO * 64;                    // [4][2][64]
  frag8* wl_convT = wl_gateT + MBO * NFO * 64;                   // [3][4][2][64]
  float* tab = reinterpret_cast<float*>(wl_convT + 3 * MBO * NFO * 64);   // gw | gb | gbias   (3 x 64)
  float* gacc_lds = tab + 3 * 64;                                          // [4 waves][2][64]
  TT* dc_res = reinterpret_cast<TT*>(gacc_lds + 4 * 2 * 64);               // [T][64 px][PITCH]   dn (pass 2) -> dconv (pass 3)
  TT* exA = dc_res + TF_T * 64 * TF_PITCH;                                 // [2][64 px][PITCH]   dgpre[t]  | x[t'] in pass 3b
  TT* exB = exA + 2 * 64 * TF_PITCH;                                       // [2][64 px][PITCH]   normed[t]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4, r16 = px;
  const int prow = wave * 16 + px;                                         // this lane's pixel row inside the 64-px tiles

  copy_frags_lds<TT>(wl_conv, Wpk, (3 * MBO * NFI + 2 * MBO * NFO + 3 * MBO * NFO) * 64, tid, 256);
  for (int i = tid; i < 64; i += 256) { tab[i] = gn_w[i]; tab[64 + i] = gn_b[i]; tab[128 + i] = bg[i]; }
  for (int i = tid; i < 4 * 2 * 64; i += 256) gacc_lds[i] = 0.f;
  __syncthreads();
  float* my_dg = gacc_lds + (wave * 2 + 0) * 64 + Q * kc;
  float* my_db = gacc_lds + (wave * 2 + 1) * 64 + Q * kc;
  const float* tgw = tab + Q * kc;
  const float* tgb = tab + 64 + Q * kc;
  const float* tbg = tab + 128 + Q * kc;

  f32x4 accC[3][4], accG[4], accCb, accGb;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) accC[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) accG[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  accCb = f32x4{0.f, 0.f, 0.f, 0.f};
  accGb = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cg = a.Cout / a.G;
  const float inv_n = 1.f / ((float)cg * (float)a.Tn);
  const int64_t nwt = (a.npix + 63) >> 6;                                  // 64-pixel workgroup tiles
  for (int64_t wt = blockIdx.x; wt < nwt; wt += gridDim.x) {
    int64_t pidx = wt * 64 + prow;
    const bool valid = pidx < a.npix;
    const float vf = valid ? 1.f : 0.f;
    if (!valid) pidx = a.npix - 1;
    const int64_t b = pidx / a.HW, hw = pidx % a.HW;
    const int64_t row0 = b * a.Tn * a.HW + hw;
    XCache<TT, NFI, TP> xc, dyc, drc;                       // x, dy (prefetched in one batch) and dres (kept for pass 4) time series
    xcache_load<TT, NFI, TP>(xc, X, row0, a, TF_C, kc, true);
    xcache_load<TT, NFI, TP>(dyc, DY, row0, a, TF_C, kc, true);
    float rs[Q], sh[Q];
    tcn_stats<TT, NFI, MBO, TP>(rs, sh, xc, X, row0, a, kc, true, wl_conv, bc, lane);
    float S1[Q], S2[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) { S1[j] = 0.f; S2[j] = 0.f; }
    // ---------------- pass 2: per-pixel backward up to dn; gate weight gradient per time step ----------------
    for (int t = 0; t < a.Tn; ++t) {
      const int64_t row = row0 + (int64_t)t * a.HW;
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<TT, NFI, MBO, TP>(acc, xc, X, row0, a, t, TF_C, kc, true, wl_conv, lane);
      float xh[Q], n[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) { xh[j] = fmaf(acc[j >> 2][j & 3], rs[j], sh[j]); n[j] = fmaf(xh[j], tgw[j], tgb[j]); }
      LQTile<TT, NFO> nt;
      acc_to_tile<TT, MBO>(nt, n);
      f32x4 gacc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pw_at<TT, NFO, MBO>(gacc, nt, wl_gate, lane);
      LQTile<TT, NFI> xt;
      xcache_get<TT, NFI, TP>(xt, xc, X, row0, a, t, TF_C, kc, true);
      LQTile<TT, NFO> dyt;
      xcache_get<TT, NFO, TP>(dyt, dyc, DY, row0, a, t, TF_C, kc, true);
      float dgp[Q], dn[Q], dr[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const float dy = lq_get<TT, NFO>(dyt, j / FE, j % FE) * vf;
        const float g = sigmoid_t<TT>(gacc[j >> 2][j & 3] + tbg[j]);
        const float o = n[j] > 0.f ? n[j] : 0.f;
        const float res = lq_get<TT, NFI>(xt, j / FE, j % FE);
        dgp[j] = dy * (o - res) * g * (1.f - g);
        dr[j] = dy * (1.f - g);
        dn[j] = n[j] > 0.f ? dy * g : 0.f;
      }
      LQTile<TT, NFO> gt;
      acc_to_tile<TT, MBO>(gt, dgp);
      f32x4 bacc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) bacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pw_at<TT, NFO, MBO>(bacc, gt, wl_gateT, lane);
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        dn[j] += bacc[j >> 2][j & 3];
        const float dxh = dn[j] * tgw[j];
        S1[j] += dxh;
        S2[j] = fmaf(dxh, xh[j], S2[j]);
      }
      // publish: dgpre[t], normed[t] into the exchange buffers; dn[t] into the resident dconv slot; dres[t] to DX (temporary)
      const int buf = t & 1;
      lds_put(exA + buf * 64 * TF_PITCH, prow, kc, gt);
      lds_put(exB + buf * 64 * TF_PITCH, prow, kc, nt);
      LQTile<TT, NFO> tt;
      acc_to_tile<TT, MBO>(tt, dn);
      lds_put(dc_res + t * 64 * TF_PITCH, prow, kc, tt);
      acc_to_tile<TT, MBO>(tt, dr);
#pragma unroll
      for (int u = 0; u < TP; ++u)
        if (u == t) drc.xs[u] = tt;
      __syncthreads();
      wgrad_tile(accG, accGb, true, exA + buf * 64 * TF_PITCH, exB + buf * 64 * TF_PITCH, wave, r16, kc);
    }
    float m1[Q], m2[Q];
    group_combine<Q>(m1, S1, cg);
    group_combine<Q>(m2, S2, cg);
    // ---------------- pass 3: GroupNorm backward -> dconv (LDS resident); d gamma / d beta ----------------
    float dgam[Q], dbet[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) { dgam[j] = 0.f; dbet[j] = 0.f; }
    for (int t = 0; t < a.Tn; ++t) {
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      tconv_at<TT, NFI, MBO, TP>(acc, xc, X, row0, a, t, TF_C, kc, true, wl_conv, lane);
      LQTile<TT, NFO> dnt;
      lds_get(dnt, dc_res + t * 64 * TF_PITCH, prow, kc);      // own rows: written by this lane in pass 2
      float dc[Q];
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const float xh = fmaf(acc[j >> 2][j & 3], rs[j], sh[j]);
        const float dnv = lq_get<TT, NFO>(dnt, j / FE, j % FE);
        const float dxh = dnv * tgw[j];
        dc[j] = rs[j] * (dxh - m1[j] * inv_n - xh * m2[j] * inv_n) * vf;
        dgam[j] = fmaf(dnv, xh, dgam[j]);
        dbet[j] += dnv;
      }
      LQTile<TT, NFO> tt;
      acc_to_tile<TT, MBO>(tt, dc);
      lds_put(dc_res + t * 64 * TF_PITCH, prow, kc, tt);
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { dgam[j] += __shfl_xor(dgam[j], off, 64); dbet[j] += __shfl_xor(dbet[j], off, 64); }
    }
    if (px == 0) {
#pragma unroll
      for (int j = 0; j < Q; ++j) { my_dg[j] += dgam[j]; my_db[j] += dbet[j]; }
    }
    __syncthreads();                                            // all dconv tiles of the workgroup are resident
    // ---------------- pass 3b: conv weight gradients  dW_k += dconv[t' - (k-1)d]^T x[t'] ----------------
    for (int tp = 0; tp < a.Tn; ++tp) {
      const int buf = tp & 1;
      LQTile<TT, NFI> xt;
      xcache_get<TT, NFI, TP>(xt, xc, X, row0, a, tp, TF_C, kc, true);
      if (!valid) { xt.f[0] = bf16x8{}; xt.f[1] = bf16x8{}; }
      lds_put(exA + buf * 64 * TF_PITCH, prow, kc, xt);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ts = tp - (k - 1) * a.dil;
        if (ts < 0 || ts >= a.Tn) continue;
        wgrad_tile(accC[k], accCb, k == 1, dc_res + ts * 64 * TF_PITCH, exA + buf * 64 * TF_PITCH, wave, r16, kc);
      }
    }
    // ---------------- pass 4: dx[t'] = sum_k W_k^T dconv[t' - (k-1)d] + dres[t'] ----------------
    for (int tp = 0; tp < a.Tn; ++tp) {
      const int64_t row = row0 + (int64_t)tp * a.HW;
      f32x4 acc[MBO];
#pragma unroll
      for (int m = 0; m < MBO; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ts = tp - (k - 1) * a.dil;
        if (ts < 0 || ts >= a.Tn) continue;
        LQTile<TT, NFO> dt;
        lds_get(dt, dc_res + ts * 64 * TF_PITCH, prow, kc);
#pragma unroll
        for (int m = 0; m < MBO; ++m)
#pragma unroll
          for (int s = 0; s < NFO; ++s) acc[m] = mfma16(wl_convT[((k * MBO + m) * NFO + s) * 64 + lane], dt.f[s], acc[m]);
      }
      if (valid) {
        LQTile<TT, NFO> rt;
        xcache_get<TT, NFO, TP>(rt, drc, DX, row0, a, tp, TF_C, kc, true);   // dres kept in registers since pass 2
        float y[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) y[j] = acc[j >> 2][j & 3] + lq_get<TT, NFO>(rt, j / FE, j % FE);
        LQTile<TT, NFO> yt;
        acc_to_tile<TT, MBO>(yt, y);
        lq_store<TT, NFO>(yt, DX, row, TF_C, kc, true);
      }
    }
    __syncthreads();                                            // resident / exchange tiles are rewritten by the next workgroup tile
  }
  // ---------------- write this workgroup's slab ----------------
  float* my = slab + (int64_t)blockIdx.x * TF_SLAB;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(k * 64 + wave * 16 + kc * 4 + r) * 64 + i * 16 + r16] = accC[k][i][r];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) my[(3 * 64 + wave * 16 + kc * 4 + r) * 64 + i * 16 + r16] = accG[i][r];
  if (r16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      my[4 * 64 * 64 + wave * 16 + kc * 4 + r] = accCb[r];
      my[4 * 64 * 64 + 64 + wave * 16 + kc * 4 + r] = accGb[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * 64; i += 256) {
    const int which = i >> 6, c = i & 63;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += gacc_lds[(w * 2 + which) * 64 + c];
    my[4 * 64 * 64 + 128 + i] = s;
  }
}

// packs conv taps, gate, gate^T and conv^T taps in one launch
__global__ void tcn_fused_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ Wc, const float* __restrict__ Wg) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int k = 0; k < 3; ++k) pack_weights_lds<TT, 2>(dst + k * 4 * 2 * 64, Wc + k, 64, 64, 4, 64 * 3, 3, tid, nt);
  frag8* p = dst + 3 * 4 * 2 * 64;
  pack_weights_lds<TT, 2>(p, Wg, 64, 64, 4, 64, 1, tid, nt);
  p += 4 * 2 * 64;
  pack_weights_lds<TT, 2>(p, Wg, 64, 64, 4, 1, 64, tid, nt);
  p += 4 * 2 * 64;
  for (int k = 0; k < 3; ++k) pack_weights_lds<TT, 2>(p + k * 4 * 2 * 64, Wc + k, 64, 64, 4, 3, 64 * 3, tid, nt);   // Weff[o=ci][i=co] = Wc[co][ci][k]
}

struct TfEpi {
  float *dWc, *dWg, *dbc, *dbg, *dgam, *dbet;
  __device__ void operator()(int64_t i, float s) const {
    if (i < 3 * 4096) {
      const int k = (int)(i / 4096), co = (int)((i % 4096) / 64), ci = (int)(i % 64);
      dWc[(co * 64 + ci) * 3 + k] = s;
    } else if (i < 4 * 4096) {
      dWg[i - 3 * 4096] = s;
    } else {
      const int j = (int)(i - 4 * 4096);
      if (j < 64) dbc[j] = s; else if (j < 128) dbg[j - 64] = s; else if (j < 192) dgam[j - 128] = s; else dbet[j - 192] = s;
    }
  }
};

static unsigned tf_grid(int64_t npix) {
  int64_t g = (npix + 63) / 64;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (unsigned)g;
}
static size_t tf_pack_bytes() { return (size_t)(3 * 4 * 2 + 2 * 4 * 2 + 3 * 4 * 2) * 64 * sizeof(frag8); }

extern "C" {

// 1 when the fused backward applies to this configuration
int frl_tcn_block_bwd_fused_supported(int T, int Cin, int Cout, int G, int has_proj, int dtype) {
  if (dtype != FRL_BF16 || Cin != 64 || Cout != 64 || T > TF_T || T < 1 || has_proj) return 0;
  if (G <= 0 || 64 % G != 0 || 16 % (64 / G) != 0) return 0;
  return 1;
}

size_t frl_tcn_block_bwd_fused_workspace_bytes(int64_t npix) {
  return (size_t)tf_grid(npix) * TF_SLAB * sizeof(float) + 256 + tf_pack_bytes();
}

// x, dy, dx [B][T][HW][64] bf16; all gradients float32 in the reference layouts
int frl_tcn_block_bwd_fused(const void* x, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                            const float* gate_w, const float* gate_b, void* dx, float* d_conv_w, float* d_conv_b, float* d_gn_w,
                            float* d_gn_b, float* d_gate_w, float* d_gate_b, int64_t npix, int HW, int T, int dilation, int G, float eps,
                            void* ws, size_t ws_bytes, hipStream_t stream) {
  if (npix <= 0) return frl_fail(-2, "tcn_block_bwd_fused: empty input");
  if (!frl_tcn_block_bwd_fused_supported(T, 64, 64, G, 0, FRL_BF16)) return frl_fail(-2, "tcn_block_bwd_fused: unsupported configuration");
  if (ws_bytes < frl_tcn_block_bwd_fused_workspace_bytes(npix)) return frl_fail(-4, "tcn_block_bwd_fused: workspace too small");
  const unsigned grid = tf_grid(npix);
  float* slab = (float*)ws;
  frag8* pk = reinterpret_cast<frag8*>(reinterpret_cast<char*>(ws) + (((size_t)grid * TF_SLAB * sizeof(float) + 255) / 256) * 256);
  FRL_LAUNCH(tcn_fused_pack_kernel, dim3(32), dim3(256), 0, stream, pk, conv_w, gate_w);
  TcnArgs a{npix, HW, T, dilation, 64, 64, G, eps};
  const size_t lds = tf_pack_bytes() + (size_t)(3 * 64 + 4 * 2 * 64) * sizeof(float) +
                     (size_t)(TF_T * 64 * TF_PITCH + 4 * 64 * TF_PITCH) * sizeof(TT);
  FRL_HIP(hipFuncSetAttribute((const void*)tcn_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH(tcn_fused_bwd_kernel, dim3(grid), dim3(256), lds, stream, (const TT*)x, (const TT*)dy, (const frag8*)pk, conv_b, gn_w, gn_b, gate_b,
             (TT*)dx, slab, a);
  launch_slab_reduce<float, TfEpi>((const float*)slab, (int)grid, (int64_t)TF_SLAB,
                                   TfEpi{d_conv_w, d_gate_w, d_conv_b, d_gate_b, d_gn_w, d_gn_b}, stream);
  return frl_check_launch("tcn_block_bwd_fused");
}

}  // extern "C"
