// build-flags: -fno-slp-vectorize
// (packed-f32 vector instructions issue slower than the two scalar ones they replace beside MFMAs on gfx950: MI355X_MICROARCH.md, cycle constants)
// Fused type encoder of the measured configuration (bf16):
//   conv1x1 64 -> 128 (no bias) -> GroupNorm(8) -> ReLU -> conv1x1 128 -> 64 (no bias) -> GroupNorm(8)
// (Conv2DEncoder, frl/models/conv2d_encoder.py:100-159 with the channel list of frl/config/frl_model_v0.yaml; kernel_size 1).
// GroupNorm statistics are per SAMPLE, and a sample (32 x 32 pixels x 64 channels = 128 KB) is small: ONE workgroup owns one sample
// and walks it three times, recomputing the cheap contractions on the matrix cores instead of round-tripping the 128-channel
// intermediates through HBM (the modular path moves ~0.5 GB forward and ~0.8 GB backward for them at 256 samples):
//   forward   pass A: y1 = W1 x            -> group sums of y1                        -> mean1, rstd1
//             pass B: h = relu(GN1(y1)), y2 = W2 h -> group sums of y2               -> mean2, rstd2
//             pass C: z = GN2(y2)                                                     -> HBM (the only tensor written)
//   backward  pass 1: recompute ..y2; per-channel sums of dz and dz * xhat2           -> d beta2, d gamma2, group terms of GN2
//             pass 2: dy2 = GN2'(dz), dh = W2^T dy2; per-channel sums for GN1         -> dy2, h, dh to HBM (operands of the two
//                     weight-gradient GEMMs, which stay with pw_wgrad_kernel), d beta1, d gamma1, group terms of GN1
//             pass 3: dy1 = GN1'(relu'(dh))                                            -> dy1 (overwrites dh)
// Intermediates are rounded to bf16 where the modular path stores them and a matrix-core operand needs bf16 anyway (h, dh, dy1, dy2); the
// GroupNorm inputs y1, y2 stay float32 (see ef_keep), and the normalisation uses the
// modular kernels' formulas (norm.hip: y = fma(x, rstd*gamma, beta - mean*rstd*gamma); dx = fma(A, d, fma(E, x, F))).
// Every wave keeps 16-pixel tiles in the lane-quarter register image (frl_common.hpp): the output of one contraction IS the B operand
// of the next.  Weights: packed MFMA fragment images in LDS (48 KB); the sample's rows are re-read per pass (L2 / Infinity Cache).
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include "frl_reduce.hpp"

typedef bf16 TT;
typedef bf16x8 frag8;
#define EF_C0 64
#define EF_C1 128
#define EF_C2 64
#define EF_G 8
#define EF_NW 8
#define EF_NTH (64 * EF_NW)
#define EF_W1_FRAGS (8 * 2 * 64)
#define EF_W2_FRAGS (4 * 4 * 64)
#define EF_W2T_FRAGS (8 * 2 * 64)
#define EF_SLAB 384                       // per-sample slab of the backward: d beta2[64] | d gamma2[64] | d beta1[128] | d gamma1[128]

template <int CTRL> __device__ __forceinline__ float ef_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (= the 16 pixels of a lane quarter); every lane of the row receives the total
__device__ __forceinline__ float ef_row_sum(float v) {
  v += ef_dpp<0xB1>(v);                   // quad_perm [1,0,3,2]
  v += ef_dpp<0x4E>(v);                   // quad_perm [2,3,0,1]
  v += ef_dpp<0x124>(v);                  // row_ror:4
  v += ef_dpp<0x128>(v);                  // row_ror:8
  return v;
}
__device__ __forceinline__ float ef_r(float v) { return (float)(bf16)v; }     // the bf16 rounding of a tensor the modular path stores
// The two convolution outputs (the GroupNorm inputs y1, y2) are NOT rounded here, although the modular path stores them as bf16: nothing of
// them is ever stored by this kernel, and their rounding is what the encoder's weight gradients are most sensitive to -- the GroupNorm
// backward removes the projections of the gradient onto 1 and xhat, so dW = dy^T x is a small remainder of cancelling sums, and a 2^-9
// perturbation of xhat leaks 3-9 % of max |dW1| back in (tools/diag/enc_err.py: with y1 / y2 rounded the result equals the float64 chain
// with the same two roundings to 2e-4, and is 3.4 % / 9.4 % away from the exact chain at 32x32 / 4x8 samples).
__device__ __forceinline__ float ef_keep(float v) { return v; }

// y1 (rounded) of one tile: 32 channels of the lane's pixel (channel 32 kc + j)
__device__ __forceinline__ void ef_conv1(float (&y1)[32], const LQTile<TT, 2>& xt, const frag8* __restrict__ w1, int lane) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) a = mfma16(w1[(m * 2 + s) * 64 + lane], xt.f[s], a);
#pragma unroll
    for (int r = 0; r < 4; ++r) y1[4 * m + r] = ef_keep(a[r]);
  }
}
// h = relu(fma(y1, A1, O1)) as the B operand of the second contraction; y2 (rounded): 16 channels of the pixel (channel 16 kc + j)
// (a1v / o1v: the lane quarter's 32 constants as eight 16-byte vectors: ds_read_b128 broadcasts)
__device__ __forceinline__ void ef_hidden(LQTile<TT, 4>& ht, const float (&y1)[32], const f32x4* __restrict__ a1v, const f32x4* __restrict__ o1v) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const f32x4 a = a1v[m], o = o1v[m];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * m + r;
      float v = fmaf(y1[j], a[r], o[r]);
      v = v > 0.f ? v : 0.f;
      ht.f[j >> 3][j & 7] = (bf16)v;
    }
  }
}
__device__ __forceinline__ void ef_conv2(float (&y2)[16], const LQTile<TT, 4>& ht, const frag8* __restrict__ w2, int lane) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) a = mfma16(w2[(m * 4 + s) * 64 + lane], ht.f[s], a);
#pragma unroll
    for (int r = 0; r < 4; ++r) y2[4 * m + r] = ef_keep(a[r]);
  }
}

// the same contractions with the weight fragments held in registers (forward kernel: 128 registers, no LDS traffic in the tile loops)
__device__ __forceinline__ void ef_conv1r(float (&y1)[32], const LQTile<TT, 2>& xt, const frag8 (&w1r)[16]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) a = mfma16(w1r[m * 2 + s], xt.f[s], a);
#pragma unroll
    for (int r = 0; r < 4; ++r) y1[4 * m + r] = ef_keep(a[r]);
  }
}
__device__ __forceinline__ void ef_conv2r(float (&y2)[16], const LQTile<TT, 4>& ht, const frag8 (&w2r)[16]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) a = mfma16(w2r[m * 4 + s], ht.f[s], a);
#pragma unroll
    for (int r = 0; r < 4; ++r) y2[4 * m + r] = ef_keep(a[r]);
  }
}

// group statistics of one layer from per-lane partial sums: s0 / s1 [2] = sums of y and y^2 over the lane's two groups.
// red: [EF_NW][8][2] floats; st: mean[8] | rstd[8] of this layer (LDS), also written to stats_out.
__device__ __forceinline__ void ef_group_stats(float (&s0)[2], float (&s1)[2], float* __restrict__ red, float* __restrict__ st, float* __restrict__ stats_out,
                                               double n, float eps, int tid) {
  const int lane = tid & 63, wave = tid >> 6, px = lane & 15, kc = lane >> 4;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float a = ef_row_sum(s0[g]), q = ef_row_sum(s1[g]);
    if (px == 0) { red[(wave * 8 + 2 * kc + g) * 2] = a; red[(wave * 8 + 2 * kc + g) * 2 + 1] = q; }
  }
  __syncthreads();
  if (tid < EF_G) {
    double a = 0.0, q = 0.0;
    for (int w = 0; w < EF_NW; ++w) { a += (double)red[(w * 8 + tid) * 2]; q += (double)red[(w * 8 + tid) * 2 + 1]; }
    const double m = a / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    st[tid] = mean; st[8 + tid] = rstd;
    if (stats_out != nullptr) { stats_out[tid] = mean; stats_out[8 + tid] = rstd; }
  }
  __syncthreads();
}

// LDS map (bytes): W1 | W2 | W2T (backward) | float tables
#define EF_OFF_W2 (EF_W1_FRAGS * 16)
#define EF_OFF_W2T (EF_OFF_W2 + EF_W2_FRAGS * 16)
#define EF_OFF_TAB (EF_OFF_W2T + EF_W2T_FRAGS * 16)
// tables (floats): A1[128] O1[128] A2[64] O2[64] | st1[16] st2[16] | red[EF_NW*16] | chan[4][128] (backward: per-channel totals)
#define EF_T_A1 0
#define EF_T_O1 128
#define EF_T_A2 256
#define EF_T_O2 320
#define EF_T_ST1 384
#define EF_T_ST2 400
#define EF_T_RED 416
#define EF_T_CH (EF_T_RED + EF_NW * 16)
#define EF_T_END (EF_T_CH + 4 * 128 + EF_NW * 4 * 128)
#define EF_LDS (EF_OFF_TAB + EF_T_END * 4)

__global__ __launch_bounds__(EF_NTH) void enc2_fwd_kernel(const TT* __restrict__ X, const frag8* __restrict__ Wpk, const float* __restrict__ g1,
                                                          const float* __restrict__ b1, const float* __restrict__ g2,
                                                          const float* __restrict__ b2, TT* __restrict__ Z, float* __restrict__ stats, int HW,
                                                          float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const frag8* w1 = reinterpret_cast<const frag8*>(smem);
  const frag8* w2 = reinterpret_cast<const frag8*>(smem + EF_OFF_W2);
  float* tab = reinterpret_cast<float*>(smem + EF_OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, kc = lane >> 4;
  const int b = blockIdx.x;
  const TT* xb = X + (int64_t)b * HW * EF_C0;
  copy_frags_lds<TT>(reinterpret_cast<frag8*>(smem), Wpk, EF_W1_FRAGS + EF_W2_FRAGS, tid, EF_NTH);
  __syncthreads();
  const int ntile = HW >> 4;
  frag8 w1r[16], w2r[16];                                        // this lane's fragments of both weight images stay in registers
#pragma unroll
  for (int i = 0; i < 16; ++i) { w1r[i] = w1[i * 64 + lane]; w2r[i] = w2[i * 64 + lane]; }
  // ---- pass A: statistics of y1 ----
  {
    float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
    LQTile<TT, 2> xt;
    lq_load<TT, 2>(xt, xb, (wave < ntile ? wave : 0) * 16 + px, EF_C0, kc, true);
    for (int t = wave; t < ntile; t += EF_NW) {
      LQTile<TT, 2> xn;                                            // next tile's rows fly behind this tile's work (two waves per SIMD hide little)
      lq_load<TT, 2>(xn, xb, (t + EF_NW < ntile ? t + EF_NW : t) * 16 + px, EF_C0, kc, true);
      float y1[32];
      ef_conv1r(y1, xt, w1r);
#pragma unroll
      for (int j = 0; j < 32; ++j) { s0[j >> 4] += y1[j]; s1[j >> 4] = fmaf(y1[j], y1[j], s1[j >> 4]); }
      xt = xn;
    }
    ef_group_stats(s0, s1, tab + EF_T_RED, tab + EF_T_ST1, stats + (int64_t)b * 32, (double)(EF_C1 / EF_G) * HW, eps, tid);
    for (int c = tid; c < EF_C1; c += EF_NTH) {
      const float a = tab[EF_T_ST1 + 8 + c / 16] * g1[c];
      tab[EF_T_A1 + c] = a;
      tab[EF_T_O1 + c] = fmaf(-tab[EF_T_ST1 + c / 16], a, b1[c]);
    }
    __syncthreads();
  }
  const f32x4* a1v = reinterpret_cast<const f32x4*>(tab + EF_T_A1 + 32 * kc);   // (per-channel constants are read from LDS where they are used:
  const f32x4* o1v = reinterpret_cast<const f32x4*>(tab + EF_T_O1 + 32 * kc);   //  16-byte broadcast reads; 96 registers of them would not fit)
  // ---- pass B: statistics of y2 ----
  {
    float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
    LQTile<TT, 2> xt;
    lq_load<TT, 2>(xt, xb, (wave < ntile ? wave : 0) * 16 + px, EF_C0, kc, true);
    for (int t = wave; t < ntile; t += EF_NW) {
      LQTile<TT, 2> xn;
      lq_load<TT, 2>(xn, xb, (t + EF_NW < ntile ? t + EF_NW : t) * 16 + px, EF_C0, kc, true);
      float y1[32], y2[16];
      ef_conv1r(y1, xt, w1r);
      LQTile<TT, 4> ht;
      int zo_ = 0;
      asm volatile("" : "+v"(zo_));                               // (opaque per tile: the constants are re-read, the weights may stay in registers)
      ef_hidden(ht, y1, a1v + zo_, o1v + zo_);
      ef_conv2r(y2, ht, w2r);
#pragma unroll
      for (int j = 0; j < 16; ++j) { s0[j >> 3] += y2[j]; s1[j >> 3] = fmaf(y2[j], y2[j], s1[j >> 3]); }
      xt = xn;
    }
    ef_group_stats(s0, s1, tab + EF_T_RED, tab + EF_T_ST2, stats + (int64_t)b * 32 + 16, (double)(EF_C2 / EF_G) * HW, eps, tid);
    for (int c = tid; c < EF_C2; c += EF_NTH) {
      const float a = tab[EF_T_ST2 + 8 + c / 8] * g2[c];
      tab[EF_T_A2 + c] = a;
      tab[EF_T_O2 + c] = fmaf(-tab[EF_T_ST2 + c / 8], a, b2[c]);
    }
    __syncthreads();
  }
  const f32x4* a2v = reinterpret_cast<const f32x4*>(tab + EF_T_A2 + 16 * kc);
  const f32x4* o2v = reinterpret_cast<const f32x4*>(tab + EF_T_O2 + 16 * kc);
  // ---- pass C: z ----
  TT* zb = Z + (int64_t)b * HW * EF_C2;
  LQTile<TT, 2> xt;
  lq_load<TT, 2>(xt, xb, (wave < ntile ? wave : 0) * 16 + px, EF_C0, kc, true);
  for (int t = wave; t < ntile; t += EF_NW) {
    LQTile<TT, 2> xn;
    lq_load<TT, 2>(xn, xb, (t + EF_NW < ntile ? t + EF_NW : t) * 16 + px, EF_C0, kc, true);
    float y1[32], y2[16];
    ef_conv1r(y1, xt, w1r);
    LQTile<TT, 4> ht;
    int zo_ = 0;
    asm volatile("" : "+v"(zo_));
    ef_hidden(ht, y1, a1v + zo_, o1v + zo_);
    ef_conv2r(y2, ht, w2r);
    bf16x8 o0, o1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 a = (a2v + zo_)[m], o = (o2v + zo_)[m];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bf16 v = (bf16)fmaf(y2[4 * m + r], a[r], o[r]);
        if (m < 2) o0[4 * m + r] = v; else o1[4 * (m - 2) + r] = v;
      }
    }
    bf16x8* zo = reinterpret_cast<bf16x8*>(zb + (int64_t)(t * 16 + px) * EF_C2 + 16 * kc);
    zo[0] = o0;
    zo[1] = o1;
    xt = xn;
  }
}

// Per-channel totals over the sample of two quantities per lane channel (NCH channels per lane: channel NCH*kc + j), then the group
// terms S1_g = sum_c gamma_c * tot0_c, S2_g = sum_c gamma_c * tot1_c (norm.hip).  chan: [2][C] totals (LDS), part: [EF_NW][2][C] scratch.
template <int NCH>
__device__ __forceinline__ void ef_channel_totals(const float (&p0)[NCH], const float (&p1)[NCH], float* __restrict__ part, float* __restrict__ chan,
                                                  float* __restrict__ out0, float* __restrict__ out1, int tid) {
  constexpr int C = 4 * NCH;
  const int lane = tid & 63, wave = tid >> 6, px = lane & 15, kc = lane >> 4;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const float a = ef_row_sum(p0[j]), q = ef_row_sum(p1[j]);
    if (px == 0) { part[(wave * 2 + 0) * C + NCH * kc + j] = a; part[(wave * 2 + 1) * C + NCH * kc + j] = q; }
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += EF_NTH) {
    const int which = i / C, c = i % C;
    float s = 0.f;
    for (int w = 0; w < EF_NW; ++w) s += part[(w * 2 + which) * C + c];        // fixed order
    chan[which * C + c] = s;
    (which == 0 ? out0 : out1)[c] = s;
  }
  __syncthreads();
}

// ---- backward ----
// LDS of the backward: weights | tables | two row-major staging tiles of a 128-pixel round (8 waves x 16 pixels):
//   T_A [128][64 + 8]  : dy2 (pass 2) / x (pass 3)          T_B [128][128 + 8] : h (pass 2) / dy1 (pass 3)
// from which every wave contracts its blocks of dW2 = dy2^T h (pass 2) and dW1 = dy1^T x (pass 3) over the round's pixels with
// transposing ds_read_b64_tr_b16 fragments (pixels are the k dimension); 16 + 16 accumulator registers per wave for the whole sample.
#define EF_PA 72
#define EF_PB 136
#define EF_OFF_TA (EF_OFF_TAB + EF_T_END * 4)
#define EF_OFF_TB (EF_OFF_TA + 128 * EF_PA * 2)
#define EF_LDS_BWD (EF_OFF_TB + 128 * EF_PB * 2)
#define EF_SLAB2 (EF_C2 * EF_C1 + EF_C1 * EF_C0 + EF_SLAB)      // dW2 [64][128] | dW1 [128][64] | d beta2 | d gamma2 | d beta1 | d gamma1

__device__ __forceinline__ bf16x8 ef_tr_frag(const TT* tile, int pitch, int pix0, int ch0, int r16) {
  const TT* a0 = tile + (pix0 + (r16 >> 2)) * pitch + ch0 + 4 * (r16 & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitch));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int NF>
__device__ __forceinline__ void ef_tile_put(TT* tile, int pitch, int prow, int kc, const LQTile<TT, NF>& t) {
  bf16x8* p = reinterpret_cast<bf16x8*>(tile + prow * pitch + NF * 8 * kc);
#pragma unroll
  for (int s = 0; s < NF; ++s) p[s] = t.f[s];
}

__global__ __launch_bounds__(EF_NTH) void enc2_bwd_kernel(const TT* __restrict__ X, const TT* __restrict__ DZ, const frag8* __restrict__ Wpk,
                                                          const float* __restrict__ g1, const float* __restrict__ b1,
                                                          const float* __restrict__ g2, const float* __restrict__ b2,
                                                          const float* __restrict__ stats, float* __restrict__ slab, int HW) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const frag8* w1 = reinterpret_cast<const frag8*>(smem);
  const frag8* w2 = reinterpret_cast<const frag8*>(smem + EF_OFF_W2);
  const frag8* w2t = reinterpret_cast<const frag8*>(smem + EF_OFF_W2T);
  float* tab = reinterpret_cast<float*>(smem + EF_OFF_TAB);
  float* chan = tab + EF_T_CH;                                    // [2][128] channel totals of the layer in flight
  float* part = chan + 4 * 128;                                   // [EF_NW][2][128]
  TT* tA = reinterpret_cast<TT*>(smem + EF_OFF_TA);
  TT* tB = reinterpret_cast<TT*>(smem + EF_OFF_TB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, kc = lane >> 4;
  const int b = blockIdx.x;
  const TT* xb = X + (int64_t)b * HW * EF_C0;
  const TT* dzb = DZ + (int64_t)b * HW * EF_C2;
  float* my = slab + (int64_t)b * EF_SLAB2;
  float* myg = my + EF_C2 * EF_C1 + EF_C1 * EF_C0;                 // d beta2[64] | d gamma2[64] | d beta1[128] | d gamma1[128]
  copy_frags_lds<TT>(reinterpret_cast<frag8*>(smem), Wpk, EF_W1_FRAGS + EF_W2_FRAGS + EF_W2T_FRAGS, tid, EF_NTH);
  if (tid < 32) tab[EF_T_ST1 + tid] = stats[(int64_t)b * 32 + tid];   // mean1[8] rstd1[8] mean2[8] rstd2[8]
  __syncthreads();
  for (int c = tid; c < EF_C1 + EF_C2; c += EF_NTH) {
    if (c < EF_C1) {
      const float a = tab[EF_T_ST1 + 8 + c / 16] * g1[c];
      tab[EF_T_A1 + c] = a;
      tab[EF_T_O1 + c] = fmaf(-tab[EF_T_ST1 + c / 16], a, b1[c]);
    } else {
      const int c2 = c - EF_C1;
      const float a = tab[EF_T_ST2 + 8 + c2 / 8] * g2[c2];
      tab[EF_T_A2 + c2] = a;
      tab[EF_T_O2 + c2] = fmaf(-tab[EF_T_ST2 + c2 / 8], a, b2[c2]);
    }
  }
  __syncthreads();
  const int ntile = HW >> 4;
  const int nround = (ntile + EF_NW - 1) / EF_NW;
  const float* a1q = tab + EF_T_A1 + 32 * kc;                      // (read from LDS where used: 64 registers would not fit beside the sums)
  const float* o1q = tab + EF_T_O1 + 32 * kc;
  const float* a2q = tab + EF_T_A2 + 16 * kc;
  const float mu2[2] = {tab[EF_T_ST2 + 2 * kc], tab[EF_T_ST2 + 2 * kc + 1]}, rs2[2] = {tab[EF_T_ST2 + 8 + 2 * kc], tab[EF_T_ST2 + 8 + 2 * kc + 1]};
  const float mu1[2] = {tab[EF_T_ST1 + 2 * kc], tab[EF_T_ST1 + 2 * kc + 1]}, rs1[2] = {tab[EF_T_ST1 + 8 + 2 * kc], tab[EF_T_ST1 + 8 + 2 * kc + 1]};

  // forward chain of one tile from x: y1 (rounded), h (B operand), y2 (rounded); the weight fragments stream from LDS (lw: opaque lane)
  auto chain = [&](const LQTile<TT, 2>& xt, float (&y1)[32], LQTile<TT, 4>& ht, float (&y2)[16], int lw) {
    ef_conv1(y1, xt, w1, lw);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float v = fmaf(y1[j], a1q[j], o1q[j]);
      v = v > 0.f ? v : 0.f;
      ht.f[j >> 3][j & 7] = (bf16)v;
    }
    ef_conv2(y2, ht, w2, lw);
  };

  // ---- pass 1: per-channel sums of d = dz and d * xhat2 ----
  {
    float p0[16], p1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { p0[j] = 0.f; p1[j] = 0.f; }
    for (int t = wave; t < ntile; t += EF_NW) {
      int lw = lane;
      asm volatile("" : "+v"(lw));                                // (opaque per tile: the 48 KB of weight fragments stay in LDS, not in 192 registers)
      LQTile<TT, 2> xt, dt;
      lq_load<TT, 2>(xt, xb, t * 16 + px, EF_C0, kc, true);
      lq_load<TT, 2>(dt, dzb, t * 16 + px, EF_C2, kc, true);
      float y1[32], y2[16];
      LQTile<TT, 4> ht;
      chain(xt, y1, ht, y2, lw);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float d = (float)dt.f[j >> 3][j & 7];
        const float xh = (y2[j] - mu2[j >> 3]) * rs2[j >> 3];
        p0[j] += d;
        p1[j] = fmaf(d, xh, p1[j]);
      }
    }
    ef_channel_totals<16>(p0, p1, part, chan, myg, myg + 64, tid);   // d beta2 | d gamma2 contributions of this sample
  }
  // group terms of GN2 for the lane's two groups: E, F of dx = fma(A, d, fma(E, x, F))
  float e2[2], f2[2];
  {
    const float inv_n = 1.f / ((float)(EF_C2 / EF_G) * (float)HW);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float S1 = 0.f, S2 = 0.f;
      for (int c = (2 * kc + g) * 8; c < (2 * kc + g + 1) * 8; ++c) { S1 = fmaf(g2[c], chan[c], S1); S2 = fmaf(g2[c], chan[64 + c], S2); }
      e2[g] = -rs2[g] * rs2[g] * S2 * inv_n;
      f2[g] = -rs2[g] * S1 * inv_n - e2[g] * mu2[g];
    }
  }
  __syncthreads();                                                // (chan is reused by pass 2)

  // dy2 of a tile (GN2 backward) and dh = W2^T dy2 (rounded like the modular path's stored tensors); d1 = relu'(.) dh
  auto back2 = [&](const LQTile<TT, 2>& dt, const float (&y1)[32], const float (&y2)[16], LQTile<TT, 2>& gt, float (&d1)[32], int lw) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float d = (float)dt.f[j >> 3][j & 7];
      gt.f[j >> 3][j & 7] = (bf16)fmaf(a2q[j], d, fmaf(e2[j >> 3], y2[j], f2[j >> 3]));
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) a = mfma16(w2t[(m * 2 + s) * 64 + lw], gt.f[s], a);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * m + r;
        float v = ef_r(a[r]);
        if (!(fmaf(y1[j], a1q[j], o1q[j]) > 0.f)) v = 0.f;
        d1[j] = v;
      }
    }
  };

  // ---- pass 2: per-channel sums of d1 and d1 * xhat1; dW2 += dy2^T h over the round's 128 pixels ----
  const int ob2 = wave & 3, ig2 = wave >> 2;                       // dW2 blocks of this wave: output block ob2, input blocks 4 ig2 .. 4 ig2 + 3
  {
    f32x4 acc2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float p0[32], p1[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { p0[j] = 0.f; p1[j] = 0.f; }
    for (int r = 0; r < nround; ++r) {
      const int t = r * EF_NW + wave;
      const bool valid = t < ntile;                              // (wave-uniform) rounds are walked by all waves: they share the barriers
      int lw = lane;
      asm volatile("" : "+v"(lw));
      LQTile<TT, 2> gt;
      LQTile<TT, 4> ht;
      if (valid) {
        LQTile<TT, 2> xt, dt;
        lq_load<TT, 2>(xt, xb, t * 16 + px, EF_C0, kc, true);
        lq_load<TT, 2>(dt, dzb, t * 16 + px, EF_C2, kc, true);
        float y1[32], y2[16], d1[32];
        chain(xt, y1, ht, y2, lw);
        back2(dt, y1, y2, gt, d1, lw);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const float xh = (y1[j] - mu1[j >> 4]) * rs1[j >> 4];
          p0[j] += d1[j];
          p1[j] = fmaf(d1[j], xh, p1[j]);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) gt.f[s] = frag8{};
#pragma unroll
        for (int s = 0; s < 4; ++s) ht.f[s] = frag8{};
      }
      ef_tile_put<2>(tA, EF_PA, wave * 16 + px, kc, gt);
      ef_tile_put<4>(tB, EF_PB, wave * 16 + px, kc, ht);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 af = ef_tr_frag(tA, EF_PA, 32 * ks + 8 * kc, 16 * ob2, px);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] = mfma16(af, ef_tr_frag(tB, EF_PB, 32 * ks + 8 * kc, 16 * (4 * ig2 + i), px), acc2[i]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) my[(16 * ob2 + 4 * kc + rr) * EF_C1 + 16 * (4 * ig2 + i) + px] = acc2[i][rr];
    ef_channel_totals<32>(p0, p1, part, chan, myg + 128, myg + 256, tid);   // d beta1 | d gamma1
  }
  float e1[2], f1[2];
  {
    const float inv_n = 1.f / ((float)(EF_C1 / EF_G) * (float)HW);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float S1 = 0.f, S2 = 0.f;
      for (int c = (2 * kc + g) * 16; c < (2 * kc + g + 1) * 16; ++c) { S1 = fmaf(g1[c], chan[c], S1); S2 = fmaf(g1[c], chan[128 + c], S2); }
      e1[g] = -rs1[g] * rs1[g] * S2 * inv_n;
      f1[g] = -rs1[g] * S1 * inv_n - e1[g] * mu1[g];
    }
  }
  // ---- pass 3: dy1 = fma(A1, d1, fma(E1, y1, F1)) (the chain and dh are recomputed: nothing was parked in HBM); dW1 += dy1^T x ----
  {
    f32x4 acc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* my1 = my + EF_C2 * EF_C1;
    for (int r = 0; r < nround; ++r) {
      const int t = r * EF_NW + wave;
      const bool valid = t < ntile;
      int lw = lane;
      asm volatile("" : "+v"(lw));
      LQTile<TT, 2> xt;
      LQTile<TT, 4> yt;                                            // dy1 of the tile
      if (valid) {
        LQTile<TT, 2> dt, gt;
        lq_load<TT, 2>(xt, xb, t * 16 + px, EF_C0, kc, true);
        lq_load<TT, 2>(dt, dzb, t * 16 + px, EF_C2, kc, true);
        float y1[32], y2[16], d1[32];
        LQTile<TT, 4> ht;
        chain(xt, y1, ht, y2, lw);
        back2(dt, y1, y2, gt, d1, lw);
#pragma unroll
        for (int j = 0; j < 32; ++j) yt.f[j >> 3][j & 7] = (bf16)fmaf(a1q[j], d1[j], fmaf(e1[j >> 4], y1[j], f1[j >> 4]));
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) xt.f[s] = frag8{};
#pragma unroll
        for (int s = 0; s < 4; ++s) yt.f[s] = frag8{};
      }
      ef_tile_put<2>(tA, EF_PA, wave * 16 + px, kc, xt);
      ef_tile_put<4>(tB, EF_PB, wave * 16 + px, kc, yt);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 af = ef_tr_frag(tB, EF_PB, 32 * ks + 8 * kc, 16 * wave, px);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[i] = mfma16(af, ef_tr_frag(tA, EF_PA, 32 * ks + 8 * kc, 16 * i, px), acc1[i]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) my1[(16 * wave + 4 * kc + rr) * EF_C0 + 16 * i + px] = acc1[i][rr];
  }
}


static const frag8* enc_packed(const float* w1, const float* w2, int bwd, frag8* ws_pk, hipStream_t st) {
  FrlPackJob jobs[3];
  size_t off = 0;
  jobs[0] = frl_pack_job_pw(w1, off, FRL_BF16, 2, EF_C1, EF_C0, 8, EF_C0, 1);
  off += (size_t)EF_W1_FRAGS * sizeof(frag8);
  jobs[1] = frl_pack_job_pw(w2, off, FRL_BF16, 4, EF_C2, EF_C1, 4, EF_C1, 1);
  off += (size_t)EF_W2_FRAGS * sizeof(frag8);
  int n = 2;
  if (bwd) {
    jobs[2] = frl_pack_job_pw(w2, off, FRL_BF16, 2, EF_C1, EF_C2, 8, 1, EF_C1);      // W2^T: [c1][c2] = W2[c2][c1]
    off += (size_t)EF_W2T_FRAGS * sizeof(frag8);
    n = 3;
  }
  bool hit = false;
  frag8* pk = ws_pk;
  if (void* img = frl_pack_cached(jobs, n, off, &hit)) pk = (frag8*)img;
  if (!hit) {
    FRL_LAUNCH((pack_weights_kernel<TT, 2>), dim3(4), dim3(256), 0, st, pk, w1, EF_C1, EF_C0, 8, (int64_t)EF_C0, (int64_t)1);
    FRL_LAUNCH((pack_weights_kernel<TT, 4>), dim3(4), dim3(256), 0, st, pk + EF_W1_FRAGS, w2, EF_C2, EF_C1, 4, (int64_t)EF_C1, (int64_t)1);
    if (bwd) FRL_LAUNCH((pack_weights_kernel<TT, 2>), dim3(4), dim3(256), 0, st, pk + EF_W1_FRAGS + EF_W2_FRAGS, w2, EF_C1, EF_C2, 8, (int64_t)1, (int64_t)EF_C1);
  }
  return pk;
}

extern "C" {

// 1 when the fused kernels cover (channel list, groups, pixels per sample, dtype); else the caller composes conv1x1 + groupnorm calls.
int frl_encoder2_supported(int C0, int C1, int C2, int G1, int G2, int HW, int dtype) {
  return (dtype == FRL_BF16 && C0 == EF_C0 && C1 == EF_C1 && C2 == EF_C2 && G1 == EF_G && G2 == EF_G && HW > 0 && (HW % 16) == 0) ? 1 : 0;
}

size_t frl_encoder2_workspace_bytes(int B) {
  return ((size_t)B * EF_SLAB2 * sizeof(float) + 255) / 256 * 256 + (size_t)(EF_W1_FRAGS + EF_W2_FRAGS + EF_W2T_FRAGS) * sizeof(frag8);
}

// x [B][HW][64] bf16 -> z [B][HW][64] bf16; w1 [128][64], w2 [64][128] f32 (no biases); gamma / beta of the two GroupNorms;
// stats [B][32] f32 = mean1[8] rstd1[8] mean2[8] rstd2[8] (saved for the backward).
int frl_encoder2_fwd(const void* x, const float* w1, const float* g1, const float* b1, const float* w2, const float* g2, const float* b2,
                     void* z, float* stats, int B, int HW, float eps, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (B <= 0 || HW <= 0 || (HW % 16) != 0) return frl_fail(-2, "encoder2_fwd: need B > 0 and HW a positive multiple of 16");
  if (ws_bytes < frl_encoder2_workspace_bytes(B)) return frl_fail(-4, "encoder2_fwd: workspace too small");
  char* wsb = (char*)ws;
  const frag8* pk = enc_packed(w1, w2, 0, reinterpret_cast<frag8*>(wsb + ((size_t)B * EF_SLAB2 * sizeof(float) + 255) / 256 * 256), stream);
  auto kern = enc2_fwd_kernel;
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EF_LDS));
  FRL_LAUNCH(enc2_fwd_kernel, dim3(B), dim3(EF_NTH), EF_LDS, stream, (const TT*)x, pk, g1, b1, g2, b2, (TT*)z, stats, HW, eps);
  return frl_check_launch("encoder2_fwd");
}

// Backward of frl_encoder2_fwd with respect to the parameters (the encoder's input is data: no dx): dw1 [128][64], dw2 [64][128] and the
// GroupNorm parameter gradients.  No intermediate tensor reaches HBM; per-sample float32 slabs are summed in a fixed order.
int frl_encoder2_bwd(const void* x, const void* dz, const float* w1, const float* g1, const float* b1, const float* w2, const float* g2,
                     const float* b2, const float* stats, float* dw1, float* dg1, float* db1, float* dw2, float* dg2, float* db2,
                     int B, int HW, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (B <= 0 || HW <= 0 || (HW % 16) != 0) return frl_fail(-2, "encoder2_bwd: need B > 0 and HW a positive multiple of 16");
  if (ws_bytes < frl_encoder2_workspace_bytes(B)) return frl_fail(-4, "encoder2_bwd: workspace too small");
  char* wsb = (char*)ws;
  const frag8* pk = enc_packed(w1, w2, 1, reinterpret_cast<frag8*>(wsb + ((size_t)B * EF_SLAB2 * sizeof(float) + 255) / 256 * 256), stream);
  auto kern = enc2_bwd_kernel;
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EF_LDS_BWD));
  FRL_LAUNCH(enc2_bwd_kernel, dim3(B), dim3(EF_NTH), EF_LDS_BWD, stream, (const TT*)x, (const TT*)dz, pk, g1, b1, g2, b2, stats, (float*)ws, HW);
  launch_slab_reduce_deferrable<float, EncEpi>((const float*)ws, B, (int64_t)EF_SLAB2, EncEpi{dw2, dw1, db2, dg2, db1, dg1}, stream);
  return frl_check_launch("encoder2_bwd");
}

}  // extern "C"
