// Pointwise (1x1) convolution over NHWC rows as an MFMA GEMM in "pixel-on-lane" orientation:
//   Y^T[Cout x P] = W[Cout x Cin] * X^T[Cin x P]      (A = packed weights in LDS, B = lane-quarter
//   image of a 16-pixel tile loaded straight from HBM, D = lane-quarter image of the output).
// Replaces: nn.Conv2d(.,.,1) call sites of the reference hot path
//   frl/models/conv2d_encoder.py:106-114 (K1,K4), frl/models/spatial.py:262-263 (K7,K8),
//   frl/models/representation.py:169 (K13), frl/models/conditioning.py:55-67 (K14),
//   decoder template frl/models/heads.py:128-198 (K16), and their autograd backward (bwd_data is the
//   same kernel on the transposed weight view; bwd_weight is pw_wgrad below).
// Roofline: HBM-bound (AI = 2*Cin*Cout/((Cin+Cout)*s) = 42 FLOP/B for 64->128 bf16 vs 312 balance).
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"

template <typename T, int NF, int NT>
__global__ __launch_bounds__(256) void pw_conv_kernel(
    const T* __restrict__ X, const T* __restrict__ Xmask, int mask_act, const typename DT<T>::frag_t* __restrict__ Wpk,
    const float* __restrict__ bias, T* __restrict__ Y, int64_t P, int Cin, int Cout, int act, const T* __restrict__ Yadd) {
  // Yadd (optional): Y = act(W x + bias) + Yadd -- lets a backward pass accumulate the gradient of a tensor with two consumers
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag_t* wl = reinterpret_cast<frag_t*>(smem);
  const int MB = (Cout + 15) >> 4;
  const int qo = 4 * MB;
  constexpr int q = NF * FE;
  copy_frags_lds<T>(wl, Wpk, MB * NF * 64, threadIdx.x, 256);
  float* bias_l = reinterpret_cast<float*>(wl + MB * NF * 64);          // [16 * MB] LDS copy of the bias (zeros without one)
  for (int i = threadIdx.x; i < 16 * MB; i += 256) bias_l[i] = (bias != nullptr && i < Cout) ? bias[i] : 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 15, kc = lane >> 4;
  const bool fast_in = (Cin == q * 4);
  const bool vec4_out = (Cout % 4) == 0;
  const int64_t ntile = (P + 15) >> 4;                    // 16-pixel tiles
  const int64_t ngroups = (ntile + NT - 1) / NT;          // NT tiles per wave iteration
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += (int64_t)gridDim.x * 4) {
    LQTile<T, NF> xt[NT];
    int64_t rows[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t row = (g * NT + t) * 16 + px;
      rows[t] = row;
      const int64_t rc = row < P ? row : P - 1;
      lq_load<T, NF>(xt[t], X, rc, Cin, kc, fast_in);
      if (Xmask != nullptr) {
        LQTile<T, NF> mt;
        lq_load<T, NF>(mt, Xmask, rc, Cin, kc, fast_in);
#pragma unroll
        for (int s = 0; s < NF; ++s)
#pragma unroll
          for (int e = 0; e < FE; ++e)
            lq_set<T, NF>(xt[t], s, e, lq_get<T, NF>(xt[t], s, e) * act_bwd_from_y(lq_get<T, NF>(mt, s, e), mask_act));
      }
    }
    for (int c0 = 0; c0 < MB; c0 += 4) {
      const int nmb = (MB - c0) < 4 ? (MB - c0) : 4;
      f32x4 acc[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < nmb) {
          const frag_t* wp = wl + ((c0 + m) * NF) * 64 + lane;
#pragma unroll
          for (int s = 0; s < NF; ++s) {
            const frag_t a = wp[s * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][m] = mfma16(a, xt[t].f[s], acc[t][m]);
          }
        }
      }
      // epilogue: lane (px, kc) owns output channels qo*kc + 4*(c0+m) + reg
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (rows[t] >= P) continue;
        T* yp = Y + rows[t] * (int64_t)Cout;
        if (nmb == 4 && (MB & 1) == 0 && (Cout & 15) == 0) {
          // full chunk: the lane's 16 output channels are contiguous -> 16-byte stores (2 for bf16, 4 for f32)
          const int cb = qo * kc + 4 * c0;
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = acc[t][j >> 2][j & 3] + bias_l[cb + j];
          if (act == FRL_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaxf(v[j], 0.f);
          } else if (act == FRL_ACT_SIGMOID) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = sigmoid_t<T>(v[j]);
          }
          if (Yadd != nullptr) {
#pragma unroll
            for (int j = 0; j < 16; j += DT<T>::VEC) {
              float a[DT<T>::VEC];
              Vec<T>::load(Yadd + rows[t] * (int64_t)Cout + cb + j, a);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) v[j + e] += a[e];
            }
          }
#pragma unroll
          for (int j = 0; j < 16; j += DT<T>::VEC) Vec<T>::store(yp + cb + j, v + j);
          continue;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (m >= nmb) continue;
          const int cb = qo * kc + 4 * (c0 + m);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = cb + r;
            const float b = c < Cout ? bias_l[c] : 0.f;
            v[r] = act_fwd(acc[t][m][r] + b, act);
            if (Yadd != nullptr && c < Cout) v[r] += to_f32(Yadd[rows[t] * (int64_t)Cout + c]);
          }
          if (vec4_out && cb + 3 < Cout) {
            if constexpr (FE == 8) {
              bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
              *reinterpret_cast<bf16x4*>(yp + cb) = o;
            } else {
              *reinterpret_cast<f32x4*>(yp + cb) = f32x4{v[0], v[1], v[2], v[3]};
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (cb + r < Cout) yp[cb + r] = from_f32<T>(v[r]);
          }
        }
      }
    }
  }
}

template <typename T, int NF>
static int launch_pw(const void* x, const void* xmask, int mask_act, const float* w, int64_t so, int64_t si,
                     const float* bias, void* y, int64_t P, int Cin, int Cout, int act, void* ws, size_t ws_bytes, hipStream_t st,
                     const void* yadd) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int NT = 1;
  const int MB = (Cout + 15) / 16;
  const size_t wbytes = (size_t)MB * NF * 64 * sizeof(frag_t);
  const size_t lds = wbytes + (size_t)16 * MB * sizeof(float);
  if (lds > 160 * 1024) return frl_fail(-3, "pw_conv: weights exceed LDS (Cin*Cout too large)");
  if (ws == nullptr || ws_bytes < wbytes) return frl_fail(-4, "pw_conv: workspace too small for the packed weights");
  const frag_t* pk = (const frag_t*)ws;                            // packed weights: this call's workspace, or the caller's image cache
  {
    const FrlPackJob job = frl_pack_job_pw(w, 0, DT<T>::ID, NF, Cout, Cin, MB, so, si);
    bool hit = false;
    if (void* img = frl_pack_cached(&job, 1, wbytes, &hit)) pk = (const frag_t*)img;
    if (!hit) FRL_LAUNCH((pack_weights_kernel<T, NF>), dim3((MB * NF * 64 + 255) / 256), dim3(256), 0, st, (frag_t*)pk, w, Cout, Cin, MB, so, si);
  }
  auto kern = pw_conv_kernel<T, NF, NT>;
  if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ngroups = ((P + 15) / 16 + NT - 1) / NT;
  int64_t grid = (ngroups + 3) / 4;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  FRL_LAUNCH_AS("pw_conv_kernel", kern, dim3((unsigned)grid), dim3(256), lds, st, (const T*)x, (const T*)xmask, mask_act,
             pk, bias, (T*)y, P, Cin, Cout, act, (const T*)yadd);
  return frl_check_launch("pw_conv");
}

// Dispatch over padded input width.  f32: Cp in {16,32,64,128,256} -> NF = Cp/4; bf16: Cp in {32..256} -> NF = Cp/32.
int frl_pw_dispatch(const void* x, const void* xmask, int mask_act, const float* w, int64_t so, int64_t si,
                    const float* bias, void* y, int64_t P, int Cin, int Cout, int act, int dtype, void* ws, size_t ws_bytes, hipStream_t st,
                    const void* yadd) {
  if (P <= 0) return 0;
  if (Cin < 1 || Cout < 1 || Cin > 512 || Cout > 1024) return frl_fail(-2, "pw_conv: unsupported channel count (Cin <= 512, Cout <= 1024)");
  if (dtype == FRL_F32) {
    if (Cin <= 16) return launch_pw<float, 4>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 32) return launch_pw<float, 8>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 64) return launch_pw<float, 16>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 128) return launch_pw<float, 32>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 256) return launch_pw<float, 64>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    return launch_pw<float, 128>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);   // Cout <= 64 (LDS)
  } else if (dtype == FRL_BF16) {
    if (Cin <= 32) return launch_pw<bf16, 1>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 64) return launch_pw<bf16, 2>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 128) return launch_pw<bf16, 4>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    if (Cin <= 256) return launch_pw<bf16, 8>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);
    return launch_pw<bf16, 16>(x, xmask, mask_act, w, so, si, bias, y, P, Cin, Cout, act, ws, ws_bytes, st, yadd);   // e.g. d(mix_head_B) at d = 128
  }
  return frl_fail(-2, "pw_conv: bad dtype");
}

extern "C" {

// workspace for the packed-weight image of any pointwise / 3-tap / 3x3 convolution of this library
size_t frl_conv_workspace_bytes(int Cin, int Cout, int taps) {
  // packed images pad the contraction width to a whole fragment chunk (<= 64 channels) and the output rows to blocks of four 16-row
  // MFMA tiles, in either direction (the backward-data calls swap the roles): both widths are rounded up to 64
  const size_t ci = (size_t)(Cin + 63) / 64 * 64, co = (size_t)(Cout + 63) / 64 * 64;
  return (size_t)taps * ci * co * sizeof(float) + 4096;
}

int frl_conv1x1_fwd(const void* x, const float* w, const float* bias, void* y, int64_t P, int Cin, int Cout,
                    int act, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return frl_pw_dispatch(x, nullptr, 0, w, Cin, 1, bias, y, P, Cin, Cout, act, dtype, ws, ws_bytes, stream);
}

// dx[P][Cin] = (dy .* act'(y))[P][Cout] * W[Cout][Cin];  y may be null (act none)
int frl_conv1x1_bwd_data(const void* dy, const void* y, int act, const float* w, void* dx, int64_t P, int Cin,
                         int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return frl_pw_dispatch(dy, act != FRL_ACT_NONE ? y : nullptr, act, w, 1, Cin, nullptr, dx, P, Cout, Cin,
                         FRL_ACT_NONE, dtype, ws, ws_bytes, stream);
}

// dx = (dy .* act'(y)) W + add: the accumulation of a second gradient stream into dx rides in the epilogue (add must not alias dx)
int frl_conv1x1_bwd_data_add(const void* dy, const void* y, int act, const float* w, void* dx, const void* add, int64_t P, int Cin,
                             int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return frl_pw_dispatch(dy, act != FRL_ACT_NONE ? y : nullptr, act, w, 1, Cin, nullptr, dx, P, Cout, Cin,
                         FRL_ACT_NONE, dtype, ws, ws_bytes, stream, add);
}

}  // extern "C"
