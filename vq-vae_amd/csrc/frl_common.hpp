// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of libfrlhip.
//
// Layout vocabulary used by every kernel in this directory
// --------------------------------------------------------
// * Activations are NHWC "rows": [P][C] with C contiguous, P = B*H*W (or B*T*H*W) pixels;
//   this is the tile's native (time, y, x, feature) order, so no transposes.
// * "Lane-quarter" register image of a 16-pixel tile: lane l holds pixel (l & 15) and the
//   contiguous channel quarter kc = l >> 4, i.e. channels [q*kc, q*kc + q), q = Cp/4.
//   The same image is (i) a coalesced global load/store image, (ii) the B operand of
//   v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32 with a k-permuted weight A operand,
//   (iii) the accumulator image when weight rows are permuted so that accumulator row
//   r = 4*(l>>4)+reg of block mb is output channel qo*(r>>2) + 4*mb + (r&3).
//   Pointwise conv chains therefore stay in registers with no LDS transpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { FRL_F32 = 0, FRL_BF16 = 1 };
enum { FRL_ACT_NONE = 0, FRL_ACT_RELU = 1, FRL_ACT_SIGMOID = 2 };

#define FRL_WAVE 64

// ---------------------------------------------------------------------------------------------
// dtype traits
// ---------------------------------------------------------------------------------------------
template <typename T> struct DT;
template <> struct DT<float> {
  typedef float frag_t;                   // one f32 per lane per 16x16x4 MFMA
  static constexpr int FE = 1;            // channels per fragment
  static constexpr int CPAD = 16;         // channel padding granule (4 quarters x 4)
  static constexpr int VEC = 4;           // elements per 16-byte access
  static constexpr int ID = FRL_F32;
};
template <> struct DT<bf16> {
  typedef bf16x8 frag_t;                  // eight bf16 per lane per 16x16x32 MFMA
  static constexpr int FE = 8;
  static constexpr int CPAD = 32;
  static constexpr int VEC = 8;
  static constexpr int ID = FRL_BF16;
};

__device__ __forceinline__ f32x4 mfma16(const float a, const float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const bf16x8 a, const bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == FRL_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == FRL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}
// fast sigmoid for the bf16 (performance-mode) kernels: v_exp_f32 + v_rcp_f32, ~1e-7 relative error
__device__ __forceinline__ float sigmoid_fast(float v) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
template <typename T> __device__ __forceinline__ float sigmoid_t(float v) {
  if constexpr (sizeof(T) == 2) return sigmoid_fast(v); else return 1.f / (1.f + expf(-v));
}

// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_bwd_from_y(float y, int act) {
  if (act == FRL_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == FRL_ACT_SIGMOID) return y * (1.f - y);
  return 1.f;
}

// ---------------------------------------------------------------------------------------------
// 16-byte vector load/store of VEC elements converted to/from float
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* o) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  __device__ static __forceinline__ void store(float* p, const float* o) {
    f32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
};
template <> struct Vec<bf16> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16* p, float* o) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  __device__ static __forceinline__ void store(bf16* p, const float* o) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};

// ---------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware block remap: blocks that share an XCD (b % 8 equal) get a contiguous chunk of the
// logical grid so neighbouring tiles hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ---------------------------------------------------------------------------------------------
// Lane-quarter tile I/O.  NF = fragments per lane = q / FE, q = Cp / 4.
// ---------------------------------------------------------------------------------------------
template <typename T, int NF> struct LQTile { typename DT<T>::frag_t f[NF]; };

// Channel index of element e (0..FE-1) of fragment s for quarter kc
template <typename T> __device__ __forceinline__ int lq_channel(int q, int kc, int s, int e) {
  return q * kc + s * DT<T>::FE + e;
}

// Loads the lane-quarter image of pixel row `row` (caller clamps row < P).  C is the real channel
// count; `fast` == (C == Cp) enables 16-byte accesses.  Optional multiplicative mask from an
// activation output (used to apply act'(y) to incoming gradients).
template <typename T, int NF>
__device__ __forceinline__ void lq_load(LQTile<T, NF>& t, const T* __restrict__ X, int64_t row, int C,
                                        int kc, bool fast) {
  constexpr int FE = DT<T>::FE;
  const int q = NF * FE;
  const T* p = X + row * (int64_t)C + q * kc;
  if (fast) {
    if constexpr (FE == 8) {
#pragma unroll
      for (int s = 0; s < NF; ++s) t.f[s] = *reinterpret_cast<const bf16x8*>(p + 8 * s);
    } else {
      static_assert(NF % 4 == 0, "f32 quarter must be a multiple of 4");
#pragma unroll
      for (int s = 0; s < NF; s += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p + s);
        t.f[s] = v[0]; t.f[s + 1] = v[1]; t.f[s + 2] = v[2]; t.f[s + 3] = v[3];
      }
    }
  } else {
    if constexpr (FE == 8) {
      if ((C & 3) == 0) {                                   // rows of 4 k channels (the 12-channel heads): 8-byte loads, zeros beyond C
#pragma unroll
        for (int s = 0; s < NF; ++s) {
          bf16x4 lo = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
          const int c = q * kc + 8 * s;
          if (c < C) lo = *reinterpret_cast<const bf16x4*>(p + 8 * s);
          if (c + 4 < C) hi = *reinterpret_cast<const bf16x4*>(p + 8 * s + 4);
          t.f[s] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        return;
      }
    }
#pragma unroll
    for (int s = 0; s < NF; ++s) {
      if constexpr (FE == 8) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = q * kc + 8 * s + e;
          v[e] = c < C ? p[8 * s + e] : (bf16)0.f;
        }
        t.f[s] = v;
      } else {
        const int c = q * kc + s;
        t.f[s] = c < C ? p[s] : 0.f;
      }
    }
  }
}

// element accessors (float view) of a lane-quarter tile
template <typename T, int NF>
__device__ __forceinline__ float lq_get(const LQTile<T, NF>& t, int s, int e) {
  if constexpr (DT<T>::FE == 8) return (float)t.f[s][e]; else return t.f[s];
}
template <typename T, int NF>
__device__ __forceinline__ void lq_set(LQTile<T, NF>& t, int s, int e, float v) {
  if constexpr (DT<T>::FE == 8) t.f[s][e] = (bf16)v; else t.f[s] = v;
}

// ---------------------------------------------------------------------------------------------
// Packed weight image in LDS: frag index (mb * NF + s) * 64 + lane, lane = (r = l & 15, kc = l >> 4)
//   row r of block mb  <-> output channel oc = qo * (r >> 2) + 4 * mb + (r & 3),  qo = 4 * MB
//   element e of frag s <-> input channel ic = q * kc + s * FE + e,               q = NF * FE
// Weff[oc][ic] = W[oc * so + ic * si]  (so/si express forward, transposed and tap-sliced views).
// ---------------------------------------------------------------------------------------------
template <typename T, int NF>
__device__ __forceinline__ void pack_weights_lds(typename DT<T>::frag_t* __restrict__ wl,
                                                 const float* __restrict__ W, int Cout, int Cin, int MB,
                                                 int64_t so, int64_t si, int tid, int nthreads) {
  constexpr int FE = DT<T>::FE;
  const int q = NF * FE, qo = 4 * MB;
  const int total = MB * NF * 64;
  for (int i = tid; i < total; i += nthreads) {
    const int lane = i & 63, fs = i >> 6;
    const int s = fs % NF, mb = fs / NF;
    const int r = lane & 15, kc = lane >> 4;
    const int oc = qo * (r >> 2) + 4 * mb + (r & 3);
    if constexpr (FE == 8) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ic = q * kc + 8 * s + e;
        v[e] = (oc < Cout && ic < Cin) ? (bf16)W[oc * so + ic * si] : (bf16)0.f;
      }
      wl[i] = v;
    } else {
      const int ic = q * kc + s;
      wl[i] = (oc < Cout && ic < Cin) ? W[oc * so + ic * si] : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pre-packed weights: a tiny prologue kernel writes the fragment-ordered image once per call into the caller's
// workspace (global memory, L2 resident); every workgroup then fills its LDS copy with 16-byte coalesced loads
// instead of re-gathering the f32 master weights element by element.
// ---------------------------------------------------------------------------------------------
template <typename T, int NF>
__global__ void pack_weights_kernel(typename DT<T>::frag_t* __restrict__ dst, const float* __restrict__ W, int Cout, int Cin, int MB,
                                    int64_t so, int64_t si) {
  pack_weights_lds<T, NF>(dst, W, Cout, Cin, MB, so, si, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// copies `nfrag` fragments (a multiple of 64) global -> LDS with 16-byte accesses
template <typename T>
__device__ __forceinline__ void copy_frags_lds(typename DT<T>::frag_t* __restrict__ dst, const typename DT<T>::frag_t* __restrict__ src,
                                               int nfrag, int tid, int nthreads) {
  if constexpr (DT<T>::FE == 8) {
    for (int i = tid; i < nfrag; i += nthreads) dst[i] = src[i];
  } else {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    for (int i = tid; i < nfrag / 4; i += nthreads) d4[i] = s4[i];
  }
}
