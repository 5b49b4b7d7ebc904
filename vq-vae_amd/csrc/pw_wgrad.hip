// Weight gradient of pointwise / temporal-tap convolutions:
//   dW[oc][ic] = sum_p (dY .* act'(Y))[p][oc] * X[p + shift][ic],   db[oc] = sum_p (dY .* act'(Y))[p][oc]
// GEMM with K = pixels: A = dY^T, B = X, both k-strided in NHWC memory, so 64-pixel tiles are staged
// row-major in LDS and the k-strided MFMA fragments are fetched with the gfx950 transposing read
// ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32, 16x16x4 takes one f32 per lane).
// Each workgroup reduces a contiguous pixel range into a private f32 slab; a second kernel sums the
// slabs in a fixed order (bit-reproducible, no float atomics).
// Replaces autograd's conv weight-gradient for: conv2d_encoder.py:106-114, spatial.py:262-263,
// tcn.py:56-69 (3 temporal taps = 3 shifted calls), conditioning.py:55-67, representation.py:169, decoders.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_reduce.hpp"
#include <type_traits>

#define WG_KP 64

template <typename T, int OBW, int IB>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(
    const T* __restrict__ dY, const T* __restrict__ Ymask, int mask_act, const T* __restrict__ X,
    float* __restrict__ slab, int64_t P, int Cout, int Cin, int64_t rows_per_wg, int HW, int Tn, int toff,
    int use_tr, int64_t ldy) {                                // ldy = row stride of dY / Ymask (>= Cout: channel-slice calls)
  typedef typename DT<T>::frag_t frag_t;
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type vec_t;
  constexpr int FE = DT<T>::FE;
  constexpr int VEC = DT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CoP = OBW * 4 * 16, CiP = IB * 16;
  constexpr int pitchA = CoP + 8, pitchB = CiP + 8;      // elements; keeps 16-B row alignment, skews banks
  constexpr int VPRA = CoP / VEC, VPRB = CiP / VEC;       // 16-byte vectors per staged row
  constexpr int NA = (WG_KP * VPRA + 255) / 256, NB = (WG_KP * VPRB + 255) / 256;   // vectors per thread and tile
  T* ldsA = reinterpret_cast<T*>(smem);
  T* ldsB = ldsA + WG_KP * pitchA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kc = lane >> 4;

  f32x4 acc[OBW][IB], accb[OBW];
#pragma unroll
  for (int o = 0; o < OBW; ++o) {
    accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IB; ++i) acc[o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int64_t p_begin = (int64_t)blockIdx.x * rows_per_wg;
  int64_t p_end = p_begin + rows_per_wg;
  if (p_end > P) p_end = P;
  // flatA: rows of dY are shorter than a 16-byte vector multiple (the 12-channel heads) but contiguous: a whole 64-row tile is copied
  // as ONE flat run of 16-byte vectors and stays flat in LDS (row pitch = Cout; the 8-byte transposing reads only need Cout % 4 == 0;
  // the columns a fragment reads beyond Cout belong to output channels that are never written)
  const bool flatA = sizeof(T) == 2 && (Cout % VEC) != 0 && (Cout % 4) == 0 && ldy == Cout && (P % WG_KP) == 0 && (Cin % VEC) == 0;
  const bool fastA = ((Cout % VEC) == 0 && (ldy % VEC) == 0) || flatA, fastB = (Cin % VEC) == 0;
  const bool fast = fastA && fastB;                          // 16-byte channel vectors: tiles are prefetched into registers
  const int pA = flatA ? Cout : pitchA;                      // row pitch of the staged dY tile (elements)
  const int nvA = flatA ? WG_KP * Cout / VEC : WG_KP * VPRA; // 16-byte vectors of the dY tile
  const int64_t shift = (int64_t)toff * HW;

  // register images of the next tile (fast path): raw dY / mask / X vectors, fetched behind the MFMAs of the current tile
  vec_t ra[NA], rm[NA], rb[NB];
  auto fetch = [&](int64_t p0) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + u * 256;
      const int row = i / VPRA, c0 = (i % VPRA) * VEC;
      const int64_t p = p0 + row;
      const bool ok = flatA ? i < nvA : (i < WG_KP * VPRA && p < p_end && c0 < Cout);
      const int64_t off = ok ? (flatA ? p0 * (int64_t)Cout + (int64_t)i * VEC : p * ldy + c0) : 0;
      vec_t v = *reinterpret_cast<const vec_t*>(dY + off);
      if (!ok) v = vec_t{};
      ra[u] = v;
      if (Ymask != nullptr) rm[u] = *reinterpret_cast<const vec_t*>(Ymask + off);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = tid + u * 256;
      const int row = i / VPRB, c0 = (i % VPRB) * VEC;
      const int64_t p = p0 + row;
      bool ok = i < WG_KP * VPRB && p < p_end && c0 < Cin;
      if (ok && Tn > 1) {
        const int t = (int)((p / HW) % Tn) + toff;
        ok = (t >= 0 && t < Tn);
      }
      const int64_t off = ok ? (p + shift) * (int64_t)Cin + c0 : 0;
      vec_t v = *reinterpret_cast<const vec_t*>(X + off);
      if (!ok) v = vec_t{};
      rb[u] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + u * 256;
      if (i >= nvA) continue;
      const int row = i / VPRA, c0 = (i % VPRA) * VEC;
      vec_t v = ra[u];
      if (Ymask != nullptr) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = from_f32<T>(to_f32(v[e]) * act_bwd_from_y(to_f32(rm[u][e]), mask_act));
      }
      *reinterpret_cast<vec_t*>(ldsA + (flatA ? i * VEC : row * pitchA + c0)) = v;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = tid + u * 256;
      if (i >= WG_KP * VPRB) continue;
      const int row = i / VPRB, c0 = (i % VPRB) * VEC;
      *reinterpret_cast<vec_t*>(ldsB + row * pitchB + c0) = rb[u];
    }
  };
  if (fast && p_begin < p_end) fetch(p_begin);

  for (int64_t p0 = p_begin; p0 < p_end; p0 += WG_KP) {
    __syncthreads();
    if (fast) {
      commit();
    } else {
      // ---- generic staging (channel counts that are not a multiple of the vector width) ----
      for (int i = tid; i < WG_KP * VPRA; i += 256) {
        const int row = i / VPRA, c0 = (i % VPRA) * VEC;
        const int64_t p = p0 + row;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = 0.f;
        if (p < p_end && c0 < Cout) {
          const T* src = dY + p * ldy + c0;
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (c0 + e < Cout) {
              v[e] = to_f32(src[e]);
              if (Ymask != nullptr) v[e] *= act_bwd_from_y(to_f32(Ymask[p * ldy + c0 + e]), mask_act);
            }
        }
        Vec<T>::store(ldsA + row * pitchA + c0, v);
      }
      for (int i = tid; i < WG_KP * VPRB; i += 256) {
        const int row = i / VPRB, c0 = (i % VPRB) * VEC;
        const int64_t p = p0 + row;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = 0.f;
        bool ok = p < p_end && c0 < Cin;
        if (ok && Tn > 1) {
          const int t = (int)((p / HW) % Tn) + toff;
          ok = (t >= 0 && t < Tn);
        }
        if (ok) {
          const T* src = X + (p + shift) * (int64_t)Cin + c0;
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (c0 + e < Cin) v[e] = to_f32(src[e]);
        }
        Vec<T>::store(ldsB + row * pitchB + c0, v);
      }
    }
    __syncthreads();
    if (fast && p0 + WG_KP < p_end) fetch(p0 + WG_KP);       // next tile's loads fly behind this tile's MFMAs
    // ---- MFMA over the tile's pixels; the bias gradient (column sums of dY) rides along against a "ones" column ----
    if constexpr (FE == 8) {
      const bf16x8 ones = (r16 == 0) ? bf16x8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f}
                                     : bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
      for (int ks = 0; ks < WG_KP / 32; ++ks) {
        const int pix0 = ks * 32 + 8 * kc;
        bf16x8 bf[IB];
#pragma unroll
        for (int i = 0; i < IB; ++i) {
          const int ch0 = i * 16;
          if (use_tr) {
            // lane i16 of the 16-lane group supplies row (i16>>2), columns 4*(i16&3)..+3 and receives column i16
            const T* a0 = ldsB + (pix0 + (r16 >> 2)) * pitchB + ch0 + 4 * (r16 & 3);
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a0));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitchB));
            bf[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[i][j] = ldsB[(pix0 + j) * pitchB + ch0 + r16];
          }
        }
#pragma unroll
        for (int o = 0; o < OBW; ++o) {
          const int ch0 = (wave * OBW + o) * 16;
          bf16x8 af;
          if (use_tr) {
            const T* a0 = ldsA + (pix0 + (r16 >> 2)) * pA + ch0 + 4 * (r16 & 3);
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a0));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pA));
            af = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = ldsA[(pix0 + j) * pA + ch0 + r16];
          }
#pragma unroll
          for (int i = 0; i < IB; ++i) acc[o][i] = mfma16(af, bf[i], acc[o][i]);
          accb[o] = mfma16(af, ones, accb[o]);
        }
      }
    } else {
      const float ones = (r16 == 0) ? 1.f : 0.f;
#pragma unroll 4
      for (int ks = 0; ks < WG_KP / 4; ++ks) {
        const int pix = ks * 4 + kc;
        float bf[IB];
#pragma unroll
        for (int i = 0; i < IB; ++i) bf[i] = ldsB[pix * pitchB + i * 16 + r16];
#pragma unroll
        for (int o = 0; o < OBW; ++o) {
          const float af = ldsA[pix * pitchA + (wave * OBW + o) * 16 + r16];
#pragma unroll
          for (int i = 0; i < IB; ++i) acc[o][i] = mfma16(af, bf[i], acc[o][i]);
          accb[o] = mfma16(af, ones, accb[o]);
        }
      }
    }
  }
  // ---- write the private slab: [Cout*Cin] weights then [Cout] bias ----
  float* my = slab + (int64_t)blockIdx.x * ((int64_t)Cout * Cin + Cout);
#pragma unroll
  for (int o = 0; o < OBW; ++o)
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = (wave * OBW + o) * 16 + kc * 4 + r, ic = i * 16 + r16;
        if (oc < Cout && ic < Cin) my[(int64_t)oc * Cin + ic] = acc[o][i][r];
      }
  if (r16 == 0) {                                            // column 0 of the "ones" product holds the row sums
#pragma unroll
    for (int o = 0; o < OBW; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = (wave * OBW + o) * 16 + kc * 4 + r;
        if (oc < Cout) my[(int64_t)Cout * Cin + oc] = accb[o][r];
      }
  }
}

// Epilogue of the fixed-order slab reduction (frl_reduce.hpp).  dw strides let the caller scatter a tap slice of a
// [Cout][Cin][ntap] tensor: dst = dW[oc * dso + ic * dsi].

static int g_wgrad_max_wgs = 512;     // A/B at BASELINE configs[1] (tools/wgrad_bench.py): 512 beats 1024 (slab traffic) and 256 (latency hiding)
static int wgrad_nwg(int64_t P) {
  int64_t n = (P + WG_KP - 1) / WG_KP;
  if (n > g_wgrad_max_wgs) n = g_wgrad_max_wgs;
  if (n < 1) n = 1;
  return (int)n;
}

template <typename T, int OBW, int IB>
static int launch_wgrad(const void* dy, const void* ymask, int mask_act, const void* x, float* ws, int64_t P,
                        int Cout, int Cin, int HW, int Tn, int toff, int use_tr, int64_t ldy, hipStream_t st) {
  const int nwg = wgrad_nwg(P);
  int64_t rows = (P + nwg - 1) / nwg;
  rows = (rows + WG_KP - 1) / WG_KP * WG_KP;
  const size_t lds = (size_t)WG_KP * ((OBW * 64 + 8) + (IB * 16 + 8)) * sizeof(T);
  auto kern = pw_wgrad_kernel<T, OBW, IB>;
  if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH_AS("pw_wgrad_kernel", kern, dim3(nwg), dim3(256), lds, st, (const T*)dy, (const T*)ymask, mask_act, (const T*)x, ws,
                     P, Cout, Cin, rows, HW, Tn, toff, use_tr, ldy);
  return frl_check_launch("pw_wgrad");
}

template <typename T>
static int dispatch_wgrad(const void* dy, const void* ymask, int mask_act, const void* x, float* ws, int64_t P,
                          int Cout, int Cin, int HW, int Tn, int toff, int use_tr, int64_t ldy, hipStream_t st) {
  const int ob = (Cout + 63) / 64;   // oc blocks per wave (4 waves x 16 rows)
  const int ib = (Cin + 15) / 16;
#define WG_CASE(O, I) return launch_wgrad<T, O, I>(dy, ymask, mask_act, x, ws, P, Cout, Cin, HW, Tn, toff, use_tr, ldy, st)
  if (ob <= 1) {
    if (ib <= 1) WG_CASE(1, 1);
    if (ib <= 2) WG_CASE(1, 2);
    if (ib <= 4) WG_CASE(1, 4);
    if (ib <= 8) WG_CASE(1, 8);
    if (ib <= 16) WG_CASE(1, 16);
  } else if (ob <= 2) {
    if (ib <= 1) WG_CASE(2, 1);
    if (ib <= 2) WG_CASE(2, 2);
    if (ib <= 4) WG_CASE(2, 4);
    if (ib <= 8) WG_CASE(2, 8);
  } else if (ob <= 4) {
    if (ib <= 1) WG_CASE(4, 1);
    if (ib <= 2) WG_CASE(4, 2);
    if (ib <= 4) WG_CASE(4, 4);
    if (ib <= 8) WG_CASE(4, 8);
  }
#undef WG_CASE
  return frl_fail(-2, "conv1x1_bwd_weight: unsupported (Cout, Cin) combination");
}

extern "C" {

// Tuning hook: upper bound of the workgroups (= float32 slabs to reduce) a weight-gradient launch uses; returns the previous bound.
// Workspaces must be sized after the call (frl_conv1x1_bwd_weight_workspace_bytes follows the bound).
int frl_wgrad_set_max_workgroups(int n) {
  const int was = g_wgrad_max_wgs;
  if (n >= 64 && n <= 4096) g_wgrad_max_wgs = n;
  return was;
}

size_t frl_conv1x1_bwd_weight_workspace_bytes(int64_t P, int Cin, int Cout) {
  return (size_t)wgrad_nwg(P) * ((size_t)Cout * Cin + Cout) * sizeof(float);
}

// General form: rows are (b, t, hw) with T time steps; X is read at time t + toff (zero outside [0,T)).
// dW destination strides (dso, dsi) let one call fill tap `k` of a [Cout][Cin][ntap] tensor.
// flags: bit0 = use scalar LDS fragment reads instead of ds_read_b64_tr_b16 (debug / A-B check),
//        bit1 = accumulate into dbias instead of overwriting.
// (may_defer: the slab reduction may be parked for frl_defer_flush -- only the plain 1x1 call with a single output slice asks for it: the
// slices of one call share the workspace, and the taps of a dilated convolution accumulate into one bias gradient in call order)
static int conv_tap_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, int64_t dso,
                               int64_t dsi, float* dbias, int64_t P, int Cin, int Cout, int HW, int T, int toff,
                               int dtype, void* ws, size_t ws_bytes, int flags, hipStream_t stream, bool may_defer) {
  if (P <= 0) return frl_fail(-2, "bwd_weight: empty input");
  if (ws_bytes < frl_conv1x1_bwd_weight_workspace_bytes(P, Cin, Cout)) return frl_fail(-4, "bwd_weight: workspace too small");
  const void* ym = act != FRL_ACT_NONE ? y : nullptr;
  const int use_tr = (flags & 1) ? 0 : 1;
  const size_t esz = dtype == FRL_BF16 ? 2 : 4;
  // output channels are processed in slices of <= 256 (accumulator budget of the kernel); a slice reads its columns of dY / y
  // through the full row stride and fills its rows of dW / db
  for (int oc0 = 0; oc0 < Cout; oc0 += 256) {
    const int co = (Cout - oc0) < 256 ? (Cout - oc0) : 256;
    const char* dyp = (const char*)dy + (size_t)oc0 * esz;
    const char* ymp = ym ? (const char*)ym + (size_t)oc0 * esz : nullptr;
    int rc;
    if (dtype == FRL_F32) rc = dispatch_wgrad<float>(dyp, ymp, act, x, (float*)ws, P, co, Cin, HW, T, toff, 0, Cout, stream);
    else if (dtype == FRL_BF16) rc = dispatch_wgrad<bf16>(dyp, ymp, act, x, (float*)ws, P, co, Cin, HW, T, toff, use_tr, Cout, stream);
    else return frl_fail(-2, "bwd_weight: bad dtype");
    if (rc) return rc;
    const int64_t n = (int64_t)co * Cin + co;
    launch_slab_reduce_deferrable<float, WgradEpi>((const float*)ws, wgrad_nwg(P), n,
                                        WgradEpi{dw + (int64_t)oc0 * dso, dso, dsi, dbias ? dbias + oc0 : nullptr, co, Cin, (flags & 2) ? 1 : 0}, stream,
                                        may_defer && Cout <= 256 && (flags & 2) == 0);
  }
  return frl_check_launch("slab_reduce");
}

int frl_conv_tap_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, int64_t dso,
                            int64_t dsi, float* dbias, int64_t P, int Cin, int Cout, int HW, int T, int toff,
                            int dtype, void* ws, size_t ws_bytes, int flags, hipStream_t stream) {
  // (the plain 1x1 form -- dense [Cout][Cin] destination, no time shift -- is what frl_conv1x1_bwd_weight issues: deferrable as well)
  const bool plain = dso == Cin && dsi == 1 && T == 1 && toff == 0;
  return conv_tap_bwd_weight(dy, y, act, x, dw, dso, dsi, dbias, P, Cin, Cout, HW, T, toff, dtype, ws, ws_bytes, flags, stream, plain);
}

int frl_conv1x1_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, float* dbias,
                           int64_t P, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return conv_tap_bwd_weight(dy, y, act, x, dw, Cin, 1, dbias, P, Cin, Cout, 1, 1, 0, dtype, ws, ws_bytes, 0, stream, true);
}

}  // extern "C"
