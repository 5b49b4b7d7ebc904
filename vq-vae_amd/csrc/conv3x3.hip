// 3x3 convolution (pad 1, stride 1) over NHWC tiles as an implicit GEMM on the matrix cores.
//   Reference call sites: mix_backbone Conv2d(2C, gh, 3, pad=1)+ReLU  frl/models/spatial.py:258-261,297 (K6)
//                         gate_net Conv2d(C,gh,3)+ReLU, Conv2d(gh,C,3)+Sigmoid  spatial.py:266-272,332 (K11)
// Forward / bwd_data (same kernel on the flipped, transposed weight view):
//   workgroup = 8 x 16 output pixels x (up to) 64 output channels; 4 waves x 2 image rows each;
//   input halo tile (10 x 18 pixels x CK channels, 16-B skewed rows) and the chunk's 9-tap weights (packed MFMA A
//   fragments) live in LDS; B fragments are the lane-quarter image of the shifted pixels read by ds_read_b128.
// bwd_weight: K = pixels.  dY tile and X halo staged row-major in LDS, k-strided fragments fetched with
//   ds_read_b64_tr_b16; every wave owns one 16-channel input block x all 9 taps x 64 output channels in registers;
//   per-workgroup f32 slabs are summed in fixed order afterwards.
// Roofline: MFMA (147 456 FLOP/px at 128->64 vs 384 B/px bf16 = 384 FLOP/B, above the 312 FLOP/B balance point).
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include <type_traits>
#include "frl_reduce.hpp"

#define C3_TH 8
#define C3_TW 16
#define C3_HP (C3_TH + 2)
#define C3_WP (C3_TW + 2)

template <typename T> struct C3 {
  static constexpr int PADE = 16 / sizeof(T);   // 16-byte skew per pixel row
};

// ------------------------------------------------------------------------------------------------
// staging helpers
// ------------------------------------------------------------------------------------------------
// halo[(hy*18+hx)*pitch + c] = X[b, y0+hy-1, x0+hx-1, ck + c] (* act'(mask)) or 0
template <typename T, int TH = C3_TH, int NTHR = 256>
__device__ __forceinline__ void stage_halo(T* __restrict__ halo, int pitch, const T* __restrict__ X, const T* __restrict__ M,
                                           int mask_act, int b, int y0, int x0, int H, int W, int C, int ck, int CK, int tid) {
  constexpr int V = DT<T>::VEC;
  const int vpc = CK / V;
  const bool fast = (C % V) == 0;
  for (int i = tid; i < (TH + 2) * C3_WP * vpc; i += NTHR) {
    const int px = i / vpc, c0 = (i % vpc) * V;
    const int hy = px / C3_WP, hx = px % C3_WP;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W && ck + c0 < C) {
      const int64_t off = (((int64_t)b * H + gy) * W + gx) * C + ck + c0;
      if (fast) {
        Vec<T>::load(X + off, v);
        if (M != nullptr) {
          float m[V];
          Vec<T>::load(M + off, m);
#pragma unroll
          for (int e = 0; e < V; ++e) v[e] *= act_bwd_from_y(m[e], mask_act);
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (ck + c0 + e < C) {
            v[e] = to_f32(X[off + e]);
            if (M != nullptr) v[e] *= act_bwd_from_y(to_f32(M[off + e]), mask_act);
          }
      }
    }
    Vec<T>::store(halo + px * pitch + c0, v);
  }
}

// ------------------------------------------------------------------------------------------------
// forward / bwd_data
// ------------------------------------------------------------------------------------------------
// Packed weight image: block (oc0/4, ck/CK) holds [9 taps][4 m][NF][64] fragments; written once per call.
template <typename T, int NF>
__global__ void c3_pack_kernel(typename DT<T>::frag_t* __restrict__ dst, const float* __restrict__ Wt, int64_t w_so, int64_t w_si,
                               int tap_rev, int Cin, int Cout) {
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE, CK = 4 * q;
  const int MB = (Cout + 15) >> 4, qo = 4 * MB;
  const int nck = (Cin + CK - 1) / CK, noc = (MB + 3) / 4;
  const int per_block = 9 * 4 * NF * 64;
  const int total = noc * nck * per_block;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
    const int blk = g / per_block, i = g % per_block;
    const int oc0 = (blk / nck) * 4, ck = (blk % nck) * CK;
    const int ln = i & 63, fs = i >> 6;
    const int s = fs % NF, m = (fs / NF) & 3, tap = fs / (NF * 4);
    const int r = ln & 15, kq = ln >> 4;
    const int oc = qo * (r >> 2) + 4 * (oc0 + m) + (r & 3);
    const int tsrc = tap_rev ? 8 - tap : tap;
    const bool mok = (oc0 + m) < MB;
    if constexpr (FE == 8) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ic = ck + q * kq + 8 * s + e;
        v[e] = (mok && oc < Cout && ic < Cin) ? (bf16)Wt[oc * w_so + ic * w_si + tsrc] : (bf16)0.f;
      }
      dst[g] = v;
    } else {
      const int ic = ck + q * kq + s;
      dst[g] = (mok && oc < Cout && ic < Cin) ? Wt[oc * w_so + ic * w_si + tsrc] : 0.f;
    }
  }
}

#define C3F_TH 16      // forward / bwd_data tile: 16 x 16 pixels, 8 waves x 2 rows (2 waves per SIMD); the TH = 32 instantiation: 32 x 16, 4 rows per wave

// Halo staging split in two so that the global loads of the NEXT stage fly behind the MFMA taps of the current one:
//   c3_halo_fetch : issues every 16-byte load of the (TH+2) x 18 x CK halo into registers (zero padding resolved by predication)
//   c3_halo_commit: applies the optional activation-derivative mask and writes the registers into the LDS halo
template <typename T, int CK, int NTHR, int TH = C3F_TH, bool MASK = true>
struct C3Halo {
  static constexpr int V = DT<T>::VEC;
  static constexpr int VPC = CK / V;
  static constexpr int TOTAL = (TH + 2) * C3_WP * VPC;
  static constexpr int ITEMS = (TOTAL + NTHR - 1) / NTHR;
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type vec_t;
  vec_t x[ITEMS], m[MASK ? ITEMS : 1];
};

template <typename T, int CK, int NTHR, int TH, bool MASK>
__device__ __forceinline__ void c3_halo_fetch(C3Halo<T, CK, NTHR, TH, MASK>& h, const T* __restrict__ X, const T* __restrict__ M, int b, int y0, int x0,
                                              int H, int W, int C, int ck, int tid) {
  typedef C3Halo<T, CK, NTHR, TH, MASK> HT;
  typedef typename HT::vec_t vec_t;
#pragma unroll
  for (int u = 0; u < HT::ITEMS; ++u) {
    const int i = tid + u * NTHR;
    const int px = i / HT::VPC, c0 = (i % HT::VPC) * HT::V;
    const int hy = px / C3_WP, hx = px % C3_WP;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = i < HT::TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W && ck + c0 < C;
    const int64_t off = ok ? ((((int64_t)b * H + gy) * W + gx) * C + ck + c0) : 0;
    vec_t v = *reinterpret_cast<const vec_t*>(X + off);
    if (!ok) v = vec_t{};
    h.x[u] = v;
    if constexpr (MASK) {
      if (M != nullptr) h.m[u] = *reinterpret_cast<const vec_t*>(M + off);
    }
  }
}

template <typename T, int CK, int NTHR, int TH, bool MASK>
__device__ __forceinline__ void c3_halo_commit(const C3Halo<T, CK, NTHR, TH, MASK>& h, T* __restrict__ halo, int pitch, bool has_mask, int mask_act,
                                               int tid) {
  typedef C3Halo<T, CK, NTHR, TH, MASK> HT;
  typedef typename HT::vec_t vec_t;
#pragma unroll
  for (int u = 0; u < HT::ITEMS; ++u) {
    const int i = tid + u * NTHR;
    if (i >= HT::TOTAL) continue;
    const int px = i / HT::VPC, c0 = (i % HT::VPC) * HT::V;
    vec_t v = h.x[u];
    if constexpr (MASK) {
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < HT::V; ++e) v[e] = from_f32<T>(to_f32(v[e]) * act_bwd_from_y(to_f32(h.m[u][e]), mask_act));
      }
    }
    *reinterpret_cast<vec_t*>(halo + px * pitch + c0) = v;
  }
}

#ifdef C3_STAMPS
__device__ unsigned long long* c3_dbg;       // diagnostic build only (tools/diag/c3_stamps.hip)
#endif
// TH: tile height (16 or 32 rows of 16 pixels); a wave owns RPW = TH / 8 consecutive rows.  TH = 32 halves the barriers / halo commits per
// pixel, doubles the MFMA burst a prefetched halo has to hide behind, and (4 rows per wave) reads the pixel fragments of a column shift
// once for all three tap rows: 108 LDS fragment reads per 288 MFMAs and stage, against 12 per 16 MFMAs with two rows per wave.
template <typename T, int NF, bool EPI = false, int TH = C3F_TH, bool MASK = true>
__global__ __launch_bounds__(512) void conv3x3_kernel(const T* __restrict__ X, const T* __restrict__ Xmask, int mask_act,
                                                      const typename DT<T>::frag_t* __restrict__ Wpk,
                                                      const float* __restrict__ bias, T* __restrict__ Y, int B, int H, int W,
                                                      int Cin, int Cout, int act, const T* __restrict__ Yadd,
                                                      const T* __restrict__ Ysub, T* __restrict__ Y2, int epi_mode) {
  // epi_mode 0: Yadd (optional): Y = act(conv + bias) + Yadd.  Ysub/Y2 (optional pair): Y2 = Ysub - Y.  Both ride in the epilogue so that a
  // backward pass can fold the gradient accumulation of a tensor with two consumers into the convolution that produces one of them.
  // epi_mode 1 (gate blend of EdgeAwareSmoothingConv2D, spatial.py:332-335 with min_gate = 0): Y = gate = act(conv + bias) and
  // Y2 = Yadd + gate * Ysub (Yadd = smoothed, Ysub = residual; gate taken as rounded for its store, as the stand-alone kernel reads it).
  // epi_mode 2 / 3 (backward-data whose consumer is again a backward pass through an activation): Y = conv * act'(Yadd) with Yadd = the
  // OUTPUT of a ReLU (2) / sigmoid (3) at the same pixel -- the next layer's backward calls then need no mask pass over their input.
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE, CK = 4 * q;
  constexpr int pitch = CK + C3<T>::PADE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* halo = reinterpret_cast<T*>(smem);
  constexpr int RPW = TH / 8;
  frag_t* wl = reinterpret_cast<frag_t*>(smem + (size_t)(TH + 2) * C3_WP * pitch * sizeof(T));
  float* bias_l = reinterpret_cast<float*>(wl + 9 * 4 * NF * 64);   // [Cout padded to 16] (zeros without a bias)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  for (int i = tid; i < ((Cout + 15) & ~15); i += 512) bias_l[i] = (bias != nullptr && i < Cout) ? bias[i] : 0.f;
  const int tiles_x = (W + C3_TW - 1) / C3_TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = B * tiles_x * tiles_y;
  const int MB = (Cout + 15) >> 4, qo = 4 * MB;
  const int nck = (Cin + CK - 1) / CK, noc = (MB + 3) / 4;
  const bool fast = (Cin % DT<T>::VEC) == 0;                    // 16-byte channel vectors: register-prefetched staging
  const bool has_mask = Xmask != nullptr;
  // Persistent workgroups over a flat sequence of stages (tile, out-chunk, in-chunk); tile order is XCD-aware (the tiles of one
  // image share an XCD, halo rows hit its L2).  With a single weight block (64 -> 64) the 72 KB image is loaded once per workgroup.
  const int bid0 = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int my_tiles = bid0 < ntiles ? (ntiles - bid0 + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int per_tile = noc * nck, nstage = my_tiles * per_tile;
  int wl_block = -1;                                             // packed-weight block currently resident in LDS
  C3Halo<T, CK, 512, TH, MASK> hreg;
#ifdef C3_STAMPS
  __shared__ unsigned long long c3_ts[8][8];
  if (tid < 64) (&c3_ts[0][0])[tid] = 0ull;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define C3_ST(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) c3_ts[wave][i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define C3_ST(i) do { } while (0)
#endif
  auto stage_geom = [&](int st, int& b, int& y0, int& x0, int& oc_i, int& ck_i) {
    const int ti = st / per_tile, r = st % per_tile;
    const int bid = bid0 + ti * (int)gridDim.x;
    b = bid / (tiles_x * tiles_y);
    const int tyx = bid % (tiles_x * tiles_y);
    y0 = (tyx / tiles_x) * TH; x0 = (tyx % tiles_x) * C3_TW;
    oc_i = r / nck; ck_i = r % nck;
  };
  if (fast && nstage > 0) {
    int b, y0, x0, oc_i, ck_i;
    stage_geom(0, b, y0, x0, oc_i, ck_i);
    c3_halo_fetch<T, CK, 512, TH, MASK>(hreg, X, Xmask, b, y0, x0, H, W, Cin, ck_i * CK, tid);
  }
  f32x4 acc[RPW][4];
  for (int st = 0; st < nstage; ++st) {
    int b, y0, x0, oc_i, ck_i;
    stage_geom(st, b, y0, x0, oc_i, ck_i);
    const int oc0 = oc_i * 4, ck = ck_i * CK;
    const int nmb = (MB - oc0) < 4 ? (MB - oc0) : 4;
    if (ck_i == 0) {
#pragma unroll
      for (int t = 0; t < RPW; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      C3_ST(0);                                                  // (epilogue + loop overhead of the previous stage)
      __syncthreads();                                           // the previous stage is done with halo / wl
      C3_ST(1);
      if (fast) c3_halo_commit<T, CK, 512, TH, MASK>(hreg, halo, pitch, has_mask, mask_act, tid);
      else stage_halo<T, TH, 512>(halo, pitch, X, Xmask, mask_act, b, y0, x0, H, W, Cin, ck, CK, tid);
      // weights of this (out-chunk, in-chunk): wl[((tap*4 + m)*NF + s)*64 + lane], copied from the packed image
      const int blk = oc_i * nck + ck_i;
      if (blk != wl_block) {
        copy_frags_lds<T>(wl, Wpk + (size_t)blk * (9 * 4 * NF * 64), 9 * 4 * NF * 64, tid, 512);
        wl_block = blk;
      }
      C3_ST(2);                                                  // halo commit (waits for the prefetched loads) + weight copy
      __syncthreads();
      C3_ST(3);
      if (fast && st + 1 < nstage) {                             // next stage's halo loads fly behind this stage's MFMA taps
        int b2, y2, x2, oc2, ck2;
        stage_geom(st + 1, b2, y2, x2, oc2, ck2);
        c3_halo_fetch<T, CK, 512, TH, MASK>(hreg, X, Xmask, b2, y2, x2, H, W, Cin, ck2 * CK, tid);
      }
      // 9 taps, software-pipelined: the LDS fragments of tap+1 are requested before the MFMAs of tap are issued
      if constexpr (RPW == 2) {
        frag_t bfr[2][2][NF], afr[2][4][NF];
        auto tap_load = [&](int tap, int buf) {
          const int dy = tap / 3, dx = tap % 3;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const T* hp = halo + ((2 * wave + t + dy) * C3_WP + px + dx) * pitch + q * kc;
#pragma unroll
            for (int s = 0; s < NF; ++s) {
              if constexpr (FE == 8) bfr[buf][t][s] = *reinterpret_cast<const bf16x8*>(hp + 8 * s);
              else bfr[buf][t][s] = hp[s];
            }
          }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int s = 0; s < NF; ++s) afr[buf][m][s] = wl[((tap * 4 + m) * NF + s) * 64 + lane];
        };
        tap_load(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int cur = tap & 1;
          if (tap + 1 < 9) tap_load(tap + 1, cur ^ 1);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (m < nmb) {
#pragma unroll
              for (int s = 0; s < NF; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][m] = mfma16(afr[cur][m][s], bfr[cur][t][s], acc[t][m]);
            }
          }
        }
      } else {
        // 4 rows per wave.  For one column shift dx the pixel fragments of the six halo rows a wave touches are read ONCE into registers and
        // serve all three tap rows (a tap-major loop reads a pixel fragment pair per (tap, output row): 72 per stage instead of 36); the
        // weight fragments stream through a two-deep register ring, one 16-row output block of one tap at a time.  LDS fragment reads per
        // stage and wave: 72 + 36 = 108 instead of 144 for 288 MFMAs -- the tap loop is bound by the LDS array (256 B/clk per CU).
        frag_t bfr[RPW + 2][NF], afr[2][NF];
        auto a_load = [&](int step, int dx, int buf) {            // step = dy * 4 + m
          const int dy = step >> 2, m = step & 3;
#pragma unroll
          for (int s = 0; s < NF; ++s) afr[buf][s] = wl[(((dy * 3 + dx) * 4 + m) * NF + s) * 64 + lane];
        };
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < RPW + 2; ++r) {
            const T* hp = halo + ((RPW * wave + r) * C3_WP + px + dx) * pitch + q * kc;
#pragma unroll
            for (int s = 0; s < NF; ++s) {
              if constexpr (FE == 8) bfr[r][s] = *reinterpret_cast<const bf16x8*>(hp + 8 * s);
              else bfr[r][s] = hp[s];
            }
          }
          a_load(0, dx, 0);
#pragma unroll
          for (int step = 0; step < 12; ++step) {
            const int dy = step >> 2, m = step & 3;
            __builtin_amdgcn_sched_barrier(0);                    // (one weight block of look-ahead, not more)
            if (step + 1 < 12) a_load(step + 1, dx, (step + 1) & 1);
            if (m < nmb) {
#pragma unroll
              for (int s = 0; s < NF; ++s)
#pragma unroll
                for (int t = 0; t < RPW; ++t) acc[t][m] = mfma16(afr[step & 1][s], bfr[t + dy][s], acc[t][m]);
            }
          }
        }
      }
    }
    C3_ST(4);                                                    // prefetch issue + 9 MFMA taps
    if (ck_i != nck - 1) continue;
    // epilogue: bias comes from the LDS copy made at kernel start; when the lane's 16 output channels are contiguous
    // (Cout a multiple of 64) they leave as two 16-byte stores per pixel (bf16) instead of four 8-byte ones
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int gy = y0 + RPW * wave + t, gx = x0 + px;
      if (gy >= H || gx >= W) continue;
      T* yp = Y + (((int64_t)b * H + gy) * W + gx) * Cout;
      if (nmb == 4 && (MB & 3) == 0 && (Cout & 63) == 0) {
        const int cb = qo * kc + 4 * oc0;
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
        if constexpr (EPI) {
          // one channel vector at a time: add, store, and the second output from the value as it was rounded for the store
          const int64_t yo = (yp - Y) + cb;
#pragma unroll
          for (int j = 0; j < 16; j += DT<T>::VEC) {
            float o[DT<T>::VEC];
#pragma unroll
            for (int e = 0; e < DT<T>::VEC; ++e) o[e] = v[j + e];
            if (epi_mode >= 2) {
              float m[DT<T>::VEC];
              Vec<T>::load(Yadd + yo + j, m);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) o[e] *= act_bwd_from_y(m[e], epi_mode == 2 ? FRL_ACT_RELU : FRL_ACT_SIGMOID);
              Vec<T>::store(Y + yo + j, o);
              continue;
            }
            if (epi_mode == 1) {
              Vec<T>::store(Y + yo + j, o);
              float a[DT<T>::VEC], r[DT<T>::VEC];
              Vec<T>::load(Yadd + yo + j, a);
              Vec<T>::load(Ysub + yo + j, r);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) a[e] = fmaf(to_f32(from_f32<T>(o[e])), r[e], a[e]);
              Vec<T>::store(Y2 + yo + j, a);
              continue;
            }
            if (Yadd != nullptr) {
              float a[DT<T>::VEC];
              Vec<T>::load(Yadd + yo + j, a);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) o[e] += a[e];
            }
            Vec<T>::store(Y + yo + j, o);
            if (Y2 != nullptr) {
              float a[DT<T>::VEC];
              Vec<T>::load(Ysub + yo + j, a);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) a[e] -= to_f32(from_f32<T>(o[e]));
              Vec<T>::store(Y2 + yo + j, a);
            }
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < 16; j += DT<T>::VEC) Vec<T>::store(yp + cb + j, v + j);
        continue;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m >= nmb) continue;
        const int cb = qo * kc + 4 * (oc0 + m);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cb + r;
          const float bb = c < Cout ? bias_l[c] : 0.f;
          v[r] = act_fwd(acc[t][m][r] + bb, act);
        }
        if (EPI && (Yadd != nullptr || Y2 != nullptr)) {         // scalar route: these epilogues are rare off the fast path
          const int64_t yo = yp - Y;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (cb + r >= Cout) continue;
            if (epi_mode >= 2) {
              yp[cb + r] = from_f32<T>(v[r] * act_bwd_from_y(to_f32(Yadd[yo + cb + r]), epi_mode == 2 ? FRL_ACT_RELU : FRL_ACT_SIGMOID));
              continue;
            }
            if (epi_mode == 1) {
              const T o = from_f32<T>(v[r]);
              yp[cb + r] = o;
              Y2[yo + cb + r] = from_f32<T>(fmaf(to_f32(o), to_f32(Ysub[yo + cb + r]), to_f32(Yadd[yo + cb + r])));
              continue;
            }
            if (Yadd != nullptr) v[r] += to_f32(Yadd[yo + cb + r]);
            const T o = from_f32<T>(v[r]);
            yp[cb + r] = o;
            if (Y2 != nullptr) Y2[yo + cb + r] = from_f32<T>(to_f32(Ysub[yo + cb + r]) - to_f32(o));
          }
          continue;
        }
        if ((Cout & 3) == 0 && cb + 3 < Cout) {
          if constexpr (FE == 8) *reinterpret_cast<bf16x4*>(yp + cb) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          else *reinterpret_cast<f32x4*>(yp + cb) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (cb + r < Cout) yp[cb + r] = from_f32<T>(v[r]);
        }
      }
    }
  }
#ifdef C3_STAMPS
  __syncthreads();
  if (tid < 64) c3_dbg[(size_t)blockIdx.x * 64 + tid] = (&c3_ts[0][0])[tid];
#endif
}

// ------------------------------------------------------------------------------------------------
// bwd_weight
// ------------------------------------------------------------------------------------------------
// NWV waves per workgroup, spatial tile WTH x 16 pixels (WTH = 2 * NWV): 8 waves = 2 per SIMD hide the staging latency
template <typename T, int IBC, int OBW, int NWV>
__global__ __launch_bounds__(64 * NWV) void conv3x3_wgrad_kernel(const T* __restrict__ dY, const T* __restrict__ Ymask, int mask_act,
                                                            const T* __restrict__ X, float* __restrict__ slab, int B, int H, int W,
                                                            int Cin, int Cout, int oc_base, int tiles_per_wg, int use_tr) {
  constexpr int FE = DT<T>::FE;
  constexpr int V = DT<T>::VEC;
  constexpr int CK = IBC * 16;
  constexpr int NOG = NWV / IBC;               // wave groups along output channels
  constexpr int WTH = 2 * NWV, NTHR = 64 * NWV;
  constexpr int OCT = NOG * OBW * 16;          // output channels per pass (64)
  constexpr int pitchA = OCT + C3<T>::PADE, pitchB = CK + C3<T>::PADE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ldsA = reinterpret_cast<T*>(smem);                                   // [128][pitchA]
  T* halo = ldsA + WTH * C3_TW * pitchA;                                // [180][pitchB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kc = lane >> 4;
  const int ib = wave % IBC, og = wave / IBC;
  const int tiles_x = (W + C3_TW - 1) / C3_TW, tiles_y = (H + WTH - 1) / WTH;
  const int ntiles = B * tiles_x * tiles_y;
  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = (t_begin + tiles_per_wg) < ntiles ? (t_begin + tiles_per_wg) : ntiles;
  const int64_t slab_n = (int64_t)OCT * Cin * 9 + OCT;
  float* my = slab + (int64_t)blockIdx.x * slab_n;
  const bool fastA = (Cout % V) == 0;
  // bias gradient = column sums of the staged dY tile: one extra MFMA per fragment against a "ones" column (wave group ib == 0)
  f32x4 accb[OBW];
#pragma unroll
  for (int o = 0; o < OBW; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ck = 0; ck < Cin; ck += CK) {
    f32x4 acc[9][OBW];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < OBW; ++o) acc[tap][o] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int b = tile / (tiles_x * tiles_y), tyx = tile % (tiles_x * tiles_y);
      const int y0 = (tyx / tiles_x) * WTH, x0 = (tyx % tiles_x) * C3_TW;
      __syncthreads();
      // dY tile (masked by act'(y)), channels [oc_base, oc_base + OCT)
      for (int i = tid; i < WTH * C3_TW * (OCT / V); i += NTHR) {
        const int px = i / (OCT / V), c0 = (i % (OCT / V)) * V;
        const int gy = y0 + px / C3_TW, gx = x0 + px % C3_TW;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.f;
        if (gy < H && gx < W && oc_base + c0 < Cout) {
          const int64_t off = (((int64_t)b * H + gy) * W + gx) * Cout + oc_base + c0;
          if (fastA) {
            Vec<T>::load(dY + off, v);
            if (Ymask != nullptr) {
              float m[V];
              Vec<T>::load(Ymask + off, m);
#pragma unroll
              for (int e = 0; e < V; ++e) v[e] *= act_bwd_from_y(m[e], mask_act);
            }
          } else {
#pragma unroll
            for (int e = 0; e < V; ++e)
              if (oc_base + c0 + e < Cout) {
                v[e] = to_f32(dY[off + e]);
                if (Ymask != nullptr) v[e] *= act_bwd_from_y(to_f32(Ymask[off + e]), mask_act);
              }
          }
        }
        Vec<T>::store(ldsA + px * pitchA + c0, v);
      }
      stage_halo<T, WTH, NTHR>(halo, pitchB, X, (const T*)nullptr, 0, b, y0, x0, H, W, Cin, ck, CK, tid);
      __syncthreads();
      if constexpr (FE == 8) {
        const bf16x8 ones = (r16 == 0) ? bf16x8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f}
                                       : bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll 1
        for (int ks = 0; ks < WTH / 2; ++ks) {
          const int row = 2 * ks + (kc >> 1), col = 8 * (kc & 1);
          bf16x8 af[OBW];
#pragma unroll
          for (int o = 0; o < OBW; ++o) {
            const int ch0 = (og * OBW + o) * 16;
            const int pb = row * C3_TW + col;
            if (use_tr) {
              const T* a0 = ldsA + (pb + (r16 >> 2)) * pitchA + ch0 + 4 * (r16 & 3);
              bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
              bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitchA));
              af[o] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) af[o][j] = ldsA[(pb + j) * pitchA + ch0 + r16];
            }
          }
          if (ck == 0 && ib == 0) {
#pragma unroll
            for (int o = 0; o < OBW; ++o) accb[o] = mfma16(af[o], ones, accb[o]);
          }
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const int hb = (row + dy) * C3_WP + col + dx;
            bf16x8 bfr;
            if (use_tr) {
              const T* a0 = halo + (hb + (r16 >> 2)) * pitchB + ib * 16 + 4 * (r16 & 3);
              bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
              bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitchB));
              bfr = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) bfr[j] = halo[(hb + j) * pitchB + ib * 16 + r16];
            }
#pragma unroll
            for (int o = 0; o < OBW; ++o) acc[tap][o] = mfma16(af[o], bfr, acc[tap][o]);
          }
        }
      } else {
        const float ones = (r16 == 0) ? 1.f : 0.f;
#pragma unroll 2
        for (int ks = 0; ks < WTH * C3_TW / 4; ++ks) {
          const int pix = 4 * ks + kc;
          const int row = pix / C3_TW, col = pix % C3_TW;
          float af[OBW];
#pragma unroll
          for (int o = 0; o < OBW; ++o) af[o] = ldsA[pix * pitchA + (og * OBW + o) * 16 + r16];
          if (ck == 0 && ib == 0) {
#pragma unroll
            for (int o = 0; o < OBW; ++o) accb[o] = mfma16(af[o], ones, accb[o]);
          }
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const float bfr = halo[((row + dy) * C3_WP + col + dx) * pitchB + ib * 16 + r16];
#pragma unroll
            for (int o = 0; o < OBW; ++o) acc[tap][o] = mfma16(af[o], bfr, acc[tap][o]);
          }
        }
      }
    }
    // slab part for this input-channel chunk, tap-major so that the 16 lanes of a row write 64 contiguous bytes:
    // my[(tap * OCT + ocl) * Cin + ic]  (the reduction epilogue restores the [oc][ic][tap] order of the weight tensor)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < OBW; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ocl = (og * OBW + o) * 16 + kc * 4 + r, ic = ck + ib * 16 + r16;
          if (ic < Cin) my[((int64_t)tap * OCT + ocl) * Cin + ic] = acc[tap][o][r];
        }
  }
  if (ib == 0 && r16 == 0) {                                   // column 0 of the "ones" product holds the row sums
#pragma unroll
    for (int o = 0; o < OBW; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(int64_t)OCT * Cin * 9 + (og * OBW + o) * 16 + kc * 4 + r] = accb[o][r];
  }
}

// epilogue of the slab reduction: out[oc][j] = sum_wg slab[wg][ocl][j]; rows beyond Cout are dropped

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int g_c3_tile32 = 1;      // test hook (frl_conv3x3_tile32): 0 = always the 16-row tile

template <typename T, int NF>
static int launch_c3(const void* x, const void* xm, int mask_act, const float* w, int64_t so, int64_t si, int tap_rev,
                     const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes,
                     hipStream_t st, const void* yadd = nullptr, const void* ysub = nullptr, void* y2 = nullptr, int epi_mode = 0) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int CK = 4 * NF * DT<T>::FE;
  const int MBt = (Cout + 15) / 16;
  const size_t nfrag = (size_t)((MBt + 3) / 4) * ((Cin + CK - 1) / CK) * 9 * 4 * NF * 64;
  if (ws == nullptr || ws_bytes < nfrag * sizeof(frag_t)) return frl_fail(-4, "conv3x3: workspace too small for the packed weights");
  const frag_t* pk = (const frag_t*)ws;                            // packed weights: this call's workspace, or the caller's image cache
  {
    const FrlPackJob job = frl_pack_job_c3(w, 0, DT<T>::ID, NF, Cout, Cin, so, si, tap_rev, (int)nfrag);
    bool hit = false;
    if (void* img = frl_pack_cached(&job, 1, nfrag * sizeof(frag_t), &hit)) pk = (const frag_t*)img;
    if (!hit) FRL_LAUNCH((c3_pack_kernel<T, NF>), dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, (frag_t*)pk, w, so, si, tap_rev, Cin, Cout);
  }
  // tile height: 32 rows (4 per wave) where the image has them and the taller halo still fits the LDS next to the 9-tap weight block
  const size_t lds_w = (size_t)9 * 4 * NF * 64 * sizeof(frag_t) + (size_t)((Cout + 15) / 16 * 16) * sizeof(float);
  const size_t lds32 = (size_t)(32 + 2) * C3_WP * (CK + C3<T>::PADE) * sizeof(T) + lds_w;
  const bool tall = g_c3_tile32 && sizeof(T) == 2 && H >= 32 && lds32 <= 160 * 1024;
  const int TH = tall ? 32 : C3F_TH;
  const size_t lds = tall ? lds32 : (size_t)(C3F_TH + 2) * C3_WP * (CK + C3<T>::PADE) * sizeof(T) + lds_w;
  if (lds > 160 * 1024) return frl_fail(-3, "conv3x3: LDS budget exceeded");
  const bool epi = yadd != nullptr || y2 != nullptr;            // the epilogue extras are their own instantiation: the plain one keeps its registers
  // (the 32-row tile keeps the halo prefetch of a convolution without an activation mask -- every forward call -- to half the registers)
  auto kern = tall ? (epi ? (xm != nullptr ? conv3x3_kernel<T, NF, true, 32, true> : conv3x3_kernel<T, NF, true, 32, false>)
                          : (xm != nullptr ? conv3x3_kernel<T, NF, false, 32, true> : conv3x3_kernel<T, NF, false, 32, false>))
                   : (epi ? conv3x3_kernel<T, NF, true, C3F_TH> : conv3x3_kernel<T, NF, false, C3F_TH>);
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = B * ((H + TH - 1) / TH) * ((W + C3_TW - 1) / C3_TW);
  const int grid = tiles < 256 ? tiles : 256;                  // LDS allows one workgroup per CU: persistent over the tiles
  FRL_LAUNCH_AS("conv3x3_kernel", kern, dim3(grid), dim3(512), lds, st, (const T*)x, (const T*)xm, mask_act, pk, bias, (T*)y, B,
                     H, W, Cin, Cout, act, (const T*)yadd, (const T*)ysub, (T*)y2, epi_mode);
  return frl_check_launch("conv3x3");
}

static int c3_dispatch(const void* x, const void* xm, int mask_act, const float* w, int64_t so, int64_t si, int tap_rev,
                       const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act, int dtype, void* ws, size_t ws_bytes,
                       hipStream_t st, const void* yadd = nullptr, const void* ysub = nullptr, void* y2 = nullptr, int epi_mode = 0) {
  if (B <= 0 || H <= 0 || W <= 0) return frl_fail(-2, "conv3x3: empty input");
  if ((ysub == nullptr) != (y2 == nullptr)) return frl_fail(-2, "conv3x3: sub_from and out2 come as a pair");
  if (dtype == FRL_F32) {
    if (Cin <= 16) return launch_c3<float, 4>(x, xm, mask_act, w, so, si, tap_rev, bias, y, B, H, W, Cin, Cout, act, ws, ws_bytes, st, yadd, ysub, y2, epi_mode);
    return launch_c3<float, 8>(x, xm, mask_act, w, so, si, tap_rev, bias, y, B, H, W, Cin, Cout, act, ws, ws_bytes, st, yadd, ysub, y2, epi_mode);
  } else if (dtype == FRL_BF16) {
    if (Cin <= 32) return launch_c3<bf16, 1>(x, xm, mask_act, w, so, si, tap_rev, bias, y, B, H, W, Cin, Cout, act, ws, ws_bytes, st, yadd, ysub, y2, epi_mode);
    return launch_c3<bf16, 2>(x, xm, mask_act, w, so, si, tap_rev, bias, y, B, H, W, Cin, Cout, act, ws, ws_bytes, st, yadd, ysub, y2, epi_mode);
  }
  return frl_fail(-2, "conv3x3: bad dtype");
}

#define C3W_TH 16     // wgrad spatial tile height (8 waves x 2 rows)
static int c3_wgrad_nwg(int B, int H, int W) {
  const int tiles = B * ((H + C3W_TH - 1) / C3W_TH) * ((W + C3_TW - 1) / C3_TW);
  return tiles < 256 ? tiles : 256;
}
// bf16 band kernel (conv3x3_wgrad.hip): 8 x 32-pixel bands, LDS-DMA halo, same slab layout
bool c3v_supported(int B, int H, int W, int Cin, int Cout, int dtype);
int c3v_workgroups(int B, int H, int W);
int c3v_launch(const void* dy, const void* ym, int act, const void* x, float* ws, int B, int H, int W, int Cin, int Cout, int oc_base,
               hipStream_t st);

extern "C" {

// test / A-B hook: 0 = the forward / bwd-data kernel always takes the 16-row tile (default 1: 32-row tiles for bf16 images of >= 32 rows)
int frl_conv3x3_tile32(int on) { const int was = g_c3_tile32; g_c3_tile32 = on ? 1 : 0; return was; }


// x [B][H][W][Cin], w [Cout][Cin][3][3] f32, bias [Cout] f32 or null, y [B][H][W][Cout]
int frl_conv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act,
                    int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return c3_dispatch(x, nullptr, 0, w, (int64_t)Cin * 9, 9, 0, bias, y, B, H, W, Cin, Cout, act, dtype, ws, ws_bytes, stream);
}

// The second gate convolution of EdgeAwareSmoothingConv2D with the blend in its epilogue (spatial.py:332-335, min_gate = 0):
// gate = sigmoid(conv3x3(x, w) + bias), out = smoothed + gate * residual; smoothed / residual / gate / out [B][H][W][Cout].
int frl_conv3x3_fwd_gate_blend(const void* x, const float* w, const float* bias, const void* smoothed, const void* residual, void* gate, void* out,
                               int B, int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (smoothed == nullptr || residual == nullptr || out == nullptr) return frl_fail(-2, "conv3x3_fwd_gate_blend: smoothed, residual and out are required");
  return c3_dispatch(x, nullptr, 0, w, (int64_t)Cin * 9, 9, 0, bias, gate, B, H, W, Cin, Cout, FRL_ACT_SIGMOID, dtype, ws, ws_bytes, stream, smoothed,
                     residual, out, 1);
}

// dx = conv3x3(dy .* act'(y), w^T flipped)
int frl_conv3x3_bwd_data(const void* dy, const void* y, int act, const float* w, void* dx, int B, int H, int W, int Cin, int Cout,
                         int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return c3_dispatch(dy, act != FRL_ACT_NONE ? y : nullptr, act, w, 9, (int64_t)Cin * 9, 1, nullptr, dx, B, H, W, Cout, Cin,
                     FRL_ACT_NONE, dtype, ws, ws_bytes, stream);
}

// The same with the epilogue extras: dx = conv(...) + add (add may be null); with sub_from/out2 (a pair, may be null) also
// out2 = sub_from - dx.  None of the extras may alias dx.
int frl_conv3x3_bwd_data_fused(const void* dy, const void* y, int act, const float* w, void* dx, const void* add, const void* sub_from,
                               void* out2, int B, int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes,
                               hipStream_t stream) {
  return c3_dispatch(dy, act != FRL_ACT_NONE ? y : nullptr, act, w, 9, (int64_t)Cin * 9, 1, nullptr, dx, B, H, W, Cout, Cin,
                     FRL_ACT_NONE, dtype, ws, ws_bytes, stream, add, sub_from, out2);
}

// dx = conv^T(dy .* act'(y)) .* out_act'(out_y): the gradient handed on to a backward pass through the activation that produced out_y
// [B][H][W][Cin] (ReLU or sigmoid output at the same pixels) arrives there already masked.  out_act: FRL_ACT_RELU or FRL_ACT_SIGMOID.
int frl_conv3x3_bwd_data_outmask(const void* dy, const void* y, int act, const float* w, void* dx, const void* out_y, int out_act, int B, int H,
                                 int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (out_y == nullptr || (out_act != FRL_ACT_RELU && out_act != FRL_ACT_SIGMOID)) return frl_fail(-2, "conv3x3_bwd_data_outmask: needs out_y and a ReLU / sigmoid out_act");
  return c3_dispatch(dy, act != FRL_ACT_NONE ? y : nullptr, act, w, 9, (int64_t)Cin * 9, 1, nullptr, dx, B, H, W, Cout, Cin,
                     FRL_ACT_NONE, dtype, ws, ws_bytes, stream, out_y, nullptr, nullptr, out_act == FRL_ACT_RELU ? 2 : 3);
}

size_t frl_conv3x3_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  (void)Cout;
  int nwg = c3_wgrad_nwg(B, H, W);
  if (H % 8 == 0 && W % 32 == 0 && c3v_workgroups(B, H, W) > nwg) nwg = c3v_workgroups(B, H, W);
  return (size_t)nwg * ((size_t)64 * Cin * 9 + 64) * sizeof(float);
}

// dw [Cout][Cin][3][3], dbias [Cout] (may be null).  flags bit0: scalar LDS fragment reads (debug A/B check)
int frl_conv3x3_bwd_weight(const void* dy, const void* y, int act, const void* x, float* dw, float* dbias, int B, int H, int W,
                           int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, int flags, hipStream_t stream) {
  if (ws_bytes < frl_conv3x3_bwd_weight_workspace_bytes(B, H, W, Cin, Cout)) return frl_fail(-4, "conv3x3_bwd_weight: workspace too small");
  const bool band = c3v_supported(B, H, W, Cin, Cout, dtype) && (flags & 1) == 0;
  const int nwg = band ? c3v_workgroups(B, H, W) : c3_wgrad_nwg(B, H, W);
  const int tiles = B * ((H + C3W_TH - 1) / C3W_TH) * ((W + C3_TW - 1) / C3_TW);
  const int tpw = (tiles + nwg - 1) / nwg;
  const void* ym = act != FRL_ACT_NONE ? y : nullptr;
  const int64_t slab_n = (int64_t)64 * Cin * 9 + 64;
  for (int oc_base = 0; oc_base < Cout; oc_base += 64) {
    if (band) {
      const int rc = c3v_launch(dy, ym, act, x, (float*)ws, B, H, W, Cin, Cout, oc_base, stream);
      if (rc) return rc;
    } else if (dtype == FRL_F32) {
      const size_t lds = ((size_t)256 * (64 + 4) + (size_t)18 * 18 * (32 + 4)) * 4;
      auto kern = conv3x3_wgrad_kernel<float, 2, 1, 8>;
      FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      FRL_LAUNCH_AS("conv3x3_wgrad_kernel", kern, dim3(nwg), dim3(512), lds, stream, (const float*)dy, (const float*)ym, act, (const float*)x, (float*)ws,
                         B, H, W, Cin, Cout, oc_base, tpw, 0);
    } else if (dtype == FRL_BF16) {
      const size_t lds = ((size_t)256 * (64 + 8) + (size_t)18 * 18 * (64 + 8)) * 2;
      auto kern = conv3x3_wgrad_kernel<bf16, 4, 2, 8>;
      FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      FRL_LAUNCH_AS("conv3x3_wgrad_kernel", kern, dim3(nwg), dim3(512), lds, stream, (const bf16*)dy, (const bf16*)ym, act, (const bf16*)x, (float*)ws, B,
                         H, W, Cin, Cout, oc_base, tpw, (flags & 1) ? 0 : 1);
    } else return frl_fail(-2, "conv3x3_bwd_weight: bad dtype");
    launch_slab_reduce_deferrable<float, C3Epi>((const float*)ws, nwg, slab_n, C3Epi{64, Cin * 9, oc_base, Cout, dw, dbias}, stream,
                                                Cout <= 64);      // (deferrable only when this is the call's single slice: the slices share `ws`)
  }
  return frl_check_launch("conv3x3_bwd_weight");
}

}  // extern "C"
