// build-flags: -fno-slp-vectorize
// (packed-f32 vector instructions issue slower than the two scalar ones they replace beside MFMAs on gfx950: MI355X_MICROARCH.md, cycle constants)
// Fused decoder + reconstruction loss for the hot configuration (bf16, hidden 128, 64 output features, latent <= 64 ch):
//   xhat = W2 relu(W1 z + b1) + b2 ;  L = mean_valid (xhat - x)^2
// Decoder definition: SURVEY.md 8a row a12 (template Conv2DHead, frl/models/heads.py:128-198; loss frl/losses/reconstruction.py:95-139).
// Forward: one pass over (z, x): the 128-channel hidden tensor and xhat never reach HBM (the modular path writes/reads
// ~0.9 GB for them on the phase path at cfg2); xhat is written only when the caller asks for it.
// Backward: recomputes the chain per pixel in registers, produces dz, and contracts the weight gradients over pixels inside
// the kernel (LDS tiles + ds_read_b64_tr_b16 fragments; every wave owns a block-row of dW2 / dW1), as tcn_fused.hip does.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include "frl_reduce.hpp"

typedef bf16 TT;
typedef bf16x8 frag8;
#define DF_H 128
#define DF_F 64
#define DF_GRID_MAX 1024

__device__ __forceinline__ bf16x8 tr_frag_p(const TT* tile, int pitch, int pix0, int ch0, int r16) {
  const TT* a0 = tile + (pix0 + (r16 >> 2)) * pitch + ch0 + 4 * (r16 & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitch));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NF>
__device__ __forceinline__ void tile_put(TT* tile, int pitch, int prow, int kc, const LQTile<TT, NF>& t) {
  bf16x8* p = reinterpret_cast<bf16x8*>(tile + prow * pitch + NF * 8 * kc);
#pragma unroll
  for (int s = 0; s < NF; ++s) p[s] = t.f[s];
}

// per-pixel forward chain; returns hidden (32 ch / lane, post-ReLU) and xhat (16 ch / lane)
template <int NFZ>
__device__ __forceinline__ void dec_chain(float (&h)[32], float (&xh)[16], const LQTile<TT, NFZ>& zt, const frag8* __restrict__ w1,
                                          const frag8* __restrict__ w2, const float* __restrict__ b1q, const float* __restrict__ b2q,
                                          LQTile<TT, 4>& ht, int lane) {
  f32x4 hacc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    hacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NFZ; ++s) hacc[m] = mfma16(w1[(m * NFZ + s) * 64 + lane], zt.f[s], hacc[m]);
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float v = hacc[j >> 2][j & 3] + b1q[j];
    h[j] = v > 0.f ? v : 0.f;
    ht.f[j >> 3][j & 7] = (bf16)h[j];
  }
  f32x4 xacc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    xacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) xacc[m] = mfma16(w2[(m * 4 + s) * 64 + lane], ht.f[s], xacc[m]);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) xh[j] = xacc[j >> 2][j & 3] + b2q[j];
}

// ------------------------------------------------------------------------------------------------ forward
template <int NFZ>
__global__ __launch_bounds__(256, 4) void dec_mse_fwd_kernel(const TT* __restrict__ Z, const frag8* __restrict__ Wpk, const float* __restrict__ b1,
                                                          const float* __restrict__ b2, const TT* __restrict__ TGT,
                                                          const uint8_t* __restrict__ mask, TT* __restrict__ XHAT, int64_t P, int Cz,
                                                          double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* w1 = reinterpret_cast<frag8*>(smem);                 // [8][NFZ][64]
  frag8* w2 = w1 + 8 * NFZ * 64;                              // [4][4][64]
  float* tb = reinterpret_cast<float*>(w2 + 16 * 64);         // b1[128] | b2[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<TT>(w1, Wpk, (8 * NFZ + 16) * 64, tid, 256);
  for (int i = tid; i < 192; i += 256) tb[i] = i < 128 ? b1[i] : b2[i - 128];
  __syncthreads();
  const float* b1q = tb + 32 * kc;
  const float* b2q = tb + 128 + 16 * kc;
  const bool fastz = (Cz == 32 * NFZ);
  float sq = 0.f, cnt = 0.f;
  const int64_t ntile = (P + 15) >> 4;
  // the next tile's rows are requested before this tile's chain (a wave is otherwise parked for a full memory latency per tile)
  const int64_t tstep = (int64_t)gridDim.x * 4;
  LQTile<TT, NFZ> zt, zn;
  LQTile<TT, 2> tt, tn;
  {
    const int64_t t0 = (int64_t)blockIdx.x * 4 + wave;
    const int64_t r0 = (t0 < ntile ? t0 : 0) * 16 + px;
    const int64_t rc = r0 < P ? r0 : P - 1;
    lq_load<TT, NFZ>(zt, Z, rc, Cz, kc, fastz);
    lq_load<TT, 2>(tt, TGT, rc, DF_F, kc, true);
  }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += tstep) {
    int64_t row = tile * 16 + px;
    bool valid = row < P;
    if (!valid) row = P - 1;
    if (valid && mask != nullptr) valid = mask[row] != 0;
    {
      const int64_t tnx = tile + tstep < ntile ? tile + tstep : tile;
      const int64_t rn = tnx * 16 + px;
      const int64_t rc = rn < P ? rn : P - 1;
      lq_load<TT, NFZ>(zn, Z, rc, Cz, kc, fastz);
      lq_load<TT, 2>(tn, TGT, rc, DF_F, kc, true);
    }
    float h[32], xh[16];
    LQTile<TT, 4> ht;
    int lw = lane;
    asm volatile("" : "+v"(lw));                                // (opaque per tile: the weight fragments stay in LDS instead of 96 hoisted registers)
    dec_chain<NFZ>(h, xh, zt, w1, w2, b1q, b2q, ht, lw);
    if (valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { const float d = xh[j] - (float)tt.f[j >> 3][j & 7]; sq = fmaf(d, d, sq); }
      cnt += 16.f;
    }
    if (XHAT != nullptr && tile * 16 + px < P) {
      bf16x8* xo = reinterpret_cast<bf16x8*>(XHAT + row * DF_F + 16 * kc);
      bf16x8 o0, o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) { o0[j] = (bf16)xh[j]; o1[j] = (bf16)xh[8 + j]; }
      xo[0] = o0;
      xo[1] = o1;
    }
    zt = zn;
    tt = tn;
  }
  double sd = wave_sum_d((double)sq), cd = wave_sum_d((double)cnt);
  __shared__ double red[8];
  if (lane == 0) { red[wave] = sd; red[4 + wave] = cd; }
  __syncthreads();
  if (tid == 0) {
    partial[blockIdx.x * 2 + 0] = red[0] + red[1] + red[2] + red[3];
    partial[blockIdx.x * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(256) void dec_mse_finalize_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double rs[256], rc[256];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  rs[threadIdx.x] = s; rc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = rc[0] > 0.0 ? (float)(rs[0] / rc[0]) : 0.f; out[1] = (float)rc[0]; }
}

// ------------------------------------------------------------------------------------------------ backward
// Barrier of one 4-wave subgroup through a counter in LDS (the scheme of tcn_hot_bwd4.hip): every wave adds 1 and waits until the counter has
// reached 4 x the number of barriers it has passed; all waves of the workgroup are resident and a subgroup's waves run the same rounds.
__device__ __forceinline__ void df_sg_sync(unsigned* bar, unsigned& gen, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  gen += 4u;
  if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    const unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(seen - gen) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
#define DF_SG_SHARE0 19      // subgroup 0's share of a workgroup's rounds in 32nds (the older waves of a SIMD run faster: tcn_hot_bwd4.hip)

// NW waves per workgroup (SG = false) or per SUBGROUP (SG = true: the workgroup is two independent 4-wave subgroups, one wave of each per
// SIMD, each with its own 64-row tiles, rounds, barriers and slab -- the two waves of a SIMD then sit in different phases of the round
// instead of both waiting at the same barrier); tile = 16*NW rows.  slab (floats): dW2 [64][128] | dW1 [128][CZP] | db2 [64] | db1 [128]
template <int NFZ, int NW, bool SG = false>
__global__ __launch_bounds__(SG ? 512 : 64 * NW) void dec_mse_bwd_kernel(const TT* __restrict__ Z, const frag8* __restrict__ Wpk, const float* __restrict__ b1,
                                                               const float* __restrict__ b2, const TT* __restrict__ TGT,
                                                               const uint8_t* __restrict__ mask, const float* __restrict__ gscale,
                                                               const float* __restrict__ stats, TT* __restrict__ DZ, int64_t P, int Cz,
                                                               float* __restrict__ slab) {
  constexpr int CZP = 32 * NFZ, CB = CZP / 16, R = 16 * NW, NTH = SG ? 512 : 64 * NW;
  constexpr int PX = DF_F + 8, PH = DF_H + 8, PZ = CZP + 8;            // LDS tile pitches (16-byte skew)
  static_assert(!SG || NW == 4, "subgroups are four waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* w1 = reinterpret_cast<frag8*>(smem);                 // [8][NFZ][64]     z -> hidden
  frag8* w2 = w1 + 8 * NFZ * 64;                              // [4][4][64]       hidden -> xhat
  frag8* w2t = w2 + 16 * 64;                                  // [8][2][64]       dxhat -> dhidden
  frag8* w1t = w2t + 16 * 64;                                 // [CB][4][64]      dhidden -> dz
  float* tb = reinterpret_cast<float*>(w1t + CB * 4 * 64);    // b1[128] | b2[64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int sg = SG ? (tid >> 8) : 0;                          // subgroup: waves 0-3 / 4-7
  const int wave = SG ? ((tid >> 6) & 3) : (tid >> 6);         // wave inside the workgroup / subgroup
  constexpr int TILE_ELEMS = R * (PX + 2 * PH + PZ);
  unsigned* bars = reinterpret_cast<unsigned*>(tb + 192);     // (SG) two arrival counters, 64 bytes apart
  TT* t_dx = reinterpret_cast<TT*>(tb + 192 + (SG ? 32 : 0)) + sg * TILE_ELEMS;   // [R][PX]
  TT* t_h = t_dx + R * PX;                                    // [R][PH]
  TT* t_dh = t_h + R * PH;                                    // [R][PH]
  TT* t_z = t_dh + R * PH;                                    // [R][PZ]
  const int px = lane & 15, kc = lane >> 4, r16 = px;
  const int prow = wave * 16 + px;
  copy_frags_lds<TT>(w1, Wpk, (8 * NFZ + 16 + 16 + CB * 4) * 64, tid, NTH);
  for (int i = tid; i < 192; i += NTH) tb[i] = i < 128 ? b1[i] : b2[i - 128];
  if (SG && tid < 2) bars[16 * tid] = 0u;
  unsigned* bar = bars + 16 * sg;
  unsigned bgen = 0;
  (void)bar; (void)bgen;
  __syncthreads();
  const float* b1q = tb + 32 * kc;
  const float* b2q = tb + 128 + 16 * kc;
  const bool fastz = (Cz == CZP);
  const float nv = stats[1];
  const float ksc = nv > 0.f ? (gscale ? gscale[0] : 1.f) * 2.f / nv : 0.f;

  // weight-gradient ownership
  constexpr int W2CB = (NW == 8) ? 4 : 8;                     // dW2 column blocks per wave
  const int w2row = (NW == 8) ? (wave & 3) : wave;            // dW2 row block (16 output features)
  const int w2col0 = (NW == 8) ? 4 * (wave >> 2) : 0;
  constexpr int W1RB = (NW == 8) ? 1 : 2;                     // dW1 row blocks per wave
  f32x4 a2[W2CB], a1[W1RB][CB], ab2, ab1[W1RB];
#pragma unroll
  for (int i = 0; i < W2CB; ++i) a2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < W1RB; ++r) {
    ab1[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CB; ++i) a1[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  ab2 = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16 one = (bf16)1.f, zero = (bf16)0.f;
  const bf16x8 ones = (r16 == 0) ? bf16x8{one, one, one, one, one, one, one, one} : bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};

  const int64_t nwt = (P + R - 1) / R;
  // rounds of this workgroup (SG: of this subgroup): first, step, end
  int64_t w_first = blockIdx.x, w_end = nwt;
  const int64_t w_step = gridDim.x;
  if (SG) {                                                    // the workgroup's rounds blockIdx.x + i gridDim.x, split 19 : 13 between the subgroups
    const int64_t n_wg = (int64_t)blockIdx.x < nwt ? (nwt - 1 - blockIdx.x) / w_step + 1 : 0;
    const int64_t n0 = (n_wg * DF_SG_SHARE0 + 16) >> 5;
    w_first = blockIdx.x + (sg ? n0 : 0) * w_step;
    w_end = blockIdx.x + (sg ? n_wg : n0) * w_step;
    if (w_end > nwt) w_end = nwt;
  }
  // The rows of the next THREE rounds are in flight while a round computes: the workgroup is alone on its CU (LDS) and a round's rows
  // are only ~19 KB, so one round of look-ahead left the kernel bound by memory latency (19 KB per ~3 us per CU = 1.3 TB/s over the chip).
  LQTile<TT, NFZ> zt, z1, z2, zn;
  LQTile<TT, 2> tt, t1, t2, tn;
  auto fetch = [&](LQTile<TT, NFZ>& zz, LQTile<TT, 2>& tg, int64_t w) {
    const int64_t wc = w < nwt ? w : (nwt - 1);
    const int64_t r0 = wc * R + prow;
    const int64_t rc = r0 < P ? r0 : P - 1;
    lq_load<TT, NFZ>(zz, Z, rc, Cz, kc, fastz);
    lq_load<TT, 2>(tg, TGT, rc, DF_F, kc, true);
  };
  fetch(zt, tt, w_first);
  fetch(z1, t1, w_first + w_step);
  fetch(z2, t2, w_first + 2 * w_step);
  for (int64_t wt = w_first; wt < w_end; wt += w_step) {
    int64_t row = wt * R + prow;
    const bool inb = row < P;
    bool valid = inb;
    if (!inb) row = P - 1;
    if (valid && mask != nullptr) valid = mask[row] != 0;
    fetch(zn, tn, wt + 3 * w_step);
    float h[32], xh[16];
    LQTile<TT, 4> ht;
    dec_chain<NFZ>(h, xh, zt, w1, w2, b1q, b2q, ht, lane);
    LQTile<TT, 2> dxt;
    const float kv = valid ? ksc : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dxt.f[j >> 3][j & 7] = (bf16)(kv * (xh[j] - (float)tt.f[j >> 3][j & 7]));
    f32x4 dhacc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      dhacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) dhacc[m] = mfma16(w2t[(m * 2 + s) * 64 + lane], dxt.f[s], dhacc[m]);
    }
    LQTile<TT, 4> dht;
#pragma unroll
    for (int j = 0; j < 32; ++j) dht.f[j >> 3][j & 7] = (bf16)(h[j] > 0.f ? dhacc[j >> 2][j & 3] : 0.f);
    f32x4 dzacc[CB];
#pragma unroll
    for (int m = 0; m < CB; ++m) {
      dzacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) dzacc[m] = mfma16(w1t[(m * 4 + s) * 64 + lane], dht.f[s], dzacc[m]);
    }
    if (inb) {
      constexpr int QZ = 4 * CB;                               // latent channels per lane quarter
      TT* dzp = DZ + row * (int64_t)Cz + QZ * kc;
      if (fastz) {
#pragma unroll
        for (int j = 0; j < QZ; j += 8) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)dzacc[(j + e) >> 2][(j + e) & 3];
          *reinterpret_cast<bf16x8*>(dzp + j) = o;
        }
      } else if ((Cz & 3) == 0) {                              // 12-channel latent rows: 8-byte stores
#pragma unroll
        for (int j = 0; j < QZ; j += 4)
          if (QZ * kc + j < Cz)
            *reinterpret_cast<bf16x4*>(dzp + j) = bf16x4{(bf16)dzacc[j >> 2][0], (bf16)dzacc[j >> 2][1], (bf16)dzacc[j >> 2][2], (bf16)dzacc[j >> 2][3]};
      } else {
#pragma unroll
        for (int j = 0; j < QZ; ++j)
          if (QZ * kc + j < Cz) dzp[j] = (bf16)dzacc[j >> 2][j & 3];
      }
    }
    if (!inb) {                                               // rows past the end contribute nothing to the weight gradients
#pragma unroll
      for (int s = 0; s < 4; ++s) dht.f[s] = bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};
    }
    tile_put<2>(t_dx, PX, prow, kc, dxt);
    tile_put<4>(t_h, PH, prow, kc, ht);
    tile_put<4>(t_dh, PH, prow, kc, dht);
    tile_put<NFZ>(t_z, PZ, prow, kc, zt);
    if constexpr (SG) df_sg_sync(bar, bgen, lane); else __syncthreads();
#pragma unroll
    for (int ks = 0; ks < R / 32; ++ks) {
      const int pix0 = ks * 32 + 8 * kc;
      const bf16x8 af = tr_frag_p(t_dx, PX, pix0, w2row * 16, r16);                 // dW2 rows: output features
#pragma unroll
      for (int i = 0; i < W2CB; ++i) a2[i] = mfma16(af, tr_frag_p(t_h, PH, pix0, (w2col0 + i) * 16, r16), a2[i]);
      if (NW == 4 || (wave >> 2) == 0) ab2 = mfma16(af, ones, ab2);
#pragma unroll
      for (int r = 0; r < W1RB; ++r) {
        const int rb = (NW == 8) ? wave : 2 * wave + r;                              // dW1 rows: hidden units
        const bf16x8 ah = tr_frag_p(t_dh, PH, pix0, rb * 16, r16);
#pragma unroll
        for (int i = 0; i < CB; ++i) a1[r][i] = mfma16(ah, tr_frag_p(t_z, PZ, pix0, i * 16, r16), a1[r][i]);
        ab1[r] = mfma16(ah, ones, ab1[r]);
      }
    }
    if constexpr (SG) df_sg_sync(bar, bgen, lane); else __syncthreads();
    zt = z1; z1 = z2; z2 = zn;
    tt = t1; t1 = t2; t2 = tn;
  }
  float* my = slab + (int64_t)(SG ? 2 * blockIdx.x + sg : blockIdx.x) * (DF_F * DF_H + DF_H * CZP + DF_F + DF_H);   // (SG: a slab per subgroup)
#pragma unroll
  for (int i = 0; i < W2CB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) my[(w2row * 16 + kc * 4 + r) * DF_H + (w2col0 + i) * 16 + r16] = a2[i][r];
#pragma unroll
  for (int rr = 0; rr < W1RB; ++rr) {
    const int rb = (NW == 8) ? wave : 2 * wave + rr;
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[DF_F * DF_H + (rb * 16 + kc * 4 + r) * CZP + i * 16 + r16] = a1[rr][i][r];
    if (r16 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) my[DF_F * DF_H + DF_H * CZP + DF_F + rb * 16 + kc * 4 + r] = ab1[rr][r];
    }
  }
  if (r16 == 0 && (NW == 4 || (wave >> 2) == 0)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) my[DF_F * DF_H + DF_H * CZP + w2row * 16 + kc * 4 + r] = ab2[r];
  }
}

template <int NFZ>
__global__ void dec_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ W1, const float* __restrict__ W2, int Cz, int bwd) {
  constexpr int CZP = 32 * NFZ, CB = CZP / 16;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  pack_weights_lds<TT, NFZ>(dst, W1, DF_H, Cz, 8, Cz, 1, tid, nt);                       // hidden[o] = sum_i W1[o][i] z[i]
  frag8* p = dst + 8 * NFZ * 64;
  pack_weights_lds<TT, 4>(p, W2, DF_F, DF_H, 4, DF_H, 1, tid, nt);                       // xhat[o] = sum_i W2[o][i] h[i]
  if (!bwd) return;
  p += 16 * 64;
  pack_weights_lds<TT, 2>(p, W2, DF_H, DF_F, 8, 1, DF_H, tid, nt);                       // dh[o=h] = sum_f W2[f][h] dx[f]
  p += 16 * 64;
  pack_weights_lds<TT, 4>(p, W1, Cz, DF_H, CB, 1, Cz, tid, nt);                          // dz[o=c] = sum_h W1[h][c] dh[h]
}

// Packed image of the decoder weights (layout of dec_pack_kernel): from the caller's image cache when one is active, else packed into `ws_pk`.
template <int NFZ>
static const frag8* dec_packed(const float* w1, const float* w2, int Cz, int bwd, frag8* ws_pk, hipStream_t st) {
  constexpr int CZP = 32 * NFZ, CB = CZP / 16;
  FrlPackJob jobs[4];
  size_t off = 0;
  jobs[0] = frl_pack_job_pw(w1, off, FRL_BF16, NFZ, DF_H, Cz, 8, Cz, 1);
  off += (size_t)8 * NFZ * 64 * sizeof(frag8);
  jobs[1] = frl_pack_job_pw(w2, off, FRL_BF16, 4, DF_F, DF_H, 4, DF_H, 1);
  off += (size_t)16 * 64 * sizeof(frag8);
  int n = 2;
  if (bwd) {
    jobs[2] = frl_pack_job_pw(w2, off, FRL_BF16, 2, DF_H, DF_F, 8, 1, DF_H);
    off += (size_t)16 * 64 * sizeof(frag8);
    jobs[3] = frl_pack_job_pw(w1, off, FRL_BF16, 4, Cz, DF_H, CB, 1, Cz);
    off += (size_t)CB * 4 * 64 * sizeof(frag8);
    n = 4;
  }
  bool hit = false;
  frag8* pk = ws_pk;
  if (void* img = frl_pack_cached(jobs, n, off, &hit)) pk = (frag8*)img;
  if (!hit) FRL_LAUNCH((dec_pack_kernel<NFZ>), dim3(32), dim3(256), 0, st, pk, w1, w2, Cz, bwd);
  return pk;
}


static unsigned df_fwd_grid(int64_t P) { int64_t g = ((P + 15) / 16 + 3) / 4; if (g > DF_GRID_MAX) g = DF_GRID_MAX; return (unsigned)(g < 1 ? 1 : g); }
static unsigned df_bwd_grid(int64_t P, int R) { int64_t g = (P + R - 1) / R; if (g > 256) g = 256; return (unsigned)(g < 1 ? 1 : g); }

template <int NFZ, int NW>
static int launch_dec_bwd(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* tgt,
                          const uint8_t* mask, const float* gscale, const float* stats, void* dz, float* dw1, float* db1, float* dw2,
                          float* db2, int64_t P, int Cz, char* ws, hipStream_t st) {
  constexpr int CZP = 32 * NFZ, CB = CZP / 16, R = 16 * NW;
  const unsigned grid = df_bwd_grid(P, R);
  const size_t slab_n = (size_t)DF_F * DF_H + DF_H * CZP + DF_F + DF_H;
  const frag8* pk = dec_packed<NFZ>(w1, w2, Cz, 1, reinterpret_cast<frag8*>(ws + ((grid * slab_n * sizeof(float) + 255) / 256) * 256), st);
  const size_t lds = (size_t)(8 * NFZ + 16 + 16 + CB * 4) * 64 * sizeof(frag8) + 192 * sizeof(float) +
                     (size_t)R * ((DF_F + 8) + 2 * (DF_H + 8) + (CZP + 8)) * sizeof(TT);
  auto kern = dec_mse_bwd_kernel<NFZ, NW>;
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH_AS("dec_mse_bwd_kernel", kern, dim3(grid), dim3(64 * NW), lds, st, (const TT*)z, (const frag8*)pk, b1, b2, (const TT*)tgt, mask, gscale, stats, (TT*)dz, P, Cz,
             (float*)ws);
  launch_slab_reduce_deferrable<float, DecEpi>((const float*)ws, (int)grid, (int64_t)slab_n, DecEpi{dw2, dw1, db2, db1, Cz, CZP, DF_F, DF_H}, st);
  return frl_check_launch("decoder_mse_bwd");
}

// two independent 4-wave subgroups per workgroup (64-row tiles each), a slab per subgroup
template <int NFZ>
static int launch_dec_bwd_sg(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* tgt,
                             const uint8_t* mask, const float* gscale, const float* stats, void* dz, float* dw1, float* db1, float* dw2,
                             float* db2, int64_t P, int Cz, char* ws, hipStream_t st) {
  constexpr int CZP = 32 * NFZ, CB = CZP / 16, R = 64;
  int64_t g = (P + 2 * R - 1) / (2 * R);
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  const unsigned grid = (unsigned)g, nslab = 2 * grid;
  const size_t slab_n = (size_t)DF_F * DF_H + DF_H * CZP + DF_F + DF_H;
  const frag8* pk = dec_packed<NFZ>(w1, w2, Cz, 1, reinterpret_cast<frag8*>(ws + ((nslab * slab_n * sizeof(float) + 255) / 256) * 256), st);
  const size_t lds = (size_t)(8 * NFZ + 16 + 16 + CB * 4) * 64 * sizeof(frag8) + (192 + 32) * sizeof(float) +
                     (size_t)2 * R * ((DF_F + 8) + 2 * (DF_H + 8) + (CZP + 8)) * sizeof(TT);
  auto kern = dec_mse_bwd_kernel<NFZ, 4, true>;
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH_AS("dec_mse_bwd_sg_kernel", kern, dim3(grid), dim3(512), lds, st, (const TT*)z, (const frag8*)pk, b1, b2, (const TT*)tgt, mask, gscale, stats, (TT*)dz, P,
                Cz, (float*)ws);
  launch_slab_reduce_deferrable<float, DecEpi>((const float*)ws, (int)nslab, (int64_t)slab_n, DecEpi{dw2, dw1, db2, db1, Cz, CZP, DF_F, DF_H}, st);
  return frl_check_launch("decoder_mse_bwd");
}

static int g_dec_subgroups = 1;     // A/B hook (frl_decoder_mse_bwd_subgroups): 0 = the lockstep 8-wave / 4-wave workgroups of round 2

extern "C" {

int frl_decoder_mse_bwd_subgroups(int on) { const int was = g_dec_subgroups; g_dec_subgroups = on ? 1 : 0; return was; }

int frl_decoder_mse_fused_supported(int Cz, int hidden, int F, int dtype) {
  return (dtype == FRL_BF16 && hidden == DF_H && F == DF_F && Cz >= 1 && Cz <= 64) ? 1 : 0;
}

size_t frl_decoder_mse_workspace_bytes(int64_t P, int Cz) {
  const size_t czp = Cz <= 32 ? 32 : 64;
  const size_t slab_n = (size_t)DF_F * DF_H + DF_H * czp + DF_F + DF_H;
  const size_t bwd = 512 * slab_n * sizeof(float) + 256 + (size_t)(8 * 2 + 32 + 16) * 64 * 16;     // (a slab per subgroup: two per workgroup)
  const size_t fwd = (size_t)DF_GRID_MAX * 2 * sizeof(double) + 256 + (size_t)(8 * 2 + 16) * 64 * 16;
  (void)P;
  return bwd > fwd ? bwd : fwd;
}

// z [P][Cz], target [P][64] bf16; w1 [128][Cz], w2 [64][128] f32.  out = {mean squared error over valid elements, n_valid};
// xhat (optional, [P][64]) receives the reconstruction.
int frl_decoder_mse_fwd(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* target,
                        const uint8_t* mask, void* xhat, float* out, int64_t P, int Cz, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (P <= 0) return frl_fail(-2, "decoder_mse_fwd: empty input");
  if (Cz < 1 || Cz > 64) return frl_fail(-2, "decoder_mse_fwd: latent width must be 1..64");
  if (ws_bytes < frl_decoder_mse_workspace_bytes(P, Cz)) return frl_fail(-4, "decoder_mse_fwd: workspace too small");
  char* w = (char*)ws;
  double* partial = (double*)w;
  frag8* ws_pk = reinterpret_cast<frag8*>(w + (size_t)DF_GRID_MAX * 2 * sizeof(double) + 256);
  const unsigned grid = df_fwd_grid(P);
  if (Cz <= 32) {
    const frag8* pk = dec_packed<1>(w1, w2, Cz, 0, ws_pk, stream);
    const size_t lds = (size_t)(8 + 16) * 64 * sizeof(frag8) + 192 * sizeof(float);
    FRL_LAUNCH((dec_mse_fwd_kernel<1>), dim3(grid), dim3(256), lds, stream, (const TT*)z, (const frag8*)pk, b1, b2, (const TT*)target, mask,
               (TT*)xhat, P, Cz, partial);
  } else {
    const frag8* pk = dec_packed<2>(w1, w2, Cz, 0, ws_pk, stream);
    const size_t lds = (size_t)(16 + 16) * 64 * sizeof(frag8) + 192 * sizeof(float);
    FRL_LAUNCH((dec_mse_fwd_kernel<2>), dim3(grid), dim3(256), lds, stream, (const TT*)z, (const frag8*)pk, b1, b2, (const TT*)target, mask,
               (TT*)xhat, P, Cz, partial);
  }
  FRL_LAUNCH(dec_mse_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double*)partial, (int)grid, out);
  return frl_check_launch("decoder_mse_fwd");
}

// gradients: dz [P][Cz] bf16; dw1 [128][Cz], db1 [128], dw2 [64][128], db2 [64] f32.  gscale: device scalar (upstream grad of
// the loss) or null; stats = frl_decoder_mse_fwd's out.
int frl_decoder_mse_bwd(const void* z, const float* w1, const float* b1, const float* w2, const float* b2, const void* target,
                        const uint8_t* mask, const float* gscale, const float* stats, void* dz, float* dw1, float* db1, float* dw2,
                        float* db2, int64_t P, int Cz, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (P <= 0) return frl_fail(-2, "decoder_mse_bwd: empty input");
  if (Cz < 1 || Cz > 64) return frl_fail(-2, "decoder_mse_bwd: latent width must be 1..64");
  if (ws_bytes < frl_decoder_mse_workspace_bytes(P, Cz)) return frl_fail(-4, "decoder_mse_bwd: workspace too small");
  // (latents of more than 32 channels: the two subgroups' tiles do not fit the LDS beside the four weight images: 172 KB)
  if (g_dec_subgroups && P >= 128 && Cz <= 32)
    return launch_dec_bwd_sg<1>(z, w1, b1, w2, b2, target, mask, gscale, stats, dz, dw1, db1, dw2, db2, P, Cz, (char*)ws, stream);
  if (Cz <= 32) return launch_dec_bwd<1, 8>(z, w1, b1, w2, b2, target, mask, gscale, stats, dz, dw1, db1, dw2, db2, P, Cz, (char*)ws, stream);
  return launch_dec_bwd<2, 4>(z, w1, b1, w2, b2, target, mask, gscale, stats, dz, dw1, db1, dw2, db2, P, Cz, (char*)ws, stream);
}

}  // extern "C"
