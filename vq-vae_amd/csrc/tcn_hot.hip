// Hot-configuration GatedResidualBlock kernels (frl/models/tcn.py:78-111): bf16, Cin = Cout = 64, T = 5, 8-channel GroupNorm
// groups, identity residual, dilation 1 / 2 / 4 -- the three blocks of the phase path at BASELINE configs[1..3].
//
// Everything that depends on (T, dilation) is a template constant, so the whole per-pixel time series lives in registers
// with static indexing (no scratch, no select chains): a wave owns 16 pixels for all 5 time steps in the lane-quarter image
// (frl_common.hpp).  The temporal convolution is computed ONCE per tile into 20 accumulator tiles (5 t x 4 channel blocks),
// each weight fragment is fetched from LDS once and reused for every time step it applies to, the GroupNorm statistics are an
// exact two-pass in-lane reduction over those accumulators, and the gate GEMM / sigmoid / blend read them in place.
//
//   forward : y = g * relu(n) + (1 - g) * x,  n = GN(conv(x) + b),  g = sigmoid(Wg n + bg)          (x read, y written: 2 x 128 B / (px,t))
//   backward: ONE launch -> dx + every parameter gradient (tcn_hot_bwd2_kernel, 8 waves per workgroup with a channel-half split;
//             the phase structure is described above the kernel).  Wave (q, h) owns rows [16q, 16q+16) x columns [32h, 32h+32) of the
//             four 64x64 gradient matrices in registers for the whole kernel; per-workgroup float32 slabs are summed in a fixed order
//             afterwards (bit-reproducible, no float atomics).
#include "tcn_hot_common.hpp"

// Diagnostic build only (tools/diag/tcn_bwd_stamps.hip defines TH_STAMPS): s_memtime stamps at the phase boundaries of the
// backward kernel, accumulated in LDS and written to a buffer nothing else reads.  The product library never defines it.
#ifdef TH_STAMPS
__device__ unsigned long long* th_dbg;
#endif
#define TH_PITCH 72       // bf16 elements per pixel row in LDS tiles (64 + 8: conflict-free 16-byte writes and tr16 reads)

struct Tile2 { frag8 f[2]; };

__device__ __forceinline__ float th_elem(const Tile2& t, int j) { return (float)t.f[j >> 3][j & 7]; }

__device__ __forceinline__ Tile2 th_pack(const float (&v)[16]) {
  Tile2 o;
#pragma unroll
  for (int j = 0; j < 16; ++j) o.f[j >> 3][j & 7] = (bf16)v[j];
  return o;
}

// conv for all time steps: acc[t][m] (+)= sum_k W_k x[t + (k-1) DIL]; each weight fragment is read from LDS once
template <int DIL>
__device__ __forceinline__ void th_conv(f32x4 (&acc)[TH_T][4], const Tile2 (&x)[TH_T], const frag8* __restrict__ wl, int lane) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const frag8 wf = wl[((k * 4 + m) * 2 + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < TH_T; ++t)
          if (th_valid<DIL>(t, k)) acc[t][m] = mfma16(wf, x[t + (k - 1) * DIL].f[s], acc[t][m]);
      }
}

// exact two-pass GroupNorm statistics of the lane's two 8-channel groups over (8 ch x 5 t); acc already holds conv + bias.
// Four independent partial sums per reduction (one per accumulator register): the 40-term chains become 10-term chains, which matters
// at two waves per SIMD where nothing else hides the dependent-add latency.
__device__ __forceinline__ void th_stats(const f32x4 (&acc)[TH_T][4], float eps, float (&mean)[2], float (&rstd)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TH_T; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += acc[t][2 * g + m][r];
    const float mu = ((s[0] + s[1]) + (s[2] + s[3])) * (1.f / 40.f);
    f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TH_T; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc[t][2 * g + m][r] - mu; q[r] = fmaf(d, d, q[r]); }
    mean[g] = mu;
    rstd[g] = 1.f / sqrtf(((q[0] + q[1]) + (q[2] + q[3])) * (1.f / 40.f) + eps);
  }
}

// element-wise products of bf16 images (Dropout1d mask applied to the conv input / its gradient)
__device__ __forceinline__ frag8 th_mul8(const frag8& a, const frag8& b) {
  frag8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * (float)b[j]);
  return o;
}
__device__ __forceinline__ Tile2 th_mul(const Tile2& a, const Tile2& b) {
  Tile2 o;
  o.f[0] = th_mul8(a.f[0], b.f[0]);
  o.f[1] = th_mul8(a.f[1], b.f[1]);
  return o;
}

__device__ __forceinline__ Tile2 th_load(const bf16* __restrict__ p) {
  Tile2 t;
  t.f[0] = *reinterpret_cast<const frag8*>(p);
  t.f[1] = *reinterpret_cast<const frag8*>(p + 8);
  return t;
}
__device__ __forceinline__ void th_store(bf16* __restrict__ p, const Tile2& t) {
  *reinterpret_cast<frag8*>(p) = t.f[0];
  *reinterpret_cast<frag8*>(p + 8) = t.f[1];
}

// =============================================================================================================
// forward
// =============================================================================================================
// MASK: training-mode Dropout1d of the block (tcn.py:53): the conv sees x .* M, the residual path the untouched x; M [B*HW][64] holds
// 0 or 1/(1-p) per (pixel series, channel).  The MASK = false instantiation is the kernel without any trace of it.
template <int DIL, bool MASK>
__global__ __launch_bounds__(256, 2) void tcn_hot_fwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ M, const frag8* __restrict__ Wpk,
                                                             const float* __restrict__ bc, const float* __restrict__ gn_w,
                                                             const float* __restrict__ gn_b, const float* __restrict__ bg,
                                                             bf16* __restrict__ Y, int64_t npix, int HW, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wl_conv = reinterpret_cast<frag8*>(smem);                 // [3][4][2][64]
  frag8* wl_gate = wl_conv + 24 * 64;                              // [4][2][64]
  float* tab = reinterpret_cast<float*>(wl_gate + 8 * 64);         // conv bias | gamma | beta | -log2e * gate bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<bf16>(wl_conv, Wpk, 32 * 64, tid, 256);
  if (tid < 64) {
    tab[tid] = bc[tid];
    tab[64 + tid] = gn_w[tid];
    tab[128 + tid] = gn_b[tid];
    tab[192 + tid] = -1.44269504088896f * bg[tid];
  }
  __syncthreads();
  const f32x4* tcb = reinterpret_cast<const f32x4*>(tab + 16 * kc);
  const float* tgw = tab + 64 + 16 * kc;
  const float* tgb = tab + 128 + 16 * kc;
  const float* tnbg = tab + 192 + 16 * kc;

  const int64_t ntile = (npix + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    int64_t pidx = tile * 16 + px;
    const bool valid = pidx < npix;
    if (!valid) pidx = npix - 1;
    const int64_t b = pidx / HW, hw = pidx % HW;
    const int64_t row0 = b * TH_T * HW + hw;
    const bf16* xp = X + row0 * 64 + 16 * kc;
    Tile2 x[TH_T];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) x[t] = th_load(xp + (int64_t)t * HW * 64);
    f32x4 acc[TH_T][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 cb = tcb[m];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) acc[t][m] = cb;
    }
    if constexpr (MASK) {
      const Tile2 mk = th_load(M + pidx * 64 + 16 * kc);
      Tile2 xc[TH_T];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) xc[t] = th_mul(x[t], mk);
      th_conv<DIL>(acc, xc, wl_conv, lane);
    } else {
      th_conv<DIL>(acc, x, wl_conv, lane);
    }
    float mean[2], rstd[2];
    th_stats(acc, eps, mean, rstd);
    const float nm[2] = {-mean[0] * rstd[0], -mean[1] * rstd[1]};  // xhat = acc * rstd + nm
    bf16* yp = Y + row0 * 64 + 16 * kc;
#pragma unroll
    for (int t = 0; t < TH_T; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      float n[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) n[j] = fmaf(fmaf(acc[t][j >> 2][j & 3], rstd[j >> 3], nm[j >> 3]), tgw[j], tgb[j]);
      const Tile2 nt = th_pack(n);
      f32x4 gacc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) gacc[m] = mfma16(wl_gate[(m * 2 + s) * 64 + lane], nt.f[s], gacc[m]);
      }
      float y[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float g = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(gacc[j >> 2][j & 3], -1.44269504088896f, tnbg[j])));
        const float o = fmaxf(n[j], 0.f);
        const float res = th_elem(x[t], j);
        y[j] = fmaf(g, o - res, res);
      }
      if (valid) th_store(yp + (int64_t)t * HW * 64, th_pack(y));
    }
  }
}

// =============================================================================================================
// forward, the whole dense phase chain in ONE launch: three GatedResidualBlocks (dilation 1, 2, 4; tcn.py:242-290) followed by the
// 1x1 phase head (representation.py:169,362-366).  A wave carries its 16 pixel series through the three blocks in registers: the
// block's output IS the next block's lane-quarter input image (rounded to bf16 exactly where the stand-alone kernels round it on their
// store), so the tile is read once; y1, y2 (the inputs the block backward passes recompute from), y3 (the head's weight-gradient operand)
// and the 12-channel head output are written once each.  Against three block launches + the head's 1x1 convolution this drops the
// re-reads of y1, y2, y3 (3 x 168 MB at cfg2) and three launches.  8 waves per workgroup (two per SIMD), all three weight images
// resident in LDS (3 x 32 KB), the next tile's rows prefetched into registers behind the current tile's three blocks.
// =============================================================================================================
template <int DIL>
__device__ __forceinline__ void th_block_regs(Tile2 (&x)[TH_T], const frag8* __restrict__ wl_conv, const frag8* __restrict__ wl_gate,
                                              const float* __restrict__ tab, int lane, int kc, float eps) {
  const f32x4* tcb = reinterpret_cast<const f32x4*>(tab + 16 * kc);
  const float* tgw0 = tab + 64 + 16 * kc;
  const float* tgb0 = tab + 128 + 16 * kc;
  const float* tnbg0 = tab + 192 + 16 * kc;
  f32x4 acc[TH_T][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 cb = tcb[m];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) acc[t][m] = cb;
  }
  th_conv<DIL>(acc, x, wl_conv, lane);
  float mean[2], rstd[2];
  th_stats(acc, eps, mean, rstd);
  const float nm[2] = {-mean[0] * rstd[0], -mean[1] * rstd[1]};
#pragma unroll
  for (int t = 0; t < TH_T; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    int zt = 0;
    asm volatile("" : "+s"(zt));                                    // per-channel constants re-read from LDS per time step (48 registers less)
    const float* tgw = tgw0 + zt;
    const float* tgb = tgb0 + zt;
    const float* tnbg = tnbg0 + zt;
    float n[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) n[j] = fmaf(fmaf(acc[t][j >> 2][j & 3], rstd[j >> 3], nm[j >> 3]), tgw[j], tgb[j]);
    const Tile2 nt = th_pack(n);
    f32x4 gacc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) gacc[m] = mfma16(wl_gate[(m * 2 + s) * 64 + lane], nt.f[s], gacc[m]);
    }
    float y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float g = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(gacc[j >> 2][j & 3], -1.44269504088896f, tnbg[j])));
      const float o = fmaxf(n[j], 0.f);
      const float res = th_elem(x[t], j);
      y[j] = fmaf(g, o - res, res);
    }
    x[t] = th_pack(y);                                              // (x[t] was last needed as this step's residual)
  }
}

struct ThChainArgs {
  const frag8* pk[3];                                              // packed images of the three blocks (layout of tcn_hot_pack_kernel)
  const float* bc[3]; const float* gw[3]; const float* gb[3]; const float* bg[3];
  const frag8* pkh;                                                // head: [2 k-steps][64] fragments (one 16-row block)
  const float* bh;                                                 // head bias [Ch]
  bf16* y[3];
  bf16* h;
};

#define THC_IMG (32 * 64)                                          // fragments of one block's forward image (conv taps + gate)
__global__ __launch_bounds__(512, 2) void tcn_chain_fwd_kernel(const bf16* __restrict__ X, ThChainArgs a, int64_t npix, int HW, int Ch, float eps, int dyn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wl = reinterpret_cast<frag8*>(smem);                      // [3][THC_IMG]
  frag8* wlh = wl + 3 * THC_IMG;                                   // [2][64]
  float* tab = reinterpret_cast<float*>(wlh + 2 * 64);             // [3][256]: conv bias | gamma | beta | -log2e * gate bias;  then head bias [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
#pragma unroll
  for (int b = 0; b < 3; ++b) copy_frags_lds<bf16>(wl + b * THC_IMG, a.pk[b], THC_IMG, tid, 512);
  copy_frags_lds<bf16>(wlh, a.pkh, 2 * 64, tid, 512);
  for (int i = tid; i < 3 * 256 + 16; i += 512) {
    float v;
    if (i < 768) {
      const int b = i >> 8, w = (i >> 6) & 3, c = i & 63;
      v = w == 0 ? a.bc[b][c] : w == 1 ? a.gw[b][c] : w == 2 ? a.gb[b][c] : -1.44269504088896f * a.bg[b][c];
    } else {
      v = (i - 768) < Ch ? a.bh[i - 768] : 0.f;
    }
    tab[i] = v;
  }
  // The workgroup's tiles (blockIdx.x * 8 + (i & 7) + (i >> 3) * 8 * gridDim.x, i = 0, 1, ...) are handed to its waves by a counter in LDS:
  // when both waves of a SIMD are ready the sequencer issues from the older one, so waves 4-7 run a tile ~40 % slower than waves 0-3
  // while both are busy (measured in tcn_hot_bwd4: profiles/r03_tcn_bwd_stamps.md) and a static split leaves the older waves idle at the
  // end.  The forward accumulates nothing across tiles, so which wave computes a tile does not change a bit of the result.
  int* tctr = reinterpret_cast<int*>(tab + 3 * 256 + 16);
  if (tid == 0) *tctr = 8;
  __syncthreads();
  const int64_t ntile = (npix + 15) >> 4;
  const int64_t tstep = (int64_t)gridDim.x * 8;
  auto tile_of = [&](int i) -> int64_t { return (int64_t)blockIdx.x * 8 + (i & 7) + (int64_t)(i >> 3) * tstep; };
  auto rows = [&](int64_t tile, bool& valid) -> int64_t {          // element offset of (t = 0, this lane's pixel, its channel quarter)
    int64_t pidx = tile * 16 + px;
    valid = pidx < npix;
    if (!valid) pidx = npix - 1;
    const int64_t b = pidx / HW, hw = pidx % HW;
    return (b * TH_T * HW + hw) * 64 + 16 * kc;
  };
  Tile2 x[TH_T];
  {
    bool v;
    const int64_t t0 = tile_of(wave);
    const int64_t r0 = rows(t0 < ntile ? t0 : 0, v);
#pragma unroll
    for (int t = 0; t < TH_T; ++t) x[t] = th_load(X + r0 + (int64_t)t * HW * 64);
  }
  int icur = wave;
  for (int64_t tile = tile_of(wave); tile < ntile;) {
    bool valid, vn;
    int inext = icur + 8;                                            // (static order: A/B hook frl_tcn_chain_static_tiles)
    if (dyn && lane == 0) inext = __hip_atomic_fetch_add(tctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (consumed at the end of the tile)
    const int64_t r0 = rows(tile, valid);
    int lw = lane, z0 = 0;
    asm volatile("" : "+v"(lw), "+s"(z0));                        // opaque per tile: neither the weight fragments nor the 3 x 48 per-channel
    const float* tb_ = tab + z0;                                  // constants are hoisted out of the tile loop into registers
    th_block_regs<1>(x, wl, wl + 24 * 64, tb_, lw, kc, eps);
    if (valid) {
#pragma unroll
      for (int t = 0; t < TH_T; ++t) th_store(a.y[0] + r0 + (int64_t)t * HW * 64, x[t]);
    }
    __builtin_amdgcn_sched_barrier(0);                             // (one block at a time: interleaved, the three bodies do not fit the register file)
    th_block_regs<2>(x, wl + THC_IMG, wl + THC_IMG + 24 * 64, tb_ + 256, lw, kc, eps);
    if (valid) {
#pragma unroll
      for (int t = 0; t < TH_T; ++t) th_store(a.y[1] + r0 + (int64_t)t * HW * 64, x[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    th_block_regs<4>(x, wl + 2 * THC_IMG, wl + 2 * THC_IMG + 24 * 64, tb_ + 512, lw, kc, eps);
    __builtin_amdgcn_sched_barrier(0);
    // phase head: h[t] = W_h y3[t] + b_h, Ch <= 16 output channels: lane quarter kc holds channels 4 kc .. 4 kc + 3
    const f32x4 hb4 = *reinterpret_cast<const f32x4*>(tb_ + 768 + 4 * kc);
    const int64_t hrow = (r0 - 16 * kc) / 64 * Ch + 4 * kc;
#pragma unroll
    for (int t = 0; t < TH_T; ++t) {
      if (valid) th_store(a.y[2] + r0 + (int64_t)t * HW * 64, x[t]);
      f32x4 ha = mfma16(wlh[lw], x[t].f[0], hb4);
      ha = mfma16(wlh[64 + lw], x[t].f[1], ha);
      if (valid && 4 * kc < Ch)
        *reinterpret_cast<bf16x4*>(a.h + hrow + (int64_t)t * HW * Ch) = bf16x4{(bf16)ha[0], (bf16)ha[1], (bf16)ha[2], (bf16)ha[3]};
    }
    {                                                               // next tile's rows
      icur = __builtin_amdgcn_readfirstlane(inext);
      const int64_t tnext = tile_of(icur);
      const int64_t rn = rows(tnext < ntile ? tnext : tile, vn);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) x[t] = th_load(X + rn + (int64_t)t * HW * 64);
      tile = tnext;
    }
  }
}

// =============================================================================================================
// backward
// =============================================================================================================
// k-strided MFMA fragment (8 consecutive pixels of one channel) from a [pixel][TH_PITCH] LDS tile
__device__ __forceinline__ frag8 th_tr(const bf16* tile, int pix0, int ch0, int r16) {
  const bf16* a0 = tile + (pix0 + (r16 >> 2)) * TH_PITCH + ch0 + 4 * (r16 & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * TH_PITCH));
  return frag8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void th_put(bf16* tile, int prow, int kc, const Tile2& t) {
  frag8* p = reinterpret_cast<frag8*>(tile + prow * TH_PITCH + 16 * kc);
  p[0] = t.f[0];
  p[1] = t.f[1];
}

#define TH_TILE (64 * TH_PITCH)

// =============================================================================================================
// backward, 8 waves per workgroup (two per SIMD): wave (q, h) owns the 16 pixels of quarter q and the channel half h of every
// lane quarter (fragment h of the lane-quarter image = one 8-channel GroupNorm group per lane), i.e. half the per-lane state of
// the 4-wave kernel above, so the compiler keeps everything in registers and two waves per SIMD hide each other's LDS / MFMA
// latencies.  The GEMMs that contract over all 64 channels (gate, gate^T, conv^T) take the partner wave's half from the very
// LDS tiles that are published for the weight-gradient GEMMs anyway:
//   S1  conv -> GroupNorm statistics -> n[t]                                  publish n[t]        | barrier A
//   S2  gate GEMM, sigmoid, dgpre[t], dres[t] (parked in dx), relu path of dn  publish dgpre[t]    | barrier B
//   S3  gate^T GEMM -> dn -> d gamma, d beta, GroupNorm backward -> dconv[t];  P2 gate weight gradient (rows 16q.., column half h)
//                                                                                                  | barrier C
//   P3  publish dconv[t], x[t]                                                                     | barrier D
//   dx = conv^T(dconv) + dres (stored);  P4 conv weight gradients                                  | barrier E
// =============================================================================================================
__device__ __forceinline__ Tile2 th_get(const bf16* tile, int prow, int kc) {
  const frag8* p = reinterpret_cast<const frag8*>(tile + prow * TH_PITCH + 16 * kc);
  Tile2 t;
  t.f[0] = p[0];
  t.f[1] = p[1];
  return t;
}

template <int DIL, bool MASK>
__global__ __launch_bounds__(512, 2) void tcn_hot_bwd2_kernel(const bf16* __restrict__ X, const bf16* __restrict__ M, const bf16* __restrict__ DY,
                                                              const frag8* __restrict__ Wpk, const float* __restrict__ bc,
                                                              const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                              const float* __restrict__ bg, bf16* DX, float* __restrict__ slab,
                                                              int64_t npix, int HW, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wl_conv = reinterpret_cast<frag8*>(smem);               // [3][4][2][64]
  frag8* wl_gate = wl_conv + 24 * 64;                            // [4][2][64]
  frag8* wl_gateT = wl_gate + 8 * 64;                            // [4][2][64]
  frag8* wl_convT = wl_gateT + 8 * 64;                           // [3][4][2][64]
  float* tab = reinterpret_cast<float*>(wl_convT + 24 * 64);     // conv bias | gamma | beta | -log2e * gate bias
  float* gacc_lds = tab + 4 * 64;                                // [8 waves][2][32]
  bf16* bufA = reinterpret_cast<bf16*>(gacc_lds + 4 * 2 * 64);   // [T][64 px][PITCH]  dgpre[t], later dconv[t]
  bf16* bufB = bufA + TH_T * TH_TILE;                            // [T][64 px][PITCH]  n[t],     later x[t]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = wave & 3, h = wave >> 2;
  const int px = lane & 15, kc = lane >> 4, r16 = px;
  const int prow = q * 16 + px;
  const int co = 16 * kc + 8 * h;                                // first of this lane's 8 channels
#ifdef TH_STAMPS
  __shared__ unsigned long long th_ts[8][12];
  if (tid < 96) (&th_ts[0][0])[tid] = 0ull;
  unsigned long long t_prev = 0;
#define TH_ST2(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) th_ts[wave][i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define TH_ST2(i) do { } while (0)
#endif

  copy_frags_lds<bf16>(wl_conv, Wpk, 64 * 64, tid, 512);
  if (tid < 64) {
    tab[tid] = bc[tid];
    tab[64 + tid] = gn_w[tid];
    tab[128 + tid] = gn_b[tid];
    tab[192 + tid] = -1.44269504088896f * bg[tid];
  }
  __syncthreads();
  float gw[8], gb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gw[e] = tab[64 + co + e]; gb[e] = tab[128 + co + e]; }
  const float* tnbg = tab + 192 + co;
  const f32x4* tcb = reinterpret_cast<const f32x4*>(tab + co);
  const frag8* wc_own = wl_conv + (2 * h) * 2 * 64 + lane;       // + ((k*4 + mm)*2 + s)*64
  const frag8* wg_own = wl_gate + (2 * h) * 2 * 64 + lane;       // + (mm*2 + s)*64
  const frag8* wgT_own = wl_gateT + (2 * h) * 2 * 64 + lane;
  const frag8* wcT_own = wl_convT + (2 * h) * 2 * 64 + lane;

  f32x4 accC[3][2], accG[2], accCb = {0.f, 0.f, 0.f, 0.f}, accGb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) accC[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) accG[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dgam[8], dbet[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
  const frag8 ones = (r16 == 0) ? frag8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f}
                                : frag8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};

  const int nwt = (int)((npix + 63) >> 6);
  const int64_t tstride = (int64_t)HW * 64;
  // this lane's pixel of workgroup tile `wt`: row offset (elements) of its time step 0, or of the last pixel when out of range
  unsigned pix_cur = 0;                                          // clamped pixel index of the current tile (mask row)
  auto pix_off = [&](int wt, bool& ok) -> int64_t {
    unsigned pidx = (unsigned)wt * 64u + (unsigned)prow;
    ok = pidx < (unsigned)npix;
    if (!ok) pidx = (unsigned)npix - 1u;
    pix_cur = pidx;
    const unsigned b = pidx / (unsigned)HW, hw = pidx - b * (unsigned)HW;
    return ((int64_t)b * TH_T * HW + hw) * 64;
  };
#ifdef TH_STAMPS
  t_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int wt = blockIdx.x; wt < nwt; wt += gridDim.x) {
    bool cur_valid;
    const int64_t off = pix_off(wt, cur_valid);
    bf16* dxp = DX + off + co;
    Tile2 x[TH_T];
    frag8 dyo[TH_T];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) x[t] = th_load(X + off + 16 * kc + t * tstride);
    frag8 xo[TH_T], m_own = frag8{};
    f32x4 xh[TH_T][2];
    float mean, rstd;
    {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const f32x4 cb = tcb[mm];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) xh[t][mm] = cb;
      }
#pragma unroll
      for (int t = 0; t < TH_T; ++t) xo[t] = h ? x[t].f[1] : x[t].f[0];     // residual path: the untouched x
      if constexpr (MASK) {                                                  // conv path: x .* M
        const Tile2 mk = th_load(M + (int64_t)pix_cur * 64 + 16 * kc);
        m_own = h ? mk.f[1] : mk.f[0];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) x[t] = th_mul(x[t], mk);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const frag8 wf = wc_own[((k * 4 + mm) * 2 + s) * 64];
#pragma unroll
            for (int t = 0; t < TH_T; ++t)
              if (th_valid<DIL>(t, k)) xh[t][mm] = mfma16(wf, x[t + (k - 1) * DIL].f[s], xh[t][mm]);
          }
    }
    // dy is requested only now (its registers would otherwise sit next to x, the conv accumulators and the hoisted weight
    // fragments): the statistics / n[t] pass below covers most of the latency
#pragma unroll
    for (int t = 0; t < TH_T; ++t) dyo[t] = *reinterpret_cast<const frag8*>(DY + off + co + t * tstride);
    if (!cur_valid) {                                              // clamped duplicate pixel: contributes nothing
#pragma unroll
      for (int t = 0; t < TH_T; ++t) dyo[t] = frag8{};
    }
    {                                                              // exact two-pass statistics of this lane's group (8 ch x 5 t)
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xh[t][e >> 2][e & 3];
      mean = s * (1.f / 40.f);
      float qq = 0.f;
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xh[t][e >> 2][e & 3] - mean; qq = fmaf(d, d, qq); }
      rstd = 1.f / sqrtf(qq * (1.f / 40.f) + eps);
    }
    // ---------------- S1: xhat, n[t] -> LDS ----------------
    {
      const float nm = -mean * rstd;
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float n[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaf(xh[t][e >> 2][e & 3], rstd, nm);
          xh[t][e >> 2][e & 3] = v;
          n[e] = fmaf(v, gw[e], gb[e]);
        }
        *reinterpret_cast<frag8*>(bufB + t * TH_TILE + prow * TH_PITCH + co) = th_pack8(n);
      }
    }
    TH_ST2(0);
    __syncthreads();                                               // A: n[t] complete (all channels of every pixel)
    TH_ST2(1);
    // ---------------- S2: gate, dgpre, dres, relu path of dn ----------------
    frag8 dn0[TH_T];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) {
      const Tile2 nt = th_get(bufB + t * TH_TILE, prow, kc);
      f32x4 gacc[2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        gacc[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) gacc[mm] = mfma16(wg_own[(mm * 2 + s) * 64], nt.f[s], gacc[mm]);
      }
      float dgp[8], drv[8], dnr[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float n = fmaf(xh[t][e >> 2][e & 3], gw[e], gb[e]);
        const float dyv = (float)dyo[t][e];
        const float g = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(gacc[e >> 2][e & 3], -1.44269504088896f, tnbg[e])));
        const float o = fmaxf(n, 0.f);
        const float res = (float)xo[t][e];
        const float dyg = dyv * g;
        drv[e] = dyv - dyg;                                        // dy (1 - g)
        dgp[e] = (o - res) * (dyg - dyg * g);                      // dy (o - res) g (1 - g)
        dnr[e] = n > 0.f ? dyg : 0.f;
      }
      *reinterpret_cast<frag8*>(bufA + t * TH_TILE + prow * TH_PITCH + co) = th_pack8(dgp);
      if (cur_valid) *reinterpret_cast<frag8*>(dxp + t * tstride) = th_pack8(drv);   // (a clamped lane must not touch its alias)
      dn0[t] = th_pack8(dnr);
    }
    TH_ST2(2);
    __syncthreads();                                               // B: dgpre[t] complete
    TH_ST2(3);
    // ---------------- S3: gate^T, dn, GroupNorm backward ----------------
    frag8 dcf[TH_T];
    {
      float S1 = 0.f, S2 = 0.f;
      frag8 dxh[TH_T];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const Tile2 gt = th_get(bufA + t * TH_TILE, prow, kc);
        float dd[8];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          f32x4 bacc = {(float)dn0[t][4 * mm], (float)dn0[t][4 * mm + 1], (float)dn0[t][4 * mm + 2], (float)dn0[t][4 * mm + 3]};
#pragma unroll
          for (int s = 0; s < 2; ++s) bacc = mfma16(wgT_own[(mm * 2 + s) * 64], gt.f[s], bacc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 4 * mm + r;
            const float dnv = bacc[r], xv = xh[t][mm][r];
            dgam[e] = fmaf(dnv, xv, dgam[e]);
            dbet[e] += dnv;
            const float d = dnv * gw[e];
            dd[e] = d;
            S1 += d;
            S2 = fmaf(d, xv, S2);
          }
        }
        dxh[t] = th_pack8(dd);
      }
      const float m1 = S1 * (1.f / 40.f), m2 = S2 * (1.f / 40.f);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float dc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dc[e] = rstd * ((float)dxh[t][e] - m1 - xh[t][e >> 2][e & 3] * m2);
        dcf[t] = th_pack8(dc);
      }
    }
    TH_ST2(4);
    // ---------------- P2: gate weight gradient (rows 16q.., columns 32h..) ----------------
#pragma unroll
    for (int t = 0; t < TH_T; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pix0 = ks * 32 + 8 * kc;
        const frag8 af = th_tr(bufA + t * TH_TILE, pix0, q * 16, r16);
#pragma unroll
        for (int i = 0; i < 2; ++i) accG[i] = mfma16(af, th_tr(bufB + t * TH_TILE, pix0, (2 * h + i) * 16, r16), accG[i]);
        accGb = mfma16(af, ones, accGb);
      }
    }
    TH_ST2(5);
    __syncthreads();                                               // C: everyone is done with n[t], dgpre[t]
    TH_ST2(6);
    // ---------------- P3: publish dconv[t], x[t] ----------------
#pragma unroll
    for (int t = 0; t < TH_T; ++t) {
      *reinterpret_cast<frag8*>(bufA + t * TH_TILE + prow * TH_PITCH + co) = dcf[t];
      *reinterpret_cast<frag8*>(bufB + t * TH_TILE + prow * TH_PITCH + co) = MASK ? th_mul8(xo[t], m_own) : xo[t];   // the conv input
    }
    __syncthreads();                                               // D
    TH_ST2(7);
    // ---------------- dx[t'] = sum_k W_k^T dconv[t' - (k-1) d] + dres[t'] (this wave's 8 channels per lane) ----------------
    {
      f32x4 dxa[TH_T][2];
      frag8 drm[TH_T];                                               // (MASK) dres is added after the mask scales the conv path
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const frag8 drt = *reinterpret_cast<const frag8*>(dxp + t * tstride);
        drm[t] = drt;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
          dxa[t][mm] = MASK ? f32x4{0.f, 0.f, 0.f, 0.f}
                            : f32x4{(float)drt[4 * mm], (float)drt[4 * mm + 1], (float)drt[4 * mm + 2], (float)drt[4 * mm + 3]};
      }
      Tile2 dct[TH_T];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) dct[t] = th_get(bufA + t * TH_TILE, prow, kc);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const frag8 wf = wcT_own[((k * 4 + mm) * 2 + s) * 64];
#pragma unroll
            for (int tp = 0; tp < TH_T; ++tp)
              if (th_valid<DIL>(tp, 2 - k)) dxa[tp][mm] = mfma16(wf, dct[tp - (k - 1) * DIL].f[s], dxa[tp][mm]);
          }
      if (cur_valid) {
#pragma unroll
        for (int t = 0; t < TH_T; ++t) {
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = MASK ? fmaf((float)m_own[e], dxa[t][e >> 2][e & 3], (float)drm[t][e]) : dxa[t][e >> 2][e & 3];
          *reinterpret_cast<frag8*>(dxp + t * tstride) = th_pack8(y);
        }
      }
    }
    TH_ST2(8);
    // ---------------- P4: conv weight gradients  dW_k += dconv[tp - (k-1) d]^T x[tp] ----------------
#pragma unroll
    for (int tp = 0; tp < TH_T; ++tp) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pix0 = ks * 32 + 8 * kc;
        frag8 bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) bf[i] = th_tr(bufB + tp * TH_TILE, pix0, (2 * h + i) * 16, r16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (!th_valid<DIL>(tp, 2 - k)) continue;
          const frag8 af = th_tr(bufA + (tp - (k - 1) * DIL) * TH_TILE, pix0, q * 16, r16);
#pragma unroll
          for (int i = 0; i < 2; ++i) accC[k][i] = mfma16(af, bf[i], accC[k][i]);
          if (k == 1) accCb = mfma16(af, ones, accCb);
        }
      }
    }
    TH_ST2(9);
    __syncthreads();                                               // E: tiles are rewritten by the next workgroup tile
    TH_ST2(10);
  }
#ifdef TH_STAMPS
  __syncthreads();
  if (tid < 96) th_dbg[(size_t)blockIdx.x * 96 + tid] = (&th_ts[0][0])[tid];
#endif
  // ---------------- d gamma / d beta: reduce over the 16 pixel lanes, then over the 4 pixel quarters ----------------
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { dgam[e] += __shfl_xor(dgam[e], off, 64); dbet[e] += __shfl_xor(dbet[e], off, 64); }
  }
  if (px == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { gacc_lds[(q * 2 + 0) * 64 + co + e] = dgam[e]; gacc_lds[(q * 2 + 1) * 64 + co + e] = dbet[e]; }
  }
  // ---------------- write this workgroup's slab (rows 16q.., columns 32h..) ----------------
  float* my = slab + (int64_t)blockIdx.x * TH_SLAB;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(k * 64 + q * 16 + kc * 4 + r) * 64 + (2 * h + i) * 16 + r16] = accC[k][i][r];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) my[(3 * 64 + q * 16 + kc * 4 + r) * 64 + (2 * h + i) * 16 + r16] = accG[i][r];
  if (r16 == 0 && h == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      my[4 * 64 * 64 + q * 16 + kc * 4 + r] = accCb[r];
      my[4 * 64 * 64 + 64 + q * 16 + kc * 4 + r] = accGb[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * 64; i += 512) {
    const int which = i >> 6, c = i & 63;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += gacc_lds[(w * 2 + which) * 64 + c];
    my[4 * 64 * 64 + 128 + i] = s;
  }
}

// packs conv taps, gate, gate^T and conv^T taps (lane-quarter A-operand images) in one launch
__global__ void tcn_hot_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ Wc, const float* __restrict__ Wg) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int k = 0; k < 3; ++k) pack_weights_lds<bf16, 2>(dst + k * 4 * 2 * 64, Wc + k, 64, 64, 4, 64 * 3, 3, tid, nt);
  frag8* p = dst + 3 * 4 * 2 * 64;
  pack_weights_lds<bf16, 2>(p, Wg, 64, 64, 4, 64, 1, tid, nt);
  p += 4 * 2 * 64;
  pack_weights_lds<bf16, 2>(p, Wg, 64, 64, 4, 1, 64, tid, nt);
  p += 4 * 2 * 64;
  for (int k = 0; k < 3; ++k) pack_weights_lds<bf16, 2>(p + k * 4 * 2 * 64, Wc + k, 64, 64, 4, 3, 64 * 3, tid, nt);   // Weff[o=ci][i=co] = Wc[co][ci][k]
}


static unsigned th_fwd_grid(int64_t npix) {
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (unsigned)g;
}
static constexpr size_t TH_FWD_LDS = (size_t)32 * 64 * sizeof(frag8) + 4 * 64 * sizeof(float);
static constexpr size_t TH_BWD_LDS = TH_PACK_BYTES + (size_t)(4 * 64 + 4 * 2 * 64) * sizeof(float) + (size_t)2 * TH_T * TH_TILE * sizeof(bf16);

template <int DIL>
static int th_launch_fwd(const void* x, const void* mask, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg,
                         void* y, int64_t npix, int HW, float eps, hipStream_t st) {
  if (mask != nullptr) {
    FRL_LAUNCH((tcn_hot_fwd_kernel<DIL, true>), dim3(th_fwd_grid(npix)), dim3(256), TH_FWD_LDS, st, (const bf16*)x, (const bf16*)mask, pk, bc, gw,
               gb, bg, (bf16*)y, npix, HW, eps);
  } else {
    FRL_LAUNCH((tcn_hot_fwd_kernel<DIL, false>), dim3(th_fwd_grid(npix)), dim3(256), TH_FWD_LDS, st, (const bf16*)x, (const bf16*)nullptr, pk, bc,
               gw, gb, bg, (bf16*)y, npix, HW, eps);
  }
  return 0;
}
template <int DIL>
static int th_launch_bwd(const void* x, const void* mask, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb,
                         const float* bg, void* dx, float* slab, unsigned grid, int64_t npix, int HW, float eps, hipStream_t st) {
  if (mask != nullptr) {
    auto kern = tcn_hot_bwd2_kernel<DIL, true>;
    FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TH_BWD_LDS));
    FRL_LAUNCH_AS("tcn_hot_bwd2_kernel", kern, dim3(grid), dim3(512), TH_BWD_LDS, st, (const bf16*)x, (const bf16*)mask, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)dx,
               slab, npix, HW, eps);
  } else {
    auto kern = tcn_hot_bwd2_kernel<DIL, false>;
    FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TH_BWD_LDS));
    FRL_LAUNCH_AS("tcn_hot_bwd2_kernel", kern, dim3(grid), dim3(512), TH_BWD_LDS, st, (const bf16*)x, (const bf16*)nullptr, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)dx,
               slab, npix, HW, eps);
  }
  return 0;
}

static int g_th_force_bwd2 = 0;

// Packed image of one block (layout of tcn_hot_pack_kernel): from the caller's image cache when one is active, else packed into `ws_pk`.
static const frag8* th_packed(const float* conv_w, const float* gate_w, frag8* ws_pk, hipStream_t stream) {
  FrlPackJob jobs[8];
  const size_t blk = (size_t)4 * 2 * 64 * sizeof(frag8);          // one [4][2][64] fragment block
  for (int k = 0; k < 3; ++k) jobs[k] = frl_pack_job_pw(conv_w + k, k * blk, FRL_BF16, 2, 64, 64, 4, 64 * 3, 3);
  jobs[3] = frl_pack_job_pw(gate_w, 3 * blk, FRL_BF16, 2, 64, 64, 4, 64, 1);
  jobs[4] = frl_pack_job_pw(gate_w, 4 * blk, FRL_BF16, 2, 64, 64, 4, 1, 64);
  for (int k = 0; k < 3; ++k) jobs[5 + k] = frl_pack_job_pw(conv_w + k, (5 + k) * blk, FRL_BF16, 2, 64, 64, 4, 3, 64 * 3);
  bool hit = false;
  frag8* pk = ws_pk;
  if (void* img = frl_pack_cached(jobs, 8, TH_PACK_BYTES, &hit)) pk = (frag8*)img;
  if (!hit) FRL_LAUNCH(tcn_hot_pack_kernel, dim3(32), dim3(256), 0, stream, pk, conv_w, gate_w);
  return pk;
}

extern "C" {

// test hook: 1 routes every hot backward through the 8-wave kernel of this file (mask / ragged-tile path), 0 restores the dispatch
int frl_tcn_hot_force_generic_tiles(int on) { const int was = g_th_force_bwd2; g_th_force_bwd2 = on; return was; }
// 4 (default): tcn_hot_bwd4_kernel (two independent 4-wave subgroups per workgroup); 3: tcn_hot_bwd3_kernel (8 waves in lockstep).  Returns the
// previous value; for A/B measurements and the parity tests of both kernels.
static int g_th_bwd_variant = 4;
int frl_tcn_hot_bwd_variant(int v) { const int was = g_th_bwd_variant; if (v == 3 || v == 4) g_th_bwd_variant = v; return was; }

// 1 when the specialised kernels apply: bf16, 64 -> 64 channels, T = 5, 8 groups, identity residual, dilation 1 / 2 / 4
int frl_tcn_hot_supported(int T, int Cin, int Cout, int G, int dilation, int has_proj, int dtype) {
  return dtype == FRL_BF16 && Cin == 64 && Cout == 64 && T == TH_T && G == 8 && !has_proj && (dilation == 1 || dilation == 2 || dilation == 4);
}

// 1 when frl_tcn_hot_bwd accepts dx = NULL for this shape (the input needs no gradient: conv^T GEMM and dx store are skipped)
int frl_tcn_hot_bwd_nodx_supported(int64_t npix, int HW) { return (th_bwd3_supported(npix, HW) && !g_th_force_bwd2) ? 1 : 0; }

size_t frl_tcn_hot_fwd_workspace_bytes(void) { return TH_PACK_BYTES; }
size_t frl_tcn_hot_bwd_workspace_bytes(int64_t npix) { return (size_t)th_bwd_grid(npix) * TH_SLAB * sizeof(float) + 256 + TH_PACK_BYTES; }

// x, y [B][5][HW][64] bf16; parameters float32 in the reference layouts (conv_w [64][64][3], gate_w [64][64]); drop_mask: NULL, or the
// Dropout1d mask [B][HW][64] bf16 (0 or 1/(1-p)) of a training-mode block (conv input = x .* mask, residual = x; tcn.py:53,89-90)
int frl_tcn_hot_fwd(const void* x, const void* drop_mask, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b, const float* gate_w,
                    const float* gate_b, void* y, int64_t npix, int HW, int dilation, float eps, void* ws, size_t ws_bytes,
                    hipStream_t stream) {
  if (npix <= 0 || HW <= 0) return frl_fail(-2, "tcn_hot_fwd: empty input");
  if (ws == nullptr || ws_bytes < TH_PACK_BYTES) return frl_fail(-4, "tcn_hot_fwd: workspace too small");
  const frag8* pk = th_packed(conv_w, gate_w, (frag8*)ws, stream);
  int rc = -2;
  if (dilation == 1) rc = th_launch_fwd<1>(x, drop_mask, pk, conv_b, gn_w, gn_b, gate_b, y, npix, HW, eps, stream);
  else if (dilation == 2) rc = th_launch_fwd<2>(x, drop_mask, pk, conv_b, gn_w, gn_b, gate_b, y, npix, HW, eps, stream);
  else if (dilation == 4) rc = th_launch_fwd<4>(x, drop_mask, pk, conv_b, gn_w, gn_b, gate_b, y, npix, HW, eps, stream);
  else return frl_fail(-2, "tcn_hot_fwd: dilation must be 1, 2 or 4");
  if (rc) return rc;
  return frl_check_launch("tcn_hot_fwd");
}

// The dense phase chain forward in one launch: x [B][5][HW][64] -> y1, y2, y3 (outputs of the blocks with dilation 1, 2, 4; same shape) and
// h [B][5][HW][Ch] = head_w y3 + head_b (Ch <= 16, a multiple of 4).  Block parameters as in frl_tcn_hot_fwd, one set per block.
size_t frl_tcn_chain_fwd_workspace_bytes(void) { return 3 * TH_PACK_BYTES + 4096; }
// A/B hook: 1 = every wave walks its own fixed tile sequence (the round-3 first version), 0 (default) = tiles handed out by an LDS counter
static int g_thc_static = 0;
int frl_tcn_chain_static_tiles(int on) { const int was = g_thc_static; g_thc_static = on ? 1 : 0; return was; }
int frl_tcn_chain_fwd(const void* x, const float* const* conv_w, const float* const* conv_b, const float* const* gn_w, const float* const* gn_b,
                      const float* const* gate_w, const float* const* gate_b, const float* head_w, const float* head_b, void* y1, void* y2, void* y3,
                      void* h, int64_t npix, int HW, int Ch, float eps, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (npix <= 0 || HW <= 0) return frl_fail(-2, "tcn_chain_fwd: empty input");
  if (Ch < 4 || Ch > 16 || (Ch & 3)) return frl_fail(-2, "tcn_chain_fwd: head width must be 4, 8, 12 or 16");
  if (ws == nullptr || ws_bytes < frl_tcn_chain_fwd_workspace_bytes()) return frl_fail(-4, "tcn_chain_fwd: workspace too small");
  ThChainArgs a;
  char* w = (char*)ws;
  for (int b = 0; b < 3; ++b) {
    a.pk[b] = th_packed(conv_w[b], gate_w[b], reinterpret_cast<frag8*>(w + b * TH_PACK_BYTES), stream);
    a.bc[b] = conv_b[b]; a.gw[b] = gn_w[b]; a.gb[b] = gn_b[b]; a.bg[b] = gate_b[b];
  }
  {
    FrlPackJob job = frl_pack_job_pw(head_w, 0, FRL_BF16, 2, Ch, 64, 1, 64, 1);
    bool hit = false;
    frag8* pk = reinterpret_cast<frag8*>(w + 3 * TH_PACK_BYTES);
    if (void* img = frl_pack_cached(&job, 1, (size_t)2 * 64 * sizeof(frag8), &hit)) pk = (frag8*)img;
    if (!hit) FRL_LAUNCH((pack_weights_kernel<bf16, 2>), dim3(1), dim3(128), 0, stream, pk, head_w, Ch, 64, 1, (int64_t)64, (int64_t)1);
    a.pkh = pk;
  }
  a.bh = head_b;
  a.y[0] = (bf16*)y1; a.y[1] = (bf16*)y2; a.y[2] = (bf16*)y3; a.h = (bf16*)h;
  int64_t g = ((npix + 15) / 16 + 7) / 8;
  if (g > 256) g = 256;
  const size_t lds = (size_t)(3 * THC_IMG + 2 * 64) * sizeof(frag8) + (3 * 256 + 16) * sizeof(float) + 16;      // (+ the tile counter)
  FRL_HIP(hipFuncSetAttribute((const void*)tcn_chain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH(tcn_chain_fwd_kernel, dim3((unsigned)g), dim3(512), lds, stream, (const bf16*)x, a, npix, HW, Ch, eps, g_thc_static ? 0 : 1);
  return frl_check_launch("tcn_chain_fwd");
}

// one launch: dx [B][5][HW][64] bf16 and all parameter gradients (float32, reference layouts); drop_mask as in the forward
int frl_tcn_hot_bwd(const void* x, const void* drop_mask, const void* dy, const float* conv_w, const float* conv_b, const float* gn_w, const float* gn_b,
                    const float* gate_w, const float* gate_b, void* dx, float* d_conv_w, float* d_conv_b, float* d_gn_w, float* d_gn_b,
                    float* d_gate_w, float* d_gate_b, int64_t npix, int HW, int dilation, float eps, void* ws, size_t ws_bytes,
                    hipStream_t stream) {
  if (npix <= 0 || HW <= 0) return frl_fail(-2, "tcn_hot_bwd: empty input");
  if (npix >= (int64_t)1 << 31) return frl_fail(-2, "tcn_hot_bwd: more than 2^31 pixels per launch");
  if (ws == nullptr || ws_bytes < frl_tcn_hot_bwd_workspace_bytes(npix)) return frl_fail(-4, "tcn_hot_bwd: workspace too small");
  const unsigned grid = th_bwd_grid(npix);
  float* slab = (float*)ws;
  const frag8* pk = th_packed(conv_w, gate_w,
                              reinterpret_cast<frag8*>(reinterpret_cast<char*>(ws) + (((size_t)grid * TH_SLAB * sizeof(float) + 255) / 256) * 256), stream);
  int rc = -2;
  if (dx == nullptr && !(drop_mask == nullptr && th_bwd3_supported(npix, HW) && !g_th_force_bwd2))
    return frl_fail(-2, "tcn_hot_bwd: dx may be NULL (input without gradient) only on the tcn_hot_bwd3 route (no mask, HW % 64 == 0)");
  if (drop_mask == nullptr && th_bwd3_supported(npix, HW) && !g_th_force_bwd2)      // the measured configuration: tcn_hot_bwd4.hip / tcn_hot_bwd3.hip
    rc = (g_th_bwd_variant == 4 && th_bwd4_supported(npix, HW))
             ? th_bwd4_launch(dilation, x, dy, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream)
             : th_bwd3_launch(dilation, x, dy, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream);
  else if (dilation == 1) rc = th_launch_bwd<1>(x, drop_mask, dy, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream);
  else if (dilation == 2) rc = th_launch_bwd<2>(x, drop_mask, dy, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream);
  else if (dilation == 4) rc = th_launch_bwd<4>(x, drop_mask, dy, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream);
  else return frl_fail(-2, "tcn_hot_bwd: dilation must be 1, 2 or 4");
  if (rc) return rc;
  launch_slab_reduce_deferrable<float, ThEpi>((const float*)slab, (int)grid, (int64_t)TH_SLAB, ThEpi{d_conv_w, d_gate_w, d_conv_b, d_gate_b, d_gn_w, d_gn_b},
                                   stream);
  return frl_check_launch("tcn_hot_bwd");
}

// The last block of the phase encoder together with the backward-data of the 1x1 phase head behind it (representation.py:169): dh
// [B][5][HW][Ch] bf16 is the head's output gradient, head_w [Ch][64] float32; dy = dh head_w is formed inside the kernel (matrix cores) and
// never written.  Same outputs as frl_tcn_hot_bwd.  Needs the two-subgroup kernel (no mask, HW % 64 == 0, dilation 4, dx != NULL).
int frl_tcn_hot_bwd_head_supported(int64_t npix, int HW, int Ch) {
  return (th_bwd4_supported(npix, HW) && !g_th_force_bwd2 && g_th_bwd_variant == 4 && Ch >= 4 && Ch <= 16 && (Ch & 3) == 0) ? 1 : 0;
}
size_t frl_tcn_hot_bwd_head_workspace_bytes(int64_t npix) { return frl_tcn_hot_bwd_workspace_bytes(npix) + 4096; }
int frl_tcn_hot_bwd_head(const void* x, const void* dh, const float* head_w, int Ch, const float* conv_w, const float* conv_b, const float* gn_w,
                         const float* gn_b, const float* gate_w, const float* gate_b, void* dx, float* d_conv_w, float* d_conv_b, float* d_gn_w,
                         float* d_gn_b, float* d_gate_w, float* d_gate_b, int64_t npix, int HW, int dilation, float eps, void* ws, size_t ws_bytes,
                         hipStream_t stream) {
  if (npix <= 0 || HW <= 0) return frl_fail(-2, "tcn_hot_bwd_head: empty input");
  if (!frl_tcn_hot_bwd_head_supported(npix, HW, Ch)) return frl_fail(-2, "tcn_hot_bwd_head: unsupported shape (HW % 64, head width 4..16 in fours)");
  if (ws == nullptr || ws_bytes < frl_tcn_hot_bwd_head_workspace_bytes(npix)) return frl_fail(-4, "tcn_hot_bwd_head: workspace too small");
  const unsigned grid = th_bwd_grid(npix);
  float* slab = (float*)ws;
  char* wpk = reinterpret_cast<char*>(ws) + (((size_t)grid * TH_SLAB * sizeof(float) + 255) / 256) * 256;
  const frag8* pk = th_packed(conv_w, gate_w, reinterpret_cast<frag8*>(wpk), stream);
  const frag8* whp;
  {                                                                  // W_h^T as an A-operand image: rows = the 64 block channels, contraction = Ch (padded to 32)
    FrlPackJob job = frl_pack_job_pw(head_w, 0, FRL_BF16, 1, 64, Ch, 4, 1, 64);
    bool hit = false;
    frag8* img = reinterpret_cast<frag8*>(wpk + TH_PACK_BYTES);
    if (void* c = frl_pack_cached(&job, 1, (size_t)4 * 64 * sizeof(frag8), &hit)) img = (frag8*)c;
    if (!hit) FRL_LAUNCH((pack_weights_kernel<bf16, 1>), dim3(1), dim3(256), 0, stream, img, head_w, 64, Ch, 4, (int64_t)1, (int64_t)64);
    whp = img;
  }
  const int rc = th_bwd4_launch(dilation, x, dh, pk, conv_b, gn_w, gn_b, gate_b, dx, slab, grid, npix, HW, eps, stream, whp, Ch);
  if (rc) return rc;
  launch_slab_reduce_deferrable<float, ThEpi>((const float*)slab, (int)grid, (int64_t)TH_SLAB, ThEpi{d_conv_w, d_gate_w, d_conv_b, d_gate_b, d_gn_w, d_gn_b},
                                   stream);
  return frl_check_launch("tcn_hot_bwd_head");
}

}  // extern "C"
