// Backward of the hot-configuration GatedResidualBlock (frl/models/tcn.py:78-111; bf16, 64 -> 64 channels, T = 5, 8-channel GroupNorm
// groups, identity residual, dilation 1 / 2 / 4, no Dropout1d mask, HW % 64 == 0): ONE launch -> dx and all six parameter gradients.
//
// Same decomposition as tcn_hot_bwd2_kernel (tcn_hot.hip): 8 waves per workgroup, wave (q, h) owns the 16 pixels of quarter q and the
// channel half h of every lane quarter; what changed is where the data lives, so that nothing spills and nothing takes a second trip
// through HBM (algorithmic traffic: x and dy read once, dx written once = 3 x 128 B per (pixel, t)):
//   * x[t] of a 64-pixel tile is brought in by LDS-DMA (global_load_lds_dwordx4) and STAYS in LDS for the whole tile: conv operands,
//     the residual and the weight-gradient operand are read from there, no register copy of x exists.  The DMA of tile i+1 is issued
//     as soon as tile i's n[t] buffer is free (after barrier C) and lands behind the conv^T / weight-gradient phases.
//   * three unpadded 40 KB tile buffers [t][64 px][128 B]: X | N | A with X and N swapping roles every tile; a chunk swizzle
//     chunk' = chunk ^ f(px) (f found by tools/diag/lds_swizzle_search.py) makes the 16-byte row accesses AND the transposing
//     ds_read_b64_tr_b16 reads bank-conflict free without padding (the DMA writes LDS linearly, so the swizzle sits on its source address).
//   * ONE packed weight image (conv taps + gate, 32 KB): the transposed operands of the gate^T / conv^T GEMMs are fetched from the
//     same image with ds_read_b64_tr_b16 (a 4 x 16 block of it IS a transposed fragment), which frees the 32 KB of the second image.
//   * dres = dy (1 - g) is carried in registers (packed bf16) instead of being parked in dx.
//
//   E' wait for the DMA of x, barrier            conv -> GroupNorm statistics -> n[t]            publish n[t] (own 8 channels)   | A
//   S2 gate GEMM, sigmoid, dgpre[t], dres[t] (registers), relu path of dn                        publish dgpre[t]                | B
//   S3 gate^T GEMM -> dn -> d gamma, d beta, GroupNorm backward -> dconv[t];  P2 gate weight gradient                            | C
//      DMA of the next tile's x into the N buffer, next dy -> registers;      publish dconv[t]                                   | D
//   dx = conv^T(dconv) + dres (stored);  P4 conv weight gradients
#include "tcn_hot_common.hpp"

#define B3_TT (64 * 128)                       // bytes of one time step of a tile buffer
#define B3_TB (TH_T * B3_TT)                   // 40960: one tile buffer
#define B3_W 0                                 // packed conv taps [3][4][2][64] + gate [4][2][64] fragments
#define B3_TAB 32768                           // conv bias | gamma | beta | -log2e * gate bias
#define B3_GACC (B3_TAB + 1024)                // [4 quarters][2][64] d gamma / d beta partial sums
#define B3_TILE (B3_GACC + 2048 + 1024)        // 36864
#define B3_LDS (B3_TILE + 3 * B3_TB)           // 159744 <= 160 KiB

// chunk swizzle of the tile buffers: 16-byte chunk c of pixel row r is stored at chunk position c ^ b3_swz(r)
__device__ __forceinline__ int b3_swz(int r) {
  return ((r >> 1) & 1) | ((((r >> 1) ^ (r >> 2)) & 1) << 1) | (((r ^ (r >> 2) ^ (r >> 3)) & 1) << 2);
}

__device__ __forceinline__ frag8 b3_ld(const char* smem, int off) { return *reinterpret_cast<const frag8*>(smem + off); }
__device__ __forceinline__ void b3_st(char* smem, int off, const frag8& v) { *reinterpret_cast<frag8*>(smem + off) = v; }
__device__ __forceinline__ bf16x4 b3_tr4(const char* smem, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(smem + off));
}
__device__ __forceinline__ frag8 b3_join(const bf16x4& lo, const bf16x4& hi) { return frag8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }
// transposed weight fragment out of the forward image (element (lane', e') of block (m', s') of W^T): two 4 x 16 blocks 2 KB apart
__device__ __forceinline__ frag8 b3_wT(const char* smem, int a) { return b3_join(b3_tr4(smem, a), b3_tr4(smem, a + 2048)); }

// Lane constants of the address arithmetic live PACKED in three registers (pk0: own | full tile offsets; pk1: transposing-read row
// bases and swizzle bits; pk2: weight-transpose offset, global element offset, pixel, quarter).  The head of every phase takes fresh
// (opaque) copies and unpacks what it needs with one VALU operation each: rebuilt inside the tile loop like this, the address
// arithmetic costs a few instructions per phase; hoisted out of it by the compiler it pinned some thirty loop-invariant registers next
// to the accumulators and pushed the allocation over 256 registers (every spill reload sits in the same vmcnt queue as the loads).
#define B3_ADDR()                                                                          \
  unsigned k0_ = pk0, k1_ = pk1, k2_ = pk2;                                                \
  asm volatile("" : "+v"(k0_), "+v"(k1_), "+v"(k2_));                                     \
  const int oo = (int)(k0_ & 0xffffu), fo = (int)(k0_ >> 16);                              \
  const int tb0 = (int)(k1_ & 0xfffu), tb1 = (int)((k1_ >> 12) & 0xfffu);                  \
  const int tf0 = (int)((k1_ >> 24) & 7u), tf1 = (int)((k1_ >> 27) & 7u);                  \
  const int wtr_ = (int)(k2_ & 0x3ffu), px_ = (int)((k2_ >> 22) & 15u), kc_ = (int)(k2_ >> 26); \
  const unsigned le_ = (k2_ >> 10) & 0xfffu;                                               \
  const int wfwd_ = ((2 * h) * 2 * 64 + px_ + 16 * kc_) * 16;                              \
  (void)fo; (void)oo; (void)tb0; (void)tb1; (void)tf0; (void)tf1; (void)wtr_; (void)wfwd_; (void)px_; (void)kc_; (void)le_
// B operand that sums the k dimension into output column `col` (bias gradients ride on the matrix cores): ones in the lanes of pixel
// `col`; the conv bias gradient accumulates in column 0 and the gate bias gradient in column 1 of ONE accumulator
#define B3_ONES(col)                                                          \
  const bf16 one_ = px_ == (col) ? (bf16)1.f : (bf16)0.f;                    \
  const frag8 ones = frag8{one_, one_, one_, one_, one_, one_, one_, one_}
// per-channel parameters of the lane's 8 channels, re-read from the LDS table inside each phase (two ds_read_b128 each) instead of
// being carried in 24 registers across the whole tile; the opaque zero keeps the compiler from merging the reads of different phases
#define B3_PARAM(name, which)                                                 \
  float name[8];                                                              \
  {                                                                           \
    int z_ = 0;                                                               \
    asm volatile("" : "+s"(z_));                                             \
    const f32x4* p_ = reinterpret_cast<const f32x4*>(tab + (which) * 64 + 16 * kc_ + 8 * h + z_); \
    const f32x4 a_ = p_[0], b_ = p_[1];                                       \
    name[0] = a_[0]; name[1] = a_[1]; name[2] = a_[2]; name[3] = a_[3];       \
    name[4] = b_[0]; name[5] = b_[1]; name[6] = b_[2]; name[7] = b_[3];       \
  }
#define B3_TR8(base, cb) b3_join(b3_tr4(smem, (base) + tb0 + (((2 * (cb)) ^ tf0) << 4)), b3_tr4(smem, (base) + tb1 + (((2 * (cb)) ^ tf1) << 4)))
#define B3_WT(matbase, mm, s) b3_wT(smem, (matbase) + ((2 * (s)) * 2 + h) * 1024 + wtr_ + 8 * (mm))
// sum over the 16 lanes of a DPP row (= the 16 pixels of a lane quarter); every lane of the row receives the total
template <int CTRL> __device__ __forceinline__ float b3_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float b3_row_sum(float v) {
  v += b3_dpp<0xB1>(v);                                            // quad_perm [1,0,3,2]
  v += b3_dpp<0x4E>(v);                                            // quad_perm [2,3,0,1]
  v += b3_dpp<0x124>(v);                                           // row_ror:4
  v += b3_dpp<0x128>(v);                                           // row_ror:8
  return v;
}
#define B3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define B3_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Diagnostic build only (tools/diag/tcn_bwd3_stamps.hip defines B3_STAMPS): s_memtime stamps at the phase boundaries, accumulated in the
// spare LDS behind the d gamma / d beta slots and written to a buffer nothing else reads.  The product library never defines it.
#ifdef B3_STAMPS
__device__ unsigned long long* b3_dbg;
#define B3_ST(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) b3_ts[wave * 12 + (i)] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define B3_ST(i) do { } while (0)
#endif

// WANT_DX = false: the block's input is data (the first block of the phase path reads the tile itself): the conv^T GEMM, the residual
// gradient and the dx store are compiled out.
template <int DIL, bool WANT_DX>
__global__ __launch_bounds__(512, 2) void tcn_hot_bwd3_kernel(const bf16* __restrict__ X, const bf16* __restrict__ DY, const frag8* __restrict__ Wpk,
                                                              const float* __restrict__ bc, const float* __restrict__ gn_w,
                                                              const float* __restrict__ gn_b, const float* __restrict__ bg, bf16* __restrict__ DX,
                                                              float* __restrict__ slab, int ntile, int HW, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem + B3_TAB);
  float* gacc_lds = reinterpret_cast<float*>(smem + B3_GACC);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 3, h = wave >> 2;
  const int px = lane & 15, kc = lane >> 4;
  const int prow = q * 16 + px;
  const int co = 16 * kc + 8 * h;                                  // first of this lane's 8 channels
  const int64_t tstride = (int64_t)HW * 64;
  // tile-buffer offsets of this lane's pixel row: own 16-byte chunk (its 8 channels) and the two chunks of its channel quarter
  const int fsw = b3_swz(prow);
  const int own_off = prow * 128 + (((2 * kc + h) ^ fsw) << 4);
  const int full_off = prow * 128 + (((2 * kc) ^ fsw) << 4);      // fragment s sits at full_off ^ (s << 4)
  // transposing reads (k-strided fragments: 8 consecutive pixels of one channel): lane (r16 = px, kc) addresses pixel row
  // 32 ks + 8 kc + (r16 >> 2) + 4 hi and the 8-byte piece (r16 & 3) of the 16-channel block cb
  int trb[2], trf[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 8 * kc + (px >> 2) + 4 * hi;
    trb[hi] = row * 128 + 8 * (px & 1);
    trf[hi] = b3_swz(row) ^ ((px >> 1) & 1);
  }
  const int wtr = (4 * kc + (px >> 2) + 16 * (px & 3)) * 16;      // transposed weight fragments: + block * 1024 + 8 * mm
  const unsigned lane_el = (unsigned)(prow * 64 + co);             // this lane's pixel, its 8 channels (element offset inside a tile's time step)
  const unsigned pk0 = (unsigned)own_off | ((unsigned)full_off << 16);
  const unsigned pk1 = (unsigned)trb[0] | ((unsigned)trb[1] << 12) | ((unsigned)trf[0] << 24) | ((unsigned)trf[1] << 27);
  const unsigned pk2 = (unsigned)wtr | (lane_el << 10) | ((unsigned)px << 22) | ((unsigned)kc << 26);
  const unsigned tps = (unsigned)HW >> 6;                          // tiles per sample
  // ---- global addressing: wave-uniform 64-bit tile base (SGPRs) + 32-bit lane offset (elements)
  auto tile_base = [&](int wt) -> int64_t {                        // element offset of (t = 0, first pixel of the tile, channel 0)
    const unsigned b = (unsigned)wt / tps;                         // sample of the tile (tiles never straddle samples: HW % 64 == 0)
    return ((int64_t)wt * 64 + (int64_t)b * (TH_T - 1) * HW) * 64;
  };
  // ---- DMA of one tile's x into a tile buffer: wave w issues pieces j = 5 w .. 5 w + 4 (1 KB = 8 pixel rows of one time step each).
  // Issued through inline asm: hipcc orders every later LDS access behind a builtin LDS-DMA with s_waitcnt vmcnt(0) (it cannot know that
  // the DMA targets another buffer), which parked all eight waves for a full HBM latency right after the issue.  Invisible to the
  // compiler, the transfers stay in flight across the conv^T / weight-gradient phases; the E' barrier waits for them explicitly.
  // (A hidden vector-memory operation only makes the compiler's own counted waits stricter, never looser: they count from the youngest.)
  auto dma_tile = [&](int wt, int dst) {
    const bf16* xb = X + tile_base(wt);
    unsigned k2_ = pk2;
    asm volatile("" : "+v"(k2_));                                  // (rebuilt per call: see B3_ADDR)
    const unsigned ln = ((k2_ >> 22) & 15u) + 16u * (k2_ >> 26);
    const unsigned r8 = ln >> 3, c8 = ln & 7u;
    unsigned voff[2];                                              // byte offset of this lane's 16 bytes inside a piece, pb even / odd
#pragma unroll
    for (int o = 0; o < 2; ++o) voff[o] = (r8 * 64u + ((c8 ^ (unsigned)b3_swz((int)(8u * o + r8))) * 8u)) * 2u;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int j = wave * 5 + jj, t = j >> 3, pb = j & 7;
      const bf16* sb = xb + (int64_t)t * tstride + pb * (8 * 64);  // wave-uniform: time step and 8-row block of the piece
      const unsigned vo = (pb & 1) ? voff[1] : voff[0];
      const int ldst = dst + j * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(sb), "s"(ldst) : "memory");
    }
  };

  int xoff = B3_TILE, noff = B3_TILE + B3_TB;
  const int aoff = B3_TILE + 2 * B3_TB;
  dma_tile(blockIdx.x, xoff);
  frag8 dyn[TH_T];
  {
    const bf16* dyb = DY + tile_base(blockIdx.x);
#pragma unroll
    for (int t = 0; t < TH_T; ++t) dyn[t] = *reinterpret_cast<const frag8*>(dyb + t * tstride + lane_el);
  }
  copy_frags_lds<bf16>(reinterpret_cast<frag8*>(smem + B3_W), Wpk, 32 * 64, tid, 512);
  gacc_lds[tid] = 0.f;                                             // [8 waves][4 kc][8 d gamma | 8 d beta]
  if (tid < 64) {
    tab[tid] = bc[tid];
    tab[64 + tid] = gn_w[tid];
    tab[128 + tid] = gn_b[tid];
    tab[192 + tid] = -1.44269504088896f * bg[tid];
  }

#ifdef B3_STAMPS
  unsigned long long* b3_ts = reinterpret_cast<unsigned long long*>(smem + B3_GACC + 2048);   // [8 waves][12]
  if (tid < 96) b3_ts[tid] = 0ull;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_start = t_prev, r_start = __builtin_amdgcn_s_memrealtime();
#endif
  f32x4 accC[3][2], accG[2], accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) accC[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) accG[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int wt = blockIdx.x; wt < ntile; wt += gridDim.x) {
    bf16* dxb = DX + tile_base(wt);
    frag8 dyo[TH_T];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) dyo[t] = dyn[t];
    B3_ST(11);
    B3_BARRIER_ALL();                                              // E': x of this tile has landed, everyone left the previous tile
    B3_ST(0);
    // ---------------- conv(x) + bias of this lane's 8 channels, all time steps ----------------
    f32x4 xh[TH_T][2];
    {
      B3_ADDR();
      const f32x4* tcb = reinterpret_cast<const f32x4*>(tab + 16 * kc_ + 8 * h);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const f32x4 cb = tcb[mm];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) xh[t][mm] = cb;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag8 xs[TH_T];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) xs[t] = b3_ld(smem, xoff + t * B3_TT + (fo ^ (s << 4)));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const frag8 wf = b3_ld(smem, B3_W + wfwd_ + ((k * 4 + mm) * 2 + s) * 1024);
#pragma unroll
            for (int t = 0; t < TH_T; ++t)
              if (th_valid<DIL>(t, k)) xh[t][mm] = mfma16(wf, xs[t + (k - 1) * DIL], xh[t][mm]);
          }
      }
    }
    float rstd;
    {                                                              // exact two-pass statistics of this lane's group (8 ch x 5 t)
      float sp[4] = {0.f, 0.f, 0.f, 0.f};                           // four interleaved partial sums: 10-long dependency chains, not 40
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) sp[e & 3] += xh[t][e >> 2][e & 3];
      const float mean = ((sp[0] + sp[1]) + (sp[2] + sp[3])) * (1.f / 40.f);
      float qp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xh[t][e >> 2][e & 3] - mean; qp[e & 3] = fmaf(d, d, qp[e & 3]); }
      rstd = 1.f / sqrtf(((qp[0] + qp[1]) + (qp[2] + qp[3])) * (1.f / 40.f) + eps);
      // ---------------- S1: xhat (kept), n[t] -> N buffer ----------------
      const float nm = -mean * rstd;
      B3_ADDR();
      B3_PARAM(gw, 1);
      B3_PARAM(gb, 2);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float n[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaf(xh[t][e >> 2][e & 3], rstd, nm);
          xh[t][e >> 2][e & 3] = v;
          n[e] = fmaf(v, gw[e], gb[e]);
        }
        b3_st(smem, noff + t * B3_TT + oo, th_pack8(n));
      }
    }
    B3_ST(1);
    B3_BARRIER();                                                  // A: n[t] complete (all channels of every pixel)
    B3_ST(2);
    // ---------------- S2: gate, dgpre, dres, relu path of dn ----------------
    frag8 dn0[TH_T], dres[TH_T];
    {
      B3_ADDR();
      B3_PARAM(gw, 1);
      B3_PARAM(gb, 2);
      B3_PARAM(tnbg, 3);
      frag8 wg[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int s = 0; s < 2; ++s) wg[mm][s] = b3_ld(smem, B3_W + wfwd_ + ((12 + mm) * 2 + s) * 1024);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const frag8 nt0 = b3_ld(smem, noff + t * B3_TT + fo), nt1 = b3_ld(smem, noff + t * B3_TT + (fo ^ 16));
        const frag8 xo = b3_ld(smem, xoff + t * B3_TT + oo);   // residual path: x of this lane's 8 channels
        f32x4 gacc[2];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          gacc[mm] = mfma16(wg[mm][0], nt0, f32x4{0.f, 0.f, 0.f, 0.f});
          gacc[mm] = mfma16(wg[mm][1], nt1, gacc[mm]);
        }
        float dgp[8], drv[8], dnr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float n = fmaf(xh[t][e >> 2][e & 3], gw[e], gb[e]);
          const float dyv = (float)dyo[t][e];
          const float g = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(gacc[e >> 2][e & 3], -1.44269504088896f, tnbg[e])));
          const float o = fmaxf(n, 0.f);
          const float res = (float)xo[e];
          const float dyg = dyv * g;
          drv[e] = dyv - dyg;                                      // dy (1 - g)
          dgp[e] = (o - res) * (dyg - dyg * g);                    // dy (o - res) g (1 - g)
          dnr[e] = n > 0.f ? dyg : 0.f;
        }
        b3_st(smem, aoff + t * B3_TT + oo, th_pack8(dgp));
        if constexpr (WANT_DX) dres[t] = th_pack8(drv);
        dn0[t] = th_pack8(dnr);
      }
    }
    B3_ST(3);
    B3_BARRIER();                                                  // B: dgpre[t] complete
    B3_ST(4);
    // ---------------- S3: gate^T, dn, GroupNorm backward ----------------
    frag8 dcf[TH_T];
    auto phase_s3 = [&]() {
      B3_ADDR();
      B3_PARAM(gw, 1);
      frag8 wgT[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int s = 0; s < 2; ++s) wgT[mm][s] = B3_WT(B3_W + 24 * 1024, mm, s);
      float S1p[4] = {0.f, 0.f, 0.f, 0.f}, S2p[4] = {0.f, 0.f, 0.f, 0.f};
      float dgam[8], dbet[8];                                      // this tile's d gamma / d beta of the lane's 8 channels
#pragma unroll
      for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
      frag8 dxh[TH_T];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const frag8 gt0 = b3_ld(smem, aoff + t * B3_TT + fo), gt1 = b3_ld(smem, aoff + t * B3_TT + (fo ^ 16));
        float dd[8];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          f32x4 bacc = {(float)dn0[t][4 * mm], (float)dn0[t][4 * mm + 1], (float)dn0[t][4 * mm + 2], (float)dn0[t][4 * mm + 3]};
          bacc = mfma16(wgT[mm][0], gt0, bacc);
          bacc = mfma16(wgT[mm][1], gt1, bacc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 4 * mm + r;
            const float dnv = bacc[r], xv = xh[t][mm][r];
            dgam[e] = fmaf(dnv, xv, dgam[e]);
            dbet[e] += dnv;
            const float d = dnv * gw[e];
            dd[e] = d;
            S1p[r] += d;
            S2p[r] = fmaf(d, xv, S2p[r]);
          }
        }
        dxh[t] = th_pack8(dd);
      }
      // sum over the 16 pixels of the lane quarter (DPP), then one LDS add per (wave, quarter, channel): only this wave touches its
      // slots, in program order, so the sums are bit-reproducible
#pragma unroll
      for (int e = 0; e < 8; ++e) { dgam[e] = b3_row_sum(dgam[e]); dbet[e] = b3_row_sum(dbet[e]); }
      if (px_ == 0) {
        float* ga = gacc_lds + (wave * 4 + kc_) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          __hip_atomic_fetch_add(ga + e, dgam[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(ga + 8 + e, dbet[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      const float m1 = ((S1p[0] + S1p[1]) + (S1p[2] + S1p[3])) * (1.f / 40.f), m2 = ((S2p[0] + S2p[1]) + (S2p[2] + S2p[3])) * (1.f / 40.f);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float dc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dc[e] = rstd * ((float)dxh[t][e] - m1 - xh[t][e >> 2][e & 3] * m2);
        dcf[t] = th_pack8(dc);
      }
    };
    // ---------------- P2: gate weight gradient (rows 16q.., columns 32h..) ----------------
    auto phase_p2 = [&]() {
      B3_ADDR();
      B3_ONES(1);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const frag8 af = B3_TR8(aoff + t * B3_TT + ks * 4096, q);
#pragma unroll
          for (int i = 0; i < 2; ++i) accG[i] = mfma16(af, B3_TR8(noff + t * B3_TT + ks * 4096, 2 * h + i), accG[i]);
          accB = mfma16(af, ones, accB);
        }
      }
    };
    // Both need only what barrier B published.  P2 first: its transposing reads and MFMAs are issued while dn0 / xh are merely parked,
    // and the allocation closes without a spill in this order.  (Letting the two waves of a SIMD take the phases in opposite order --
    // one in the vector-ALU chain while its partner streams LDS reads into MFMAs -- costs 50 spilled registers at the join of the two
    // code paths, as a branch and as a two-trip loop alike.)
    phase_p2();
    B3_ST(5);
    phase_s3();
    B3_ST(6);
    B3_BARRIER();                                                  // C: everyone is done with n[t], dgpre[t]
    B3_ST(7);
    // ---------------- the next tile's x -> N buffer (LDS-DMA), its dy -> registers; publish dconv[t] ----------------
    const int wtn = wt + gridDim.x;
    if (wtn < ntile) dma_tile(wtn, noff);
    {                                                              // (unconditional, from a clamped tile: dyn must not stay live across the tile)
      const bf16* dyb = DY + tile_base(wtn < ntile ? wtn : ntile - 1);
      B3_ADDR();
#pragma unroll
      for (int t = 0; t < TH_T; ++t) dyn[t] = *reinterpret_cast<const frag8*>(dyb + t * tstride + le_);
    }
    {
      B3_ADDR();
#pragma unroll
      for (int t = 0; t < TH_T; ++t) b3_st(smem, aoff + t * B3_TT + oo, dcf[t]);
    }
    B3_ST(8);
    B3_BARRIER();                                                  // D (LDS only: the DMA and the dy loads stay in flight)
    B3_ST(9);
    // ---------------- dx[t'] = sum_k W_k^T dconv[t' - (k-1) d] + dres[t'] (this wave's 8 channels per lane) ----------------
    if constexpr (WANT_DX) {
      B3_ADDR();
      f32x4 dxa[TH_T][2];
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
          dxa[t][mm] = f32x4{(float)dres[t][4 * mm], (float)dres[t][4 * mm + 1], (float)dres[t][4 * mm + 2], (float)dres[t][4 * mm + 3]};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag8 dcs[TH_T];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) dcs[t] = b3_ld(smem, aoff + t * B3_TT + (fo ^ (s << 4)));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const frag8 wf = B3_WT(B3_W + k * 8 * 1024, mm, s);
#pragma unroll
            for (int tp = 0; tp < TH_T; ++tp)
              if (th_valid<DIL>(tp, 2 - k)) dxa[tp][mm] = mfma16(wf, dcs[tp - (k - 1) * DIL], dxa[tp][mm]);
          }
      }
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = dxa[t][e >> 2][e & 3];
        *reinterpret_cast<frag8*>(dxb + t * tstride + le_) = th_pack8(y);
      }
    }
    B3_ST(10);
    // ---------------- P4: conv weight gradients  dW_k += dconv[tp - (k-1) d]^T x[tp] ----------------
    {
      B3_ADDR();
      B3_ONES(0);
#pragma unroll
      for (int tp = 0; tp < TH_T; ++tp) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          frag8 bf[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) bf[i] = B3_TR8(xoff + tp * B3_TT + ks * 4096, 2 * h + i);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (!th_valid<DIL>(tp, 2 - k)) continue;
            const frag8 af = B3_TR8(aoff + (tp - (k - 1) * DIL) * B3_TT + ks * 4096, q);
#pragma unroll
            for (int i = 0; i < 2; ++i) accC[k][i] = mfma16(af, bf[i], accC[k][i]);
            if (k == 1) accB = mfma16(af, ones, accB);
          }
        }
      }
    }
    const int tmp = xoff; xoff = noff; noff = tmp;                 // the DMA target becomes X, the old X buffer receives the next n[t]
  }
  B3_BARRIER_ALL();
#ifdef B3_STAMPS
  if (tid < 96) b3_dbg[(size_t)blockIdx.x * 96 + tid] = b3_ts[tid];
  if (tid == 0) {                                                  // shader-clock cycles and 100 MHz ticks of the whole tile loop
    b3_dbg[(size_t)gridDim.x * 96 + 2048 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
    b3_dbg[(size_t)gridDim.x * 96 + 2048 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r_start;
  }
#endif
  // ---------------- write this workgroup's slab (rows 16q.., columns 32h..) ----------------
  B3_ADDR();
  float* my = slab + (int64_t)blockIdx.x * TH_SLAB;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(k * 64 + q * 16 + kc_ * 4 + r) * 64 + (2 * h + i) * 16 + px_] = accC[k][i][r];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) my[(3 * 64 + q * 16 + kc_ * 4 + r) * 64 + (2 * h + i) * 16 + px_] = accG[i][r];
  if (px_ < 2 && h == 0) {                                         // column 0: d conv bias, column 1: d gate bias
#pragma unroll
    for (int r = 0; r < 4; ++r) my[4 * 64 * 64 + 64 * px_ + q * 16 + kc_ * 4 + r] = accB[r];
  }
  __syncthreads();
  for (int i = tid; i < 2 * 64; i += 512) {
    const int which = i >> 6, c = i & 63;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += gacc_lds[((w + 4 * ((c >> 3) & 1)) * 4 + (c >> 4)) * 16 + 8 * which + (c & 7)];
    my[4 * 64 * 64 + 128 + i] = s;
  }
}

bool th_bwd3_supported(int64_t npix, int HW) { return HW > 0 && HW % 64 == 0 && npix % HW == 0 && npix / 64 < ((int64_t)1 << 30); }

template <int DIL>
static void b3_launch(const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg, void* dx,
                      float* slab, unsigned grid, int ntile, int HW, float eps, hipStream_t st) {
  if (dx != nullptr) {
    auto kern = tcn_hot_bwd3_kernel<DIL, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_LDS);
    FRL_LAUNCH_AS("tcn_hot_bwd3_kernel", kern, dim3(grid), dim3(512), B3_LDS, st, (const bf16*)x, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)dx, slab, ntile, HW, eps);
  } else {
    auto kern = tcn_hot_bwd3_kernel<DIL, false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_LDS);
    FRL_LAUNCH_AS("tcn_hot_bwd3_nodx_kernel", kern, dim3(grid), dim3(512), B3_LDS, st, (const bf16*)x, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)nullptr, slab, ntile, HW, eps);
  }
}

int th_bwd3_launch(int dilation, const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg,
                   void* dx, float* slab, unsigned grid, int64_t npix, int HW, float eps, hipStream_t st) {
  const int ntile = (int)(npix / 64);
  if (dilation == 1) b3_launch<1>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else if (dilation == 2) b3_launch<2>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else if (dilation == 4) b3_launch<4>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else return frl_fail(-2, "tcn_hot_bwd: dilation must be 1, 2 or 4");
  return 0;
}
