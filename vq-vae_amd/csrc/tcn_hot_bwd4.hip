// build-flags: -fno-slp-vectorize
// (packed-f32 vector instructions issue slower than the two scalar ones they replace beside MFMAs on gfx950: MI355X_MICROARCH.md, cycle constants)
// Backward of the hot-configuration GatedResidualBlock (frl/models/tcn.py:78-111; bf16, 64 -> 64 channels, T = 5, 8-channel GroupNorm
// groups, identity residual, dilation 1 / 2 / 4, no Dropout1d mask, HW % 64 == 0): ONE launch -> dx and all six parameter gradients.
//
// Same math, data placement and phase sequence as tcn_hot_bwd3_kernel (tcn_hot_bwd3.hip), re-cut so that the units of a SIMD stop taking
// turns: the workgroup's 8 waves are TWO independent subgroups of 4 waves (one wave per SIMD each), every subgroup walks its own
// 32-pixel tiles (wave (q, h) of a subgroup owns the 16 pixels of half q and the channel half h of every lane quarter) and synchronises
// only with itself, through a counter in LDS instead of s_barrier.  In tcn_hot_bwd3 all 8 waves sit in the same phase between two
// workgroup barriers, so the two waves of a SIMD want the matrix pipe, the vector ALU or the LDS at the same moment and each unit idles
// while another one is the bottleneck (~7.5 k cycles of each per 64 pixels, ~30 k cycles in total); here the two waves of a SIMD belong
// to different subgroups and drift apart, one streaming MFMAs while the other is in the GroupNorm / sigmoid chain.
//   * LDS: the ONE packed weight image (32 KB) and the parameter table are shared; each subgroup has its own three 20 KB tile buffers
//     X | N | A (x by LDS-DMA one tile ahead, chunk swizzle and transposing reads exactly as in tcn_hot_bwd3: the swizzle only looks at
//     the low four bits of the pixel row).
//   * weight gradients contract over the 32 pixels of the subgroup's tile = ONE 16x16x32 MFMA k-step; a wave accumulates a 32 x 32
//     block of each of the four matrices (16 accumulator tiles instead of 8) -- the two subgroups' sums are added through LDS in a
//     fixed order before the slab is written, so the slab layout and the reduction are those of tcn_hot_bwd3.
//
//   E' wait for the DMA of x, subgroup barrier   conv -> GroupNorm statistics -> n[t]            publish n[t] (own 8 channels)   | A
//   S2 gate GEMM, sigmoid, dgpre[t], dres[t] (registers), relu path of dn                        publish dgpre[t]                | B
//   S3 gate^T GEMM -> dn -> d gamma, d beta, GroupNorm backward -> dconv[t];  P2 gate weight gradient                            | C
//      DMA of the next tile's x into the N buffer, next dy -> registers;      publish dconv[t]                                   | D
//   dx = conv^T(dconv) + dres (stored);  P4 conv weight gradients
#include "tcn_hot_common.hpp"

#define B4_TT (32 * 128)                       // bytes of one time step of a tile buffer (32 pixel rows)
#define B4_TB (TH_T * B4_TT)                   // 20480: one tile buffer
#define B4_W 0                                 // packed conv taps [3][4][2][64] + gate [4][2][64] fragments
#define B4_TAB 32768                           // conv bias | gamma | beta | -log2e * gate bias
#define B4_GACC (B4_TAB + 1024)                // [4 quarters][2][64] d gamma / d beta partial sums
#define B4_BAR (B4_GACC + 2048 + 768)          // arrival counters of the two subgroups (128 bytes apart; behind the diagnostic stamps)
#define B4_TILE (B4_GACC + 2048 + 1024)        // 36864
#define B4_SG (3 * B4_TB)                      // 61440: the three tile buffers of one subgroup
#define B4_LDS (B4_TILE + 2 * B4_SG)           // 159744 <= 160 KiB
#define B4_HEADW B4_LDS                        // HEAD instantiation: 4 KB image of W_h^T behind the tile buffers (163840 = the whole LDS)
// Subgroup 0's share of the workgroup's tiles, in 32nds.  The split is static (dynamic hand-out would make the float32 summation order of
// the weight gradients depend on timing), and it is NOT half: when both waves of a SIMD are ready the sequencer issues from the older
// one, so subgroup 0 (waves 0-3) runs a tile in ~22 k cycles and subgroup 1 in ~31 k while both are busy (s_setprio on subgroup 1 does
// not change that); with 16/32 subgroup 0 finished its tiles at 76 % of the workgroup's lifetime.  19/32 lets both end together on the
// measured configuration (tools/diag/tcn_bwd3_stamps.hip: workgroup lifetime 459 k -> 425 k cycles; 18: 434 k, 20: 442 k).
#define B4_SHARE0 19

// chunk swizzle of the tile buffers: 16-byte chunk c of pixel row r is stored at chunk position c ^ b4_swz(r)
__device__ __forceinline__ int b4_swz(int r) {
  return ((r >> 1) & 1) | ((((r >> 1) ^ (r >> 2)) & 1) << 1) | (((r ^ (r >> 2) ^ (r >> 3)) & 1) << 2);
}

__device__ __forceinline__ frag8 b4_ld(const char* smem, int off) { return *reinterpret_cast<const frag8*>(smem + off); }
__device__ __forceinline__ void b4_st(char* smem, int off, const frag8& v) { *reinterpret_cast<frag8*>(smem + off) = v; }
__device__ __forceinline__ bf16x4 b4_tr4(const char* smem, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(smem + off));
}
__device__ __forceinline__ frag8 b4_join(const bf16x4& lo, const bf16x4& hi) { return frag8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }
// transposed weight fragment out of the forward image (element (lane', e') of block (m', s') of W^T): two 4 x 16 blocks 2 KB apart
__device__ __forceinline__ frag8 b4_wT(const char* smem, int a) { return b4_join(b4_tr4(smem, a), b4_tr4(smem, a + 2048)); }

// Lane constants of the address arithmetic live PACKED in three registers (pk0: own | full tile offsets; pk1: transposing-read row
// bases and swizzle bits; pk2: weight-transpose offset, global element offset, pixel, quarter).  The head of every phase takes fresh
// (opaque) copies and unpacks what it needs with one VALU operation each: rebuilt inside the tile loop like this, the address
// arithmetic costs a few instructions per phase; hoisted out of it by the compiler it pinned some thirty loop-invariant registers next
// to the accumulators and pushed the allocation over 256 registers (every spill reload sits in the same vmcnt queue as the loads).
#define B4_ADDR()                                                                          \
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
#define B4_ONES(col)                                                          \
  const bf16 one_ = px_ == (col) ? (bf16)1.f : (bf16)0.f;                    \
  const frag8 ones = frag8{one_, one_, one_, one_, one_, one_, one_, one_}
// per-channel parameters of the lane's 8 channels, re-read from the LDS table inside each phase (two ds_read_b128 each) instead of
// being carried in 24 registers across the whole tile; the opaque zero keeps the compiler from merging the reads of different phases
#define B4_PARAM(name, which)                                                 \
  float name[8];                                                              \
  {                                                                           \
    int z_ = 0;                                                               \
    asm volatile("" : "+s"(z_));                                             \
    const f32x4* p_ = reinterpret_cast<const f32x4*>(tab + (which) * 64 + 16 * kc_ + 8 * h + z_); \
    const f32x4 a_ = p_[0], b_ = p_[1];                                       \
    name[0] = a_[0]; name[1] = a_[1]; name[2] = a_[2]; name[3] = a_[3];       \
    name[4] = b_[0]; name[5] = b_[1]; name[6] = b_[2]; name[7] = b_[3];       \
  }
#define B4_TR8(base, cb) b4_join(b4_tr4(smem, (base) + tb0 + (((2 * (cb)) ^ tf0) << 4)), b4_tr4(smem, (base) + tb1 + (((2 * (cb)) ^ tf1) << 4)))
#define B4_WT(matbase, mm, s) b4_wT(smem, (matbase) + ((2 * (s)) * 2 + h) * 1024 + wtr_ + 8 * (mm))
// sum over the 16 lanes of a DPP row (= the 16 pixels of a lane quarter); every lane of the row receives the total
template <int CTRL> __device__ __forceinline__ float b4_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float b4_row_sum(float v) {
  v += b4_dpp<0xB1>(v);                                            // quad_perm [1,0,3,2]
  v += b4_dpp<0x4E>(v);                                            // quad_perm [2,3,0,1]
  v += b4_dpp<0x124>(v);                                           // row_ror:4
  v += b4_dpp<0x128>(v);                                           // row_ror:8
  return v;
}
// Barrier of one subgroup (4 waves): every wave adds 1 to the subgroup's LDS counter and waits until the counter has reached 4 x the
// number of barriers it has passed.  The LDS serves a wave's requests in issue order, so whatever a wave wrote before its add is visible
// to a wave that has seen the add.  All 8 waves of the workgroup are resident, every wave of a subgroup runs the same number of tiles
// and of barriers per tile: the wait always ends.
__device__ __forceinline__ void b4_sync(unsigned* bar, unsigned& gen, int lane) {
  gen += 4u;
  if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    const unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(seen - gen) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
}
#define B4_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); b4_sync(bar, bgen, lane); asm volatile("" ::: "memory"); } while (0)
#define B4_BARRIER_ALL() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); b4_sync(bar, bgen, lane); asm volatile("" ::: "memory"); } while (0)

// Diagnostic build only (tools/diag/tcn_bwd3_stamps.hip defines B4_STAMPS): s_memtime stamps at the phase boundaries, accumulated in the
// spare LDS behind the d gamma / d beta slots and written to a buffer nothing else reads.  The product library never defines it.
#ifdef B4_STAMPS
__device__ unsigned long long* b4_dbg;
__device__ int b4_knob[2] = {0, 16};           // wave priority experiment, subgroup 0's share of the workgroup's tiles in 32nds
#define B4_ST(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) b4_ts[wave * 12 + (i)] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define B4_ST(i) do { } while (0)
#endif

// WANT_DX = false: the block's input is data (the first block of the phase path reads the tile itself): the conv^T GEMM, the residual
// gradient and the dx store are compiled out.
// HEAD = true (the last block of the phase encoder): DY is not the block's output gradient but dh [B][5][HW][Chd], the gradient of the 1x1
// phase head's output (representation.py:169), and dy = dh W_h is formed on the matrix cores from a 4 KB image of W_h^T in LDS (Whp:
// [4 row blocks][64 lanes] fragments, contraction width padded to 32): the head's backward-data launch and the round trip of dy
// (168 MB written, 168 MB read) disappear; dy stays float32 instead of being rounded to bf16 on the way.
template <int DIL, bool WANT_DX, bool HEAD>
__global__ __launch_bounds__(512, 2) void tcn_hot_bwd4_kernel(const bf16* __restrict__ X, const bf16* __restrict__ DY, const frag8* __restrict__ Wpk,
                                                              const float* __restrict__ bc, const float* __restrict__ gn_w,
                                                              const float* __restrict__ gn_b, const float* __restrict__ bg, bf16* __restrict__ DX,
                                                              float* __restrict__ slab, int ntile, int HW, float eps,
                                                              const frag8* __restrict__ Whp, int Chd, int share0_arg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem + B4_TAB);
  float* gacc_lds = reinterpret_cast<float*>(smem + B4_GACC);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sg = wave >> 2;                                        // subgroup: waves 0-3 / 4-7 = one wave per SIMD each
  const int q = wave & 1, h = (wave >> 1) & 1;
  const int px = lane & 15, kc = lane >> 4;
  unsigned* bar = reinterpret_cast<unsigned*>(smem + B4_BAR + 128 * sg);
  unsigned bgen = 0;
  const int prow = q * 16 + px;
  const int co = 16 * kc + 8 * h;                                  // first of this lane's 8 channels
  const int64_t tstride = (int64_t)HW * 64;
  // tile-buffer offsets of this lane's pixel row: own 16-byte chunk (its 8 channels) and the two chunks of its channel quarter
  const int fsw = b4_swz(prow);
  const int own_off = prow * 128 + (((2 * kc + h) ^ fsw) << 4);
  const int full_off = prow * 128 + (((2 * kc) ^ fsw) << 4);      // fragment s sits at full_off ^ (s << 4)
  // transposing reads (k-strided fragments: 8 consecutive pixels of one channel): lane (r16 = px, kc) addresses pixel row
  // 8 kc + (r16 >> 2) + 4 hi and the 8-byte piece (r16 & 3) of the 16-channel block cb
  int trb[2], trf[2];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 8 * kc + (px >> 2) + 4 * hi;
    trb[hi] = row * 128 + 8 * (px & 1);
    trf[hi] = b4_swz(row) ^ ((px >> 1) & 1);
  }
  const int wtr = (4 * kc + (px >> 2) + 16 * (px & 3)) * 16;      // transposed weight fragments: + block * 1024 + 8 * mm
  const unsigned lane_el = (unsigned)(prow * 64 + co);             // this lane's pixel, its 8 channels (element offset inside a tile's time step)
  const unsigned pk0 = (unsigned)own_off | ((unsigned)full_off << 16);
  const unsigned pk1 = (unsigned)trb[0] | ((unsigned)trb[1] << 12) | ((unsigned)trf[0] << 24) | ((unsigned)trf[1] << 27);
  const unsigned pk2 = (unsigned)wtr | (lane_el << 10) | ((unsigned)px << 22) | ((unsigned)kc << 26);
  const unsigned tps = (unsigned)HW >> 5;                          // (32-pixel) tiles per sample
  // ---- global addressing: wave-uniform 64-bit tile base (SGPRs) + 32-bit lane offset (elements)
  auto tile_base = [&](int wt) -> int64_t {                        // element offset of (t = 0, first pixel of the tile, channel 0)
    const unsigned b = (unsigned)wt / tps;                         // sample of the tile (tiles never straddle samples: HW % 64 == 0)
    return ((int64_t)wt * 32 + (int64_t)b * (TH_T - 1) * HW) * 64;
  };
  // ---- DMA of one tile's x into a tile buffer: wave w of the subgroup issues pieces j = 5 w .. 5 w + 4 (1 KB = 8 pixel rows of one time step each).
  // Issued through inline asm: hipcc orders every later LDS access behind a builtin LDS-DMA with s_waitcnt vmcnt(0) (it cannot know that
  // the DMA targets another buffer), which parked all eight waves for a full HBM latency right after the issue.  Invisible to the
  // compiler, the transfers stay in flight across the conv^T / weight-gradient phases; the E' barrier waits for them explicitly.
  // (A hidden vector-memory operation only makes the compiler's own counted waits stricter, never looser: they count from the youngest.)
  auto dma_tile = [&](int wt, int dst) {
    const bf16* xb = X + tile_base(wt);
    unsigned k2_ = pk2;
    asm volatile("" : "+v"(k2_));                                  // (rebuilt per call: see B4_ADDR)
    const unsigned ln = ((k2_ >> 22) & 15u) + 16u * (k2_ >> 26);
    const unsigned r8 = ln >> 3, c8 = ln & 7u;
    unsigned voff[2];                                              // byte offset of this lane's 16 bytes inside a piece, pb even / odd
#pragma unroll
    for (int o = 0; o < 2; ++o) voff[o] = (r8 * 64u + ((c8 ^ (unsigned)b4_swz((int)(8u * o + r8))) * 8u)) * 2u;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int j = (wave & 3) * 5 + jj, t = j >> 2, pb = j & 3;
      const bf16* sb = xb + (int64_t)t * tstride + pb * (8 * 64);  // wave-uniform: time step and 8-row block of the piece
      const unsigned vo = (pb & 1) ? voff[1] : voff[0];
      const int ldst = dst + j * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(sb), "s"(ldst) : "memory");
    }
  };

  // ---- dy of one tile -> registers.  HEAD: the lane's slice of the dh row instead (B operand of dy = W_h^T dh: channels 8 kc .. 8 kc + 7 of
  // its pixel, zero beyond Chd; rows are Chd * 2 bytes apart, so the two 8-byte halves are loaded separately)
  auto dy_load = [&](int wt, frag8 (&dst)[TH_T]) {
    unsigned k2_ = pk2;
    asm volatile("" : "+v"(k2_));
    if constexpr (!HEAD) {
      const bf16* dyb = DY + tile_base(wt);
      const unsigned le = (k2_ >> 10) & 0xfffu;
#pragma unroll
      for (int t = 0; t < TH_T; ++t) dst[t] = *reinterpret_cast<const frag8*>(dyb + t * tstride + le);
    } else {
      const int pxl = (int)((k2_ >> 22) & 15u), kcl = (int)(k2_ >> 26);
      const unsigned b = (unsigned)wt / tps;
      const bf16* dhb = DY + ((int64_t)wt * 32 + (int64_t)b * (TH_T - 1) * HW + q * 16 + pxl) * Chd + 8 * kcl;
      const bool lo_ok = 8 * kcl < Chd, hi_ok = 8 * kcl + 4 < Chd;
      const bf16x4 z4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const bf16* p = dhb + (int64_t)t * HW * Chd;
        const bf16x4 lo = lo_ok ? *reinterpret_cast<const bf16x4*>(p) : z4;
        const bf16x4 hi = hi_ok ? *reinterpret_cast<const bf16x4*>(p + 4) : z4;
        dst[t] = b4_join(lo, hi);
      }
    }
  };
  int xoff = B4_TILE + sg * B4_SG, noff = xoff + B4_TB;
  const int aoff = xoff + 2 * B4_TB;
  // the workgroup's tiles are blockIdx.x + i gridDim.x; subgroup 0 takes the first B4_SHARE0 / 32 of them, subgroup 1 the rest
  const int wstep = (int)gridDim.x;
  const int n_wg = ((int)blockIdx.x < ntile) ? (ntile - 1 - (int)blockIdx.x) / wstep + 1 : 0;
#ifdef B4_STAMPS
  const int share0 = b4_knob[1];
  if (b4_knob[0] == 1 && sg == 1) __builtin_amdgcn_s_setprio(1);
  if (b4_knob[0] == 2 && sg == 1) __builtin_amdgcn_s_setprio(3);
  if (b4_knob[0] == 3 && sg == 0) __builtin_amdgcn_s_setprio(1);
#else
  const int share0 = share0_arg;
#endif
  const int n0 = (n_wg * share0 + 16) >> 5;
  const int i_beg = sg ? n0 : 0, i_end = sg ? n_wg : n0;
  const int wt0 = (int)blockIdx.x + i_beg * wstep, wt_end = (int)blockIdx.x + i_end * wstep;   // this subgroup's tiles: wt0, wt0 + wstep, ... < wt_end
  dma_tile(wt0 < ntile ? wt0 : ntile - 1, xoff);
  frag8 dyn[TH_T];
  dy_load(wt0 < ntile ? wt0 : ntile - 1, dyn);
  copy_frags_lds<bf16>(reinterpret_cast<frag8*>(smem + B4_W), Wpk, 32 * 64, tid, 512);
  if constexpr (HEAD) copy_frags_lds<bf16>(reinterpret_cast<frag8*>(smem + B4_HEADW), Whp, 4 * 64, tid, 512);
  gacc_lds[tid] = 0.f;                                             // [8 waves][4 kc][8 d gamma | 8 d beta]
  if (tid < 2) *reinterpret_cast<unsigned*>(smem + B4_BAR + 128 * tid) = 0u;
  if (tid < 64) {
    tab[tid] = bc[tid];
    tab[64 + tid] = gn_w[tid];
    tab[128 + tid] = gn_b[tid];
    tab[192 + tid] = -1.44269504088896f * bg[tid];
  }

#ifdef B4_STAMPS
  unsigned long long* b4_ts = reinterpret_cast<unsigned long long*>(smem + B4_GACC + 2048);   // [8 waves][12]
  if (tid < 96) b4_ts[tid] = 0ull;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_start = t_prev, r_start = __builtin_amdgcn_s_memrealtime();
#endif
  // weight-gradient accumulators: rows 16 (2q + j).., columns 16 (2h + i).. of the three conv taps and the gate matrix; bias sums of
  // row block 2q + h (column 0: conv bias, column 1: gate bias)
  f32x4 accC[3][2][2], accG[2][2], accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k) accC[k][j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      accG[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  __syncthreads();                                                 // weight image, table, counters (the only workgroup barrier before the end)

  for (int wt = wt0; wt < wt_end; wt += wstep) {
    bf16* dxb = DX + tile_base(wt);
    frag8 dyo[TH_T];
#pragma unroll
    for (int t = 0; t < TH_T; ++t) dyo[t] = dyn[t];
    B4_ST(11);
    B4_BARRIER_ALL();                                              // E': x of this tile has landed, everyone left the previous tile
    B4_ST(0);
    // ---------------- conv(x) + bias of this lane's 8 channels, all time steps ----------------
    f32x4 xh[TH_T][2];
    {
      B4_ADDR();
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
        for (int t = 0; t < TH_T; ++t) xs[t] = b4_ld(smem, xoff + t * B4_TT + (fo ^ (s << 4)));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const frag8 wf = b4_ld(smem, B4_W + wfwd_ + ((k * 4 + mm) * 2 + s) * 1024);
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
      B4_ADDR();
      B4_PARAM(gw, 1);
      B4_PARAM(gb, 2);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float n[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaf(xh[t][e >> 2][e & 3], rstd, nm);
          xh[t][e >> 2][e & 3] = v;
          n[e] = fmaf(v, gw[e], gb[e]);
        }
        b4_st(smem, noff + t * B4_TT + oo, th_pack8(n));
      }
    }
    B4_ST(1);
    B4_BARRIER();                                                  // A: n[t] complete (all channels of every pixel)
    B4_ST(2);
    // ---------------- S2: gate, dgpre, dy g (kept packed: dres = dy - dy g and the relu path of dn = [n > 0] dy g are rebuilt from it) ----------------
    frag8 dyg8[TH_T];
    {
      B4_ADDR();
      B4_PARAM(gw, 1);
      B4_PARAM(gb, 2);
      B4_PARAM(tnbg, 3);
      frag8 wg[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int s = 0; s < 2; ++s) wg[mm][s] = b4_ld(smem, B4_W + wfwd_ + ((12 + mm) * 2 + s) * 1024);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const frag8 nt0 = b4_ld(smem, noff + t * B4_TT + fo), nt1 = b4_ld(smem, noff + t * B4_TT + (fo ^ 16));
        const frag8 xo = b4_ld(smem, xoff + t * B4_TT + oo);   // residual path: x of this lane's 8 channels
        f32x4 gacc[2], dyf[2];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          gacc[mm] = mfma16(wg[mm][0], nt0, f32x4{0.f, 0.f, 0.f, 0.f});
          gacc[mm] = mfma16(wg[mm][1], nt1, gacc[mm]);
          if constexpr (HEAD) dyf[mm] = mfma16(b4_ld(smem, B4_HEADW + ((2 * h + mm) * 64) * 16 + (px_ + 16 * kc_) * 16), dyo[t], f32x4{0.f, 0.f, 0.f, 0.f});
        }
        (void)dyf;
        float dgp[8], dygv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float n = fmaf(xh[t][e >> 2][e & 3], gw[e], gb[e]);
          const float dyv = HEAD ? dyf[e >> 2][e & 3] : (float)dyo[t][e];
          const float g = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(gacc[e >> 2][e & 3], -1.44269504088896f, tnbg[e])));
          const float o = fmaxf(n, 0.f);
          const float res = (float)xo[e];
          const float dyg = dyv * g;
          dygv[e] = dyg;
          dgp[e] = (o - res) * (dyg - dyg * g);                    // dy (o - res) g (1 - g)
        }
        b4_st(smem, aoff + t * B4_TT + oo, th_pack8(dgp));
        dyg8[t] = th_pack8(dygv);
      }
    }
    B4_ST(3);
    B4_BARRIER();                                                  // B: dgpre[t] complete
    B4_ST(4);
    // ---------------- S3: gate^T, dn, GroupNorm backward ----------------
    frag8 dcf[TH_T];
    auto phase_s3 = [&]() {
      B4_ADDR();
      B4_PARAM(gw, 1);
      B4_PARAM(gb, 2);
      frag8 wgT[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int s = 0; s < 2; ++s) wgT[mm][s] = B4_WT(B4_W + 24 * 1024, mm, s);
      float S1p[4] = {0.f, 0.f, 0.f, 0.f}, S2p[4] = {0.f, 0.f, 0.f, 0.f};
      float dgam[8], dbet[8];                                      // this tile's d gamma / d beta of the lane's 8 channels
#pragma unroll
      for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
      frag8 dxh[TH_T];
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        const frag8 gt0 = b4_ld(smem, aoff + t * B4_TT + fo), gt1 = b4_ld(smem, aoff + t * B4_TT + (fo ^ 16));
        float dd[8];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          f32x4 bacc;                                              // relu path: [n > 0] dy g
#pragma unroll
          for (int r = 0; r < 4; ++r) bacc[r] = fmaf(xh[t][mm][r], gw[4 * mm + r], gb[4 * mm + r]) > 0.f ? (float)dyg8[t][4 * mm + r] : 0.f;
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
      for (int e = 0; e < 8; ++e) { dgam[e] = b4_row_sum(dgam[e]); dbet[e] = b4_row_sum(dbet[e]); }
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
    // ---------------- P2: gate weight gradient (rows 32q.., columns 32h..; one k-step = the tile's 32 pixels) ----------------
    auto phase_p2 = [&]() {
      B4_ADDR();
      B4_ONES(1);
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        frag8 af[2], bf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) af[j] = B4_TR8(aoff + t * B4_TT, 2 * q + j);
#pragma unroll
        for (int i = 0; i < 2; ++i) bf[i] = B4_TR8(noff + t * B4_TT, 2 * h + i);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) accG[j][i] = mfma16(af[j], bf[i], accG[j][i]);
        accB = mfma16(h ? af[1] : af[0], ones, accB);
      }
    };
    // Both need only what barrier B published.  P2 first: its transposing reads and MFMAs are issued while dn0 / xh are merely parked,
    // and the allocation closes without a spill in this order.  (Letting the two waves of a SIMD take the phases in opposite order --
    // one in the vector-ALU chain while its partner streams LDS reads into MFMAs -- costs 50 spilled registers at the join of the two
    // code paths, as a branch and as a two-trip loop alike.)
    phase_p2();
    B4_ST(5);
    phase_s3();
    B4_ST(6);
    B4_BARRIER();                                                  // C: everyone is done with n[t], dgpre[t]
    B4_ST(7);
    // ---------------- the next tile's x -> N buffer (LDS-DMA), its dy -> registers; publish dconv[t] ----------------
    const int wtn = wt + wstep;
    if (wtn < wt_end) dma_tile(wtn, noff);
    dy_load(wtn < wt_end ? wtn : wt, dyn);                           // (unconditional, from a clamped tile: dyn must not stay live across the tile)
    {
      B4_ADDR();
#pragma unroll
      for (int t = 0; t < TH_T; ++t) b4_st(smem, aoff + t * B4_TT + oo, dcf[t]);
    }
    B4_ST(8);
    B4_BARRIER();                                                  // D (LDS only: the DMA and the dy loads stay in flight)
    B4_ST(9);
    // ---------------- dx[t'] = sum_k W_k^T dconv[t' - (k-1) d] + dres[t'] (this wave's 8 channels per lane) ----------------
    if constexpr (WANT_DX) {
      B4_ADDR();
      // dres = dy - dy g: dy of this tile is fetched a second time (an L2 hit: the same compute unit read it for S2) instead of being
      // carried in 20 registers from S2 to here; the loads land behind the conv^T GEMM
      frag8 dyr[TH_T];
      dy_load(wt, dyr);
      f32x4 dxa[TH_T][2];
#pragma unroll
      for (int t = 0; t < TH_T; ++t)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) dxa[t][mm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag8 dcs[TH_T];
#pragma unroll
        for (int t = 0; t < TH_T; ++t) dcs[t] = b4_ld(smem, aoff + t * B4_TT + (fo ^ (s << 4)));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const frag8 wf = B4_WT(B4_W + k * 8 * 1024, mm, s);
#pragma unroll
            for (int tp = 0; tp < TH_T; ++tp)
              if (th_valid<DIL>(tp, 2 - k)) dxa[tp][mm] = mfma16(wf, dcs[tp - (k - 1) * DIL], dxa[tp][mm]);
          }
      }
#pragma unroll
      for (int t = 0; t < TH_T; ++t) {
        float y[8];
        if constexpr (HEAD) {                                        // dy = W_h^T dh once more (two MFMAs) instead of a second read of 128 bytes
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) dxa[t][mm] = mfma16(b4_ld(smem, B4_HEADW + ((2 * h + mm) * 64) * 16 + (px_ + 16 * kc_) * 16), dyr[t], dxa[t][mm]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = dxa[t][e >> 2][e & 3] + ((HEAD ? 0.f : (float)dyr[t][e]) - (float)dyg8[t][e]);
        *reinterpret_cast<frag8*>(dxb + t * tstride + le_) = th_pack8(y);
      }
    }
    B4_ST(10);
    // ---------------- P4: conv weight gradients  dW_k += dconv[tp - (k-1) d]^T x[tp] ----------------
    {
      B4_ADDR();
      B4_ONES(0);
#pragma unroll
      for (int tp = 0; tp < TH_T; ++tp) {
        frag8 bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) bf[i] = B4_TR8(xoff + tp * B4_TT, 2 * h + i);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (!th_valid<DIL>(tp, 2 - k)) continue;
          frag8 af[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) af[j] = B4_TR8(aoff + (tp - (k - 1) * DIL) * B4_TT, 2 * q + j);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) accC[k][j][i] = mfma16(af[j], bf[i], accC[k][j][i]);
          if (k == 1) accB = mfma16(h ? af[1] : af[0], ones, accB);
        }
      }
    }
    const int tmp = xoff; xoff = noff; noff = tmp;                 // the DMA target becomes X, the old X buffer receives the next n[t]
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the DMA of a clamped tile may still be in flight)
  __syncthreads();                                                 // both subgroups have left their last tile
#ifdef B4_STAMPS
  if (tid < 96) b4_dbg[(size_t)blockIdx.x * 96 + tid] = b4_ts[tid];
  if (tid == 0) {                                                  // shader-clock cycles and 100 MHz ticks of the whole tile loop
    b4_dbg[(size_t)gridDim.x * 96 + 2048 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
    b4_dbg[(size_t)gridDim.x * 96 + 2048 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r_start;
  }
  if (lane == 0) b4_dbg[(size_t)gridDim.x * 96 + blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_ID.simd_id
#endif
  // ---------------- subgroup 1 parks its sums in the free tile buffers, subgroup 0 adds them (fixed order) and writes the slab ----------------
  B4_ADDR();
  float* xch = reinterpret_cast<float*>(smem + B4_TILE) + (wave & 3) * 64 + lane;     // [68][256]
  if (sg == 1) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) xch[256 * n++] = accC[k][j][i][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[256 * n++] = accG[j][i][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) xch[256 * n++] = accB[r];
  }
  __syncthreads();
  float* my = slab + (int64_t)blockIdx.x * TH_SLAB;
  if (sg == 0) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = (2 * h + i) * 16 + px_;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) my[(k * 64 + (2 * q + j) * 16 + kc_ * 4 + r) * 64 + col] = accC[k][j][i][r] + xch[256 * n++];
#pragma unroll
        for (int r = 0; r < 4; ++r) my[(3 * 64 + (2 * q + j) * 16 + kc_ * 4 + r) * 64 + col] = accG[j][i][r] + xch[256 * n++];
      }
    if (px_ < 2) {                                                 // column 0: d conv bias, column 1: d gate bias (row block 2q + h)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[4 * 64 * 64 + 64 * px_ + (2 * q + h) * 16 + kc_ * 4 + r] = accB[r] + xch[256 * (n + r)];
    }
  }
  for (int i = tid; i < 2 * 64; i += 512) {                        // d gamma / d beta: the four waves (subgroup, q) that own channel half h of quarter kc
    const int which = i >> 6, c = i & 63, hh = (c >> 3) & 1;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += gacc_lds[(((w >> 1) * 4 + 2 * hh + (w & 1)) * 4 + (c >> 4)) * 16 + 8 * which + (c & 7)];
    my[4 * 64 * 64 + 128 + i] = s;
  }
}

bool th_bwd4_supported(int64_t npix, int HW) { return HW > 0 && HW % 64 == 0 && npix % HW == 0 && npix / 32 < ((int64_t)1 << 30); }

// subgroup 0's share of the tiles (32nds) per variant: with dx, without dx, head; test / tuning hook frl_tcn_hot_bwd4_share
static int g_b4_share[3] = {B4_SHARE0, B4_SHARE0 + 1, B4_SHARE0};   // (tools/diag/bwd4_share_ab.py: 19 / 20 / 19 of 32 are the minima: 223.5, 173.1, 202.4 us per call)
extern "C" int frl_tcn_hot_bwd4_share(int variant, int share) {
  if (variant < 0 || variant > 2) return -1;
  const int was = g_b4_share[variant];
  if (share >= 1 && share <= 31) g_b4_share[variant] = share;
  return was;
}

template <int DIL>
static void b4_launch(const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg, void* dx,
                      float* slab, unsigned grid, int ntile, int HW, float eps, hipStream_t st) {
  if (dx != nullptr) {
    auto kern = tcn_hot_bwd4_kernel<DIL, true, false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS);
    FRL_LAUNCH_AS("tcn_hot_bwd4_kernel", kern, dim3(grid), dim3(512), B4_LDS, st, (const bf16*)x, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)dx, slab, ntile, HW, eps,
                  (const frag8*)nullptr, 0, g_b4_share[0]);
  } else {
    auto kern = tcn_hot_bwd4_kernel<DIL, false, false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4_LDS);
    FRL_LAUNCH_AS("tcn_hot_bwd4_nodx_kernel", kern, dim3(grid), dim3(512), B4_LDS, st, (const bf16*)x, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)nullptr, slab, ntile, HW, eps,
                  (const frag8*)nullptr, 0, g_b4_share[1]);
  }
}

// whp != NULL: `dy` is dh [B][5][HW][chd], the output gradient of the 1x1 head behind this block (dilation 4, dx wanted): tcn_hot_bwd4_kernel<4, true, true>
int th_bwd4_launch(int dilation, const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg,
                   void* dx, float* slab, unsigned grid, int64_t npix, int HW, float eps, hipStream_t st, const frag8* whp, int chd) {
  const int ntile = (int)(npix / 32);
  if (whp != nullptr) {
    if (dilation != 4 || dx == nullptr) return frl_fail(-2, "tcn_hot_bwd_head: the head variant serves the last block (dilation 4, dx wanted)");
    auto kern = tcn_hot_bwd4_kernel<4, true, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B4_LDS + 4096));
    FRL_LAUNCH_AS("tcn_hot_bwd4_head_kernel", kern, dim3(grid), dim3(512), B4_LDS + 4096, st, (const bf16*)x, (const bf16*)dy, pk, bc, gw, gb, bg, (bf16*)dx, slab, ntile, HW,
                  eps, whp, chd, g_b4_share[2]);
    return 0;
  }
  if (dilation == 1) b4_launch<1>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else if (dilation == 2) b4_launch<2>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else if (dilation == 4) b4_launch<4>(x, dy, pk, bc, gw, gb, bg, dx, slab, grid, ntile, HW, eps, st);
  else return frl_fail(-2, "tcn_hot_bwd: dilation must be 1, 2 or 4");
  return 0;
}
