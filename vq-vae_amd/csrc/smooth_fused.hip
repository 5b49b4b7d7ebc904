// build-flags: -fno-slp-vectorize
// (packed-f32 vector instructions issue slower than the two scalar ones they replace beside MFMAs on gfx950: MI355X_MICROARCH.md, cycle constants)
// Fused mixing heads + softmaxes + directional filter bank of EdgeAwareSmoothingConv2D for the hot configuration
// (bf16, 64 channels, 64 hidden features, 4 directions x 2 scales, rank 4):
//   A   = softmax_k(W_A feat + b_A)   viewed [K = 8][R = 4]          frl/models/spatial.py:262,300-302   (K7)
//   B   = softmax_r(W_B feat + b_B)   viewed [C = 64][R = 4]         spatial.py:263,305-307              (K8)
//   smoothed[c] = sum_k (sum_r A[k][r] B[c][r]) * f_k[c],  f_k = 3-tap line average of x along direction k / scale (K9, K10)
//   residual = x - smoothed                                           spatial.py:314-331
// Forward (smooth_heads_fwd_kernel): ONE pass over (feat, x): both 1x1 heads run on the matrix cores with the wave's 16-pixel
//   lane-quarter image as the B operand, the [P,32] + [P,256] logits, their soft-maxed copies and the 8 filtered copies of x live in
//   registers only (the modular chain writes and re-reads ~0.6 GB of them per step at cfg2).  The 32 direction weights of a pixel are
//   soft-maxed by its four lanes together (quarter max / sum by v_permlane16_swap / v_permlane32_swap) and parked in a per-wave LDS
//   row that all four lanes read back filter by filter (one ds_read_b128 per filter): 32 registers less per lane.
// Backward (two launches):
//   smooth_heads_bwd_kernel  recomputes the heads from `feat`, derives d logits in registers, and produces d feat (W^T on the matrix
//     cores), the four head parameter gradients (pixel contraction through LDS tiles + ds_read_b64_tr_b16, per-workgroup slabs) and
//     the two small exchange tensors of the transposed stencil: u[p][c][r] = ds[p][c] B[p][c][r] and the soft-maxed A (bf16);
//   smooth_dx_kernel         dx[p][c] = ds[p][c] / 3 + 1/3 sum over the 16 star neighbours q of sum_r A[q][k(q->p)][r] u[q][c][r] (+ add)
//     (the centre taps of the 8 filters collapse to ds / 3 because sum_k A = sum_r B = 1).
//   dA / dB [P,288], their bwd-data and weight-gradient launches and the b_soft round trip of the modular chain do not exist.
// Roofline: HBM by bytes (fwd 4 x 128 B per pixel); vector-ALU bound in practice (two softmaxes + 8 x 64 rank contractions per pixel).
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include "frl_reduce.hpp"

typedef bf16x8 frag8;
#define SH_C 64            // channels of x
#define SH_HID 64          // channels of feat
#define SH_K 8             // filters: 4 directions x {fine, coarse}
#define SH_R 4             // rank
#define SH_NA (SH_K * SH_R)     // 32 head-A logits
#define SH_NB (SH_C * SH_R)     // 256 head-B logits
#define SH_LOG2E 1.44269504088896f
#define SH_PA 36           // pitch (floats) of a pixel's soft-maxed A row in LDS (32 used; 144 bytes keeps the transposing reads of the aliased tile spread)

// fragment counts of the packed weight images
#define SH_FR_B (16 * 2 * 64)   // W_B  forward:    MB = 16, NF = 2
#define SH_FR_A (2 * 2 * 64)    // W_A  forward:    MB = 2,  NF = 2
#define SH_FR_BT (4 * 8 * 64)   // W_B^T: d feat <- d logit B, MB = 4, NF = 8
#define SH_FR_AT (4 * 1 * 64)   // W_A^T: d feat <- d logit A, MB = 4, NF = 1

__global__ void sh_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ WA, const float* __restrict__ WB, int bwd) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  pack_weights_lds<bf16, 2>(dst, WB, SH_NB, SH_HID, 16, SH_HID, 1, tid, nt);                       // logit B[o] = sum_i W_B[o][i] feat[i]
  pack_weights_lds<bf16, 2>(dst + SH_FR_B, WA, SH_NA, SH_HID, 2, SH_HID, 1, tid, nt);
  if (!bwd) return;
  pack_weights_lds<bf16, 8>(dst + SH_FR_B + SH_FR_A, WB, SH_HID, SH_NB, 4, 1, SH_HID, tid, nt);    // d feat[o] = sum_l W_B[l][o] d logit B[l]
  pack_weights_lds<bf16, 1>(dst + SH_FR_B + SH_FR_A + SH_FR_BT, WA, SH_HID, SH_NA, 4, 1, SH_HID, tid, nt);
}

static const frag8* sh_packed(const float* wa, const float* wb, int bwd, frag8* ws_pk, hipStream_t st) {
  FrlPackJob jobs[4];
  size_t off = 0;
  jobs[0] = frl_pack_job_pw(wb, off, FRL_BF16, 2, SH_NB, SH_HID, 16, SH_HID, 1);
  off += (size_t)SH_FR_B * sizeof(frag8);
  jobs[1] = frl_pack_job_pw(wa, off, FRL_BF16, 2, SH_NA, SH_HID, 2, SH_HID, 1);
  off += (size_t)SH_FR_A * sizeof(frag8);
  int n = 2;
  if (bwd) {
    jobs[2] = frl_pack_job_pw(wb, off, FRL_BF16, 8, SH_HID, SH_NB, 4, 1, SH_HID);
    off += (size_t)SH_FR_BT * sizeof(frag8);
    jobs[3] = frl_pack_job_pw(wa, off, FRL_BF16, 1, SH_HID, SH_NA, 4, 1, SH_HID);
    off += (size_t)SH_FR_AT * sizeof(frag8);
    n = 4;
  }
  bool hit = false;
  frag8* pk = ws_pk;
  if (void* img = frl_pack_cached(jobs, n, off, &hit)) pk = (frag8*)img;
  if (!hit) FRL_LAUNCH(sh_pack_kernel, dim3(32), dim3(256), 0, st, pk, wa, wb, bwd);
  return pk;
}

// direction of filter k (templates of spatial.py:224-229: E-W, N-S, diagonal, anti-diagonal) and its dilation
__device__ __forceinline__ void sh_dir(int k, int dil, int& dy, int& dx) {
  const int i = k >> 1, d = (k & 1) ? dil : 1;
  dy = (i == 0) ? 0 : d;
  dx = (i == 0) ? d : (i == 1) ? 0 : (i == 2) ? d : -d;
}

// sum / max over the four lane quarters (lanes l, l^16, l^32, l^48) without LDS: v_permlane16_swap / v_permlane32_swap exchange the
// odd rows (upper half) of one register with the even rows (lower half) of another; fed two copies of v, the two results hold v of the
// lane's own row and of its partner row.  (__builtin_bit_cast applied directly to an element of the builtin's result vector reads
// element 0 for both -- hipcc, ROCm 7.2 -- hence the scalar copies.)
__device__ __forceinline__ void sh_swap16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void sh_swap32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float sh_quarter_sum(float v) {
  float a, b;
  sh_swap16(v, a, b);
  sh_swap32(a + b, a, b);
  return a + b;
}
__device__ __forceinline__ float sh_quarter_max(float v) {
  float a, b;
  sh_swap16(v, a, b);
  sh_swap32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}

// Head A of one 16-pixel tile: lane quarter kc holds the logits of filters k = 2 kc, 2 kc + 1 (4 rank slots each); softmax over the 8
// filters per rank slot (spatial.py:300-302) across the pixel's four lanes; the soft-maxed row is written to arow = this pixel's
// LDS row [k][r] (SH_PA floats), from which every lane of the pixel reads all 32 values.
__device__ __forceinline__ void sh_head_a(float* __restrict__ arow, const LQTile<bf16, 2>& ft, const frag8* __restrict__ wA,
                                          const float* __restrict__ tb, int lane, int kc) {
  f32x4 a[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(tb + SH_NB + 8 * kc + 4 * m);
    acc = mfma16(wA[(m * 2 + 0) * 64 + lane], ft.f[0], acc);
    a[m] = mfma16(wA[(m * 2 + 1) * 64 + lane], ft.f[1], acc);
  }
#pragma unroll
  for (int r = 0; r < SH_R; ++r) {
    const float ms = sh_quarter_max(fmaxf(a[0][r], a[1][r])) * SH_LOG2E;
    a[0][r] = __builtin_amdgcn_exp2f(fmaf(a[0][r], SH_LOG2E, -ms));
    a[1][r] = __builtin_amdgcn_exp2f(fmaf(a[1][r], SH_LOG2E, -ms));
    const float inv = __builtin_amdgcn_rcpf(sh_quarter_sum(a[0][r] + a[1][r]));
    a[0][r] *= inv;
    a[1][r] *= inv;
  }
  *reinterpret_cast<f32x4*>(arow + 8 * kc) = a[0];
  *reinterpret_cast<f32x4*>(arow + 8 * kc + 4) = a[1];
}

// head B for the channels 16 * kc + 8 * h + e (e < 8) of the lane: 8 accumulator tiles -> soft-maxed Bw[e][r] (spatial.py:305-307)
__device__ __forceinline__ void sh_head_b_half(f32x4 (&Bw)[8], const LQTile<bf16, 2>& ft, const frag8* __restrict__ wB,
                                               const float* __restrict__ tb, int lane, int kc, int h) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int m = 8 * h + e;
    f32x4 acc = *reinterpret_cast<const f32x4*>(tb + 64 * kc + 4 * m);
    acc = mfma16(wB[(m * 2 + 0) * 64 + lane], ft.f[0], acc);
    acc = mfma16(wB[(m * 2 + 1) * 64 + lane], ft.f[1], acc);
    const float mx = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])) * SH_LOG2E;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < SH_R; ++r) { acc[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], SH_LOG2E, -mx)); sum += acc[r]; }
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int r = 0; r < SH_R; ++r) acc[r] *= inv;
    Bw[e] = acc;
  }
}

// the two outer taps of filter k at pixel (x, y): 8 channels each (zero padding: weight 0 on a safe address)
struct ShTaps { bf16x8 a, b; float wa, wb; };
__device__ __forceinline__ ShTaps sh_taps(const bf16* __restrict__ xp8, int k, int dil, int x, int y, int H, int W) {
  int dy, dx;
  sh_dir(k, dil, dy, dx);
  const int off = (dy * W + dx) * SH_C;
  const bool oka = (unsigned)(y - dy) < (unsigned)H && (unsigned)(x - dx) < (unsigned)W;
  const bool okb = (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
  ShTaps t;
  t.a = *reinterpret_cast<const bf16x8*>(oka ? xp8 - off : xp8);
  t.b = *reinterpret_cast<const bf16x8*>(okb ? xp8 + off : xp8);
  t.wa = oka ? 1.f : 0.f;
  t.wb = okb ? 1.f : 0.f;
  return t;
}

__device__ __forceinline__ void sh_unpack8(const bf16x8 v, float* o) {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 3) void smooth_heads_fwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ FEAT,
                                                                  const frag8* __restrict__ Wpk, const float* __restrict__ bA,
                                                                  const float* __restrict__ bB, bf16* __restrict__ SM, bf16* __restrict__ RES,
                                                                  int B, int H, int W, int dil) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wB = reinterpret_cast<frag8*>(smem);
  frag8* wA = wB + SH_FR_B;
  float* tb = reinterpret_cast<float*>(wA + SH_FR_A);         // b_B[256] | b_A[32]
  float* t_a = tb + SH_NB + SH_NA;                            // [64 pixels][SH_PA] soft-maxed A rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<bf16>(wB, Wpk, SH_FR_B + SH_FR_A, tid, 256);
  for (int i = tid; i < SH_NB + SH_NA; i += 256) tb[i] = i < SH_NB ? bB[i] : bA[i - SH_NB];
  __syncthreads();
  float* arow = t_a + (wave * 16 + px) * SH_PA;
  const int npix = B * H * W;
  const int ntile = (npix + 15) >> 4;
  const float third = 1.f / 3.f;
  for (int tile = (int)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave; tile < ntile; tile += (int)gridDim.x * 4) {
    const int p = tile * 16 + px;
    const bool inb = p < npix;
    const int pc = inb ? p : npix - 1;
    const int x = pc % W, y = (pc / W) % H;
    LQTile<bf16, 2> ft;
    lq_load<bf16, 2>(ft, FEAT, pc, SH_HID, kc, true);
    const bf16* xp = X + (int64_t)pc * SH_C + 16 * kc;
    int lw = lane;
    asm volatile("" : "+v"(lw));                               // (opaque per tile: the weight fragments stay in LDS instead of hoisted registers)
    sh_head_a(arow, ft, wA, tb, lw, kc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 Bw[8];
      sh_head_b_half(Bw, ft, wB, tb, lw, kc, h);
      float ctr[8], sm[8];
      sh_unpack8(*reinterpret_cast<const bf16x8*>(xp + 8 * h), ctr);
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[e] = 0.f;
      ShTaps nx = sh_taps(xp + 8 * h, 0, dil, x, y, H, W);
#pragma unroll 1
      for (int k = 0; k < SH_K; ++k) {
        const ShTaps cu = nx;
        nx = sh_taps(xp + 8 * h, k < SH_K - 1 ? k + 1 : k, dil, x, y, H, W);       // next filter's taps fly behind this filter's arithmetic
        const f32x4 Ak = *reinterpret_cast<const f32x4*>(arow + 4 * k);
        float f[8], va[8], vb[8];
        sh_unpack8(cu.a, va);
        sh_unpack8(cu.b, vb);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaf(cu.wb, vb[e], fmaf(cu.wa, va[e], ctr[e]));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float w = Ak[0] * Bw[e][0];
          w = fmaf(Ak[1], Bw[e][1], w);
          w = fmaf(Ak[2], Bw[e][2], w);
          w = fmaf(Ak[3], Bw[e][3], w);
          sm[e] = fmaf(w, f[e], sm[e]);
        }
      }
      if (inb) {
        bf16x8 so, ro;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = third * sm[e];
          so[e] = (bf16)a;
          ro[e] = (bf16)(ctr[e] - a);
        }
        *reinterpret_cast<bf16x8*>(SM + (int64_t)p * SH_C + 16 * kc + 8 * h) = so;
        *reinterpret_cast<bf16x8*>(RES + (int64_t)p * SH_C + 16 * kc + 8 * h) = ro;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, kernel 1
// 8 waves x 16 pixels per round.  slab (floats): dW_B [256][64] | dW_A [32][64] | db_B [256] | db_A [32]
#define SHB_NW 8
#define SHB_R (16 * SHB_NW)
#define SHB_PLB (128 + 8)       // pitch of the d-logit-B half tile (bf16 elements; 16-byte skew)
#define SHB_PLA (2 * SH_PA)     // the d-logit-A tile aliases the pixel's soft-maxed A row (same bytes, bf16 view)
#define SHB_PF (64 + 8)
#define SHB_SLAB (SH_NB * SH_HID + SH_NA * SH_HID + SH_NB + SH_NA)

__device__ __forceinline__ bf16x8 sh_tr_frag(const bf16* tile, int pitch, int pix0, int ch0, int r16) {
  const bf16* a0 = tile + (pix0 + (r16 >> 2)) * pitch + ch0 + 4 * (r16 & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitch));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(64 * SHB_NW, 2) void smooth_heads_bwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ FEAT,
                                                                         const bf16* __restrict__ DS, const frag8* __restrict__ Wpk,
                                                                         const float* __restrict__ bA, const float* __restrict__ bB,
                                                                         bf16* __restrict__ DFEAT, bf16* __restrict__ U, bf16* __restrict__ AS,
                                                                         float* __restrict__ slab, int B, int H, int W, int dil, int dfeat_relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wB = reinterpret_cast<frag8*>(smem);
  frag8* wA = wB + SH_FR_B;
  frag8* wBt = wA + SH_FR_A;
  frag8* wAt = wBt + SH_FR_BT;
  float* tb = reinterpret_cast<float*>(wAt + SH_FR_AT);       // b_B[256] | b_A[32]
  float* t_a = tb + SH_NB + SH_NA;                            // [R][SH_PA] f32: soft-maxed A rows; re-used as the d-logit-A tile (bf16)
  bf16* t_dla = reinterpret_cast<bf16*>(t_a);                 // [R][SHB_PLA]
  bf16* t_dlb = reinterpret_cast<bf16*>(t_a + SHB_R * SH_PA); // [R][PLB]  one half (32 logits per lane quarter) of d logit B
  bf16* t_ft = t_dlb + SHB_R * SHB_PLB;                       // [R][PF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4, r16 = px;
  const int prow = wave * 16 + px;
  copy_frags_lds<bf16>(wB, Wpk, SH_FR_B + SH_FR_A + SH_FR_BT + SH_FR_AT, tid, 64 * SHB_NW);
  for (int i = tid; i < SH_NB + SH_NA; i += 64 * SHB_NW) tb[i] = i < SH_NB ? bB[i] : bA[i - SH_NB];
  __syncthreads();
  float* arow = t_a + prow * SH_PA;
  const int npix = B * H * W;
  const int nround = (npix + SHB_R - 1) / SHB_R;
  const float third = 1.f / 3.f;
  const bf16 one = (bf16)1.f, zero = (bf16)0.f;
  const bf16x8 ones = (r16 == 0) ? bf16x8{one, one, one, one, one, one, one, one} : bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};
  const bf16x8 zeros = bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};
  // weight-gradient ownership: d W_B rows 64 * (wave / 2) + 16 * (wave % 2) + 32 * pass + i  (the tile's column block `wave`);
  // d W_A block (row block wave / 4, column block wave % 4)
  f32x4 gW[2][4], gb[2], gWa, gba;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    gb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gW[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  gWa = f32x4{0.f, 0.f, 0.f, 0.f};
  gba = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int rd = (int)xcd_remap(blockIdx.x, gridDim.x); rd < nround; rd += (int)gridDim.x) {
    const int p = rd * SHB_R + prow;
    const bool inb = p < npix;
    const int pc = inb ? p : npix - 1;
    const int x = pc % W, y = (pc / W) % H;
    const bf16* xp = X + (int64_t)pc * SH_C + 16 * kc;
    const bf16* dsp = DS + (int64_t)pc * SH_C + 16 * kc;
    int lw = lane;
    asm volatile("" : "+v"(lw));
    f32x4 dA0 = f32x4{0.f, 0.f, 0.f, 0.f}, dA1 = dA0, dfa[4];   // d A[k][r] of the two filters this lane quarter owns (k = 2 kc, 2 kc + 1)
    {
      LQTile<bf16, 2> ft;
      lq_load<bf16, 2>(ft, FEAT, pc, SH_HID, kc, true);
      sh_head_a(arow, ft, wA, tb, lw, kc);
      bf16x8* fp = reinterpret_cast<bf16x8*>(t_ft + prow * SHB_PF + 16 * kc);      // feat for the pixel contraction (zero rows past the end)
      fp[0] = inb ? ft.f[0] : zeros;
      fp[1] = inb ? ft.f[1] : zeros;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) dfa[m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 Bw[8];
      {
        LQTile<bf16, 2> ft;                                     // (re-read per half: an L1 hit instead of 8 registers held across the half)
        lq_load<bf16, 2>(ft, FEAT, pc, SH_HID, kc, true);
        sh_head_b_half(Bw, ft, wB, tb, lw, kc, h);
      }
      float ds[8], ctr[8];
      sh_unpack8(*reinterpret_cast<const bf16x8*>(dsp + 8 * h), ds);
      sh_unpack8(*reinterpret_cast<const bf16x8*>(xp + 8 * h), ctr);
      if (!inb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ds[e] = 0.f;
      }
      f32x4 slot[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) slot[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      ShTaps nx = sh_taps(xp + 8 * h, 0, dil, x, y, H, W);
#pragma unroll 1
      for (int k = 0; k < SH_K; ++k) {
        const ShTaps cu = nx;
        nx = sh_taps(xp + 8 * h, k < SH_K - 1 ? k + 1 : k, dil, x, y, H, W);
        const f32x4 Ak = *reinterpret_cast<const f32x4*>(arow + 4 * k);
        float f[8], va[8], vb[8];
        sh_unpack8(cu.a, va);
        sh_unpack8(cu.b, vb);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaf(cu.wb, vb[e], fmaf(cu.wa, va[e], ctr[e]));
        f32x4 dAk = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = ds[e] * f[e];
#pragma unroll
          for (int r = 0; r < SH_R; ++r) {
            slot[e][r] = fmaf(Ak[r], f[e], slot[e][r]);
            dAk[r] = fmaf(g, Bw[e][r], dAk[r]);
          }
        }
        // the pixel's four lanes add up their channel quarters; the lane quarter that owns filter k (k / 2 == kc) keeps the sum
        const bool own0 = k == 2 * kc, own1 = k == 2 * kc + 1;
#pragma unroll
        for (int r = 0; r < SH_R; ++r) {
          const float t = sh_quarter_sum(dAk[r]);
          dA0[r] += own0 ? t : 0.f;
          dA1[r] += own1 ? t : 0.f;
        }
      }
      // d logit B (softmax over r backward, the 1/3 tap weight folded in) and u = ds * B for this half
      LQTile<bf16, 4> dl, uu;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float smv = Bw[e][0] * slot[e][0];
#pragma unroll
        for (int r = 1; r < SH_R; ++r) smv = fmaf(Bw[e][r], slot[e][r], smv);
        const float dsc = ds[e] * third;
#pragma unroll
        for (int r = 0; r < SH_R; ++r) {
          dl.f[e >> 1][(e & 1) * 4 + r] = (bf16)(Bw[e][r] * dsc * (slot[e][r] - smv));
          uu.f[e >> 1][(e & 1) * 4 + r] = (bf16)(ds[e] * Bw[e][r]);
        }
      }
      if (inb) {
        bf16x8* up = reinterpret_cast<bf16x8*>(U + (int64_t)p * SH_NB + 64 * kc + 32 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) up[s] = uu.f[s];
      }
      // d feat += W_B^T d logit B over this half's k-steps
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) dfa[m] = mfma16(wBt[(m * 8 + 4 * h + s) * 64 + lw], dl.f[s], dfa[m]);
      {
        bf16x8* tp = reinterpret_cast<bf16x8*>(t_dlb + prow * SHB_PLB + 32 * kc);    // this half of d logit B for the pixel contraction
#pragma unroll
        for (int s = 0; s < 4; ++s) tp[s] = dl.f[s];
      }
      if (h == 1) {
        // d logit A of the lane quarter's own two filters: softmax over k backward (the sum over k spans the pixel's four lanes)
        const f32x4 A0 = *reinterpret_cast<const f32x4*>(arow + 8 * kc), A1 = *reinterpret_cast<const f32x4*>(arow + 8 * kc + 4);
        bf16x8 dlq, aq;                                         // channels 8 * kc + e <-> (k = 2 kc + e / 4, r = e % 4)
#pragma unroll
        for (int r = 0; r < SH_R; ++r) {
          const float d0 = dA0[r] * third, d1 = dA1[r] * third;
          const float dot = sh_quarter_sum(fmaf(A0[r], d0, A1[r] * d1));
          dlq[r] = (bf16)(A0[r] * (d0 - dot));
          dlq[4 + r] = (bf16)(A1[r] * (d1 - dot));
          aq[r] = (bf16)A0[r];
          aq[4 + r] = (bf16)A1[r];
        }
        if (inb) *reinterpret_cast<bf16x8*>(AS + (int64_t)p * SH_NA + 8 * kc) = aq;
        if (!inb) dlq = zeros;
#pragma unroll
        for (int m = 0; m < 4; ++m) dfa[m] = mfma16(wAt[m * 64 + lw], dlq, dfa[m]);
        if (inb) {
          bf16x8 o0, o1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { o0[j] = (bf16)dfa[j >> 2][j & 3]; o1[j] = (bf16)dfa[2 + (j >> 2)][j & 3]; }
          if (dfeat_relu) {                                       // feat = relu(.): hand the consumer d feat . [feat > 0] (its own mask pass is then not needed)
            const bf16x8* fp = reinterpret_cast<const bf16x8*>(t_ft + prow * SHB_PF + 16 * kc);
            const bf16x8 f0 = fp[0], f1 = fp[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (float)f0[j] > 0.f ? o0[j] : zero; o1[j] = (float)f1[j] > 0.f ? o1[j] : zero; }
          }
          bf16* fo = DFEAT + (int64_t)p * SH_HID + 16 * kc;
          *reinterpret_cast<bf16x8*>(fo) = o0;
          *reinterpret_cast<bf16x8*>(fo + 8) = o1;
        }
        // the pixel's A row has been read by its four lanes (same wave, program order): its bytes now carry the d-logit-A tile row
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        *reinterpret_cast<bf16x8*>(t_dla + prow * SHB_PLA + 8 * kc) = dlq;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < SHB_R / 32; ++ks) {
        const int pix0 = ks * 32 + 8 * kc;
        const bf16x8 af = sh_tr_frag(t_dlb, SHB_PLB, pix0, 16 * wave, r16);
#pragma unroll
        for (int i = 0; i < 4; ++i) gW[h][i] = mfma16(af, sh_tr_frag(t_ft, SHB_PF, pix0, 16 * i, r16), gW[h][i]);
        gb[h] = mfma16(af, ones, gb[h]);
        if (h == 1) {
          const bf16x8 aa = sh_tr_frag(t_dla, SHB_PLA, pix0, 16 * (wave >> 2), r16);
          gWa = mfma16(aa, sh_tr_frag(t_ft, SHB_PF, pix0, 16 * (wave & 3), r16), gWa);
          gba = mfma16(aa, ones, gba);
        }
      }
      __syncthreads();
    }
  }
  float* my = slab + (int64_t)blockIdx.x * SHB_SLAB;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row0 = 64 * (wave >> 1) + 16 * (wave & 1) + 32 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(row0 + 4 * kc + r) * SH_HID + 16 * i + r16] = gW[q][i][r];
    if (r16 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) my[SH_NB * SH_HID + SH_NA * SH_HID + row0 + 4 * kc + r] = gb[q][r];
    }
  }
  {
    const int row0 = 16 * (wave >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) my[SH_NB * SH_HID + (row0 + 4 * kc + r) * SH_HID + 16 * (wave & 3) + r16] = gWa[r];
    if (r16 == 0 && (wave & 3) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) my[SH_NB * SH_HID + SH_NA * SH_HID + SH_NB + row0 + 4 * kc + r] = gba[r];
    }
  }
}


// ------------------------------------------------------------------------------------------------ backward, kernel 2
// dx[p][c] = ds[p][c] / 3 + 1/3 sum_{k, +-} sum_r A[q][k][r] u[q][c][r],  q = p +- delta_k inside the image  (+ dx_add[p][c])
// Thread = (pixel, 8 channels); the rank contraction runs on packed bf16 pairs (v_dot2_f32_bf16).
typedef __attribute__((ext_vector_type(2))) __bf16 shbf2;
__device__ __forceinline__ float sh_dot4(const shbf2 b0, const shbf2 b1, const shbf2 a0, const shbf2 a1) {
  return __builtin_amdgcn_fdot2_f32_bf16(b0, a0, __builtin_amdgcn_fdot2_f32_bf16(b1, a1, 0.f, false), false);
}

__global__ __launch_bounds__(256, 4) void smooth_dx_kernel(const bf16* __restrict__ DS, const bf16* __restrict__ U, const bf16* __restrict__ AS,
                                                           const bf16* __restrict__ DXADD, bf16* __restrict__ DX, int B, int H, int W, int dil) {
  constexpr int VPR = 8, PPW = 256 / VPR;
  const int npix = B * H * W;
  const int nloop = (npix + PPW - 1) / PPW;
  const float third = 1.f / 3.f;
  for (int it = (int)xcd_remap(blockIdx.x, gridDim.x); it < nloop; it += (int)gridDim.x) {
    const int p = it * PPW + (int)threadIdx.x / VPR;
    const int cvi = (int)threadIdx.x % VPR;
    const bool active = p < npix;
    const int pc = active ? p : npix - 1;
    const int x = pc % W, y = (pc / W) % H;
    float dx[8];
    sh_unpack8(*reinterpret_cast<const bf16x8*>(DS + (int64_t)pc * SH_C + 8 * cvi), dx);     // centre taps of the 8 filters: ds / 3
#pragma unroll 2
    for (int k = 0; k < SH_K; ++k) {
      int dy, dxo;
      sh_dir(k, dil, dy, dxo);
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int qy = y + sgn * dy, qx = x + sgn * dxo;
        const bool ok = (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
        const int q = ok ? pc + sgn * (dy * W + dxo) : pc;
        const shbf2* aq = reinterpret_cast<const shbf2*>(AS + (int64_t)q * SH_NA + k * SH_R);
        shbf2 a0 = aq[0], a1 = aq[1];
        if (!ok) { a0 = shbf2{(bf16)0.f, (bf16)0.f}; a1 = a0; }
        const bf16x8* uq = reinterpret_cast<const bf16x8*>(U + (int64_t)q * SH_NB + 32 * cvi);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 v = uq[i];
          dx[2 * i] += sh_dot4(__builtin_shufflevector(v, v, 0, 1), __builtin_shufflevector(v, v, 2, 3), a0, a1);
          dx[2 * i + 1] += sh_dot4(__builtin_shufflevector(v, v, 4, 5), __builtin_shufflevector(v, v, 6, 7), a0, a1);
        }
      }
    }
    if (active) {
      float ad[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ad[e] = 0.f;
      if (DXADD != nullptr) sh_unpack8(*reinterpret_cast<const bf16x8*>(DXADD + (int64_t)p * SH_C + 8 * cvi), ad);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaf(third, dx[e], ad[e]);
      *reinterpret_cast<bf16x8*>(DX + (int64_t)p * SH_C + 8 * cvi) = o;
    }
  }
}

// Tiled form of kernel 2 (W a multiple of 16, a band + halo fits the LDS): a workgroup owns TH full-width rows of one image, its
// thread the 64 channels of one pixel.  The band's soft-maxed A rows (+- dil halo rows) arrive ONCE by LDS-DMA; u arrives one 8-channel
// group at a time (64 bytes per pixel, 16-byte chunks XOR-swizzled by (pixel >> 2) & 3 through the DMA's per-lane SOURCE address, so the
// ds_read_b128 of 16 consecutive pixels are conflict-free); the 16 star neighbours are then LDS reads instead of 72-byte gathers from
// L2, and every byte of u is fetched (TH + 2 dil) / TH times instead of 16.  dx leaves as whole 128-byte rows.
// LDS (dynamic, no static objects: byte offsets double as M0 values): A tile [ntp][64 B] | u tile [ntp][64 B].
__device__ __forceinline__ void sh_glds16(const void* src, int lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}

__global__ __launch_bounds__(512) void smooth_dx_tiled_kernel(const bf16* __restrict__ DS, const bf16* __restrict__ U, const bf16* __restrict__ AS,
                                                              const bf16* __restrict__ DXADD, bf16* __restrict__ DX, int B, int H, int W, int dil,
                                                              int TH, int tile_px_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = (int)blockDim.x >> 6;
  const int nband = (H + TH - 1) / TH;
  const int blk = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int b = blk / nband, y0 = (blk % nband) * TH;
  const int ylo = y0 - dil > 0 ? y0 - dil : 0, yhi = y0 + TH + dil < H ? y0 + TH + dil : H;
  const int ntp = (yhi - ylo) * W;                             // pixels of the tile (a multiple of 16)
  const int gbase = (b * H + ylo) * W;                         // global pixel of tile pixel 0 (full-width rows: the tile is contiguous)
  const int a_off = 0, u_off = tile_px_max * 64;
  const int nchunk = ntp * 4;
  // ---- A rows of the tile: 64 bytes per pixel, linear
  for (int c0 = wave * 64; c0 < nchunk; c0 += nwave * 64) {
    const int ch = c0 + lane;
    sh_glds16(AS + (int64_t)(gbase + (ch >> 2)) * SH_NA + (ch & 3) * 8, __builtin_amdgcn_readfirstlane(a_off + c0 * 16));
  }
  const int ty = y0 + tid / W, x = tid % W;
  const bool valid = ty < H;                                   // (H not a multiple of TH: the last band has idle threads)
  const int tme = ((valid ? ty : y0) - ylo) * W + x;
  const int p = (b * H + (valid ? ty : y0)) * W + x;
  float acc[SH_C];
  {
    const bf16x8* dp = reinterpret_cast<const bf16x8*>(DS + (int64_t)p * SH_C);   // centre taps of the 8 filters: ds / 3
#pragma unroll
    for (int i = 0; i < 8; ++i) sh_unpack8(dp[i], acc + 8 * i);
  }
  const char* ta = smem + a_off;
  const char* tu = smem + u_off;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    for (int c0 = wave * 64; c0 < nchunk; c0 += nwave * 64) {
      const int ch = c0 + lane, t = ch >> 2, pos = ch & 3;
      sh_glds16(U + (int64_t)(gbase + t) * SH_NB + g * 32 + ((pos ^ ((t >> 2) & 3)) << 3), __builtin_amdgcn_readfirstlane(u_off + c0 * 16));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float dx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dx[e] = 0.f;
#pragma unroll 2
    for (int k = 0; k < SH_K; ++k) {
      int dy, dxo;
      sh_dir(k, dil, dy, dxo);
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int qy = ty + sgn * dy, qx = x + sgn * dxo;
        const bool ok = valid && (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
        const int tq = ok ? tme + sgn * (dy * W + dxo) : tme;
        const shbf2* aq = reinterpret_cast<const shbf2*>(ta + tq * 64 + k * 8);
        shbf2 a0 = aq[0], a1 = aq[1];
        if (!ok) { a0 = shbf2{(bf16)0.f, (bf16)0.f}; a1 = a0; }
        const char* uq = tu + tq * 64;
        const int sw = (tq >> 2) & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(uq + ((i ^ sw) << 4));
          dx[2 * i] += sh_dot4(__builtin_shufflevector(v, v, 0, 1), __builtin_shufflevector(v, v, 2, 3), a0, a1);
          dx[2 * i + 1] += sh_dot4(__builtin_shufflevector(v, v, 4, 5), __builtin_shufflevector(v, v, 6, 7), a0, a1);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[8 * g + e] += dx[e];
    __syncthreads();                                            // every thread has read this group's u tile: the next group may land
  }
  if (valid) {
    const float third = 1.f / 3.f;
    bf16x8* op = reinterpret_cast<bf16x8*>(DX + (int64_t)p * SH_C);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float ad[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ad[e] = 0.f;
      if (DXADD != nullptr) sh_unpack8(reinterpret_cast<const bf16x8*>(DXADD + (int64_t)p * SH_C)[i], ad);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaf(third, acc[8 * i + e], ad[e]);
      op[i] = o;
    }
  }
}

static unsigned sh_fwd_grid(int64_t npix) {
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 1024) g = 1024;
  return (unsigned)(g < 1 ? 1 : g);
}

static int g_sh_force_gather = 0;

extern "C" {

// test hook: 1 = kernel 2 of the backward always takes the gather form (the tiled form is the default wherever it applies)
int frl_smooth_heads_force_gather(int on) { const int was = g_sh_force_gather; g_sh_force_gather = on ? 1 : 0; return was; }

int frl_smooth_heads_supported(int C, int hidden, int rank, int dtype) {
  return (dtype == FRL_BF16 && C == SH_C && hidden == SH_HID && rank == SH_R) ? 1 : 0;
}

size_t frl_smooth_heads_workspace_bytes(int64_t npix) {
  (void)npix;
  const size_t pk = ((size_t)(SH_FR_B + SH_FR_A + SH_FR_BT + SH_FR_AT) * sizeof(frag8) + 255) / 256 * 256;
  const size_t slab = (size_t)256 * (SH_NB * SH_HID + SH_NA * SH_HID + SH_NB + SH_NA) * sizeof(float);
  return pk + slab + 512;
}

// x [B][H][W][64], feat [B][H][W][64] bf16; wa [32][64], ba [32], wb [256][64], bb [256] f32 (output channel k * R + r resp. c * R + r)
// -> smoothed, residual [B][H][W][64] bf16
int frl_smooth_heads_fwd(const void* x, const void* feat, const float* wa, const float* ba, const float* wb, const float* bb, void* smoothed,
                         void* residual, int B, int H, int W, int dil, void* ws, size_t ws_bytes, hipStream_t stream) {
  const int64_t npix = (int64_t)B * H * W;
  if (npix <= 0) return frl_fail(-2, "smooth_heads_fwd: empty input");
  if (npix * SH_NB >= (int64_t)1 << 31) return frl_fail(-2, "smooth_heads_fwd: too many pixels for 32-bit element offsets");
  if (dil < 1) return frl_fail(-2, "smooth_heads_fwd: dilation must be positive");
  if (ws_bytes < frl_smooth_heads_workspace_bytes(npix)) return frl_fail(-4, "smooth_heads_fwd: workspace too small");
  const frag8* pk = sh_packed(wa, wb, 0, reinterpret_cast<frag8*>(ws), stream);
  const size_t lds = (size_t)(SH_FR_B + SH_FR_A) * sizeof(frag8) + (SH_NB + SH_NA + 64 * SH_PA) * sizeof(float);
  FRL_HIP(hipFuncSetAttribute((const void*)smooth_heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH(smooth_heads_fwd_kernel, dim3(sh_fwd_grid(npix)), dim3(256), lds, stream, (const bf16*)x, (const bf16*)feat, pk, ba, bb,
             (bf16*)smoothed, (bf16*)residual, B, H, W, dil);
  return frl_check_launch("smooth_heads_fwd");
}


// Backward of frl_smooth_heads_fwd.  d_smoothed = gradient w.r.t. `smoothed` with the residual path already folded in by the caller
// (d smoothed - d residual); dx_add (optional) is added to dx (the direct path of x into the residual).  Outputs: dx, dfeat [P][64] bf16,
// dwa [32][64], dba [32], dwb [256][64], dbb [256] f32.  scratch: u [P][256] + a_soft [P][32] bf16, caller-provided (exchange tensors
// between the two launches): frl_smooth_heads_bwd_scratch_bytes.
int frl_smooth_heads_bwd_masked(const void* d_smoothed, const void* x, const void* feat, const float* wa, const float* ba, const float* wb,
                                const float* bb, const void* dx_add, void* dx, void* dfeat, float* dwa, float* dba, float* dwb, float* dbb,
                                void* scratch, size_t scratch_bytes, int B, int H, int W, int dil, int dfeat_relu, void* ws, size_t ws_bytes,
                                hipStream_t stream);
int frl_smooth_heads_bwd(const void* d_smoothed, const void* x, const void* feat, const float* wa, const float* ba, const float* wb,
                         const float* bb, const void* dx_add, void* dx, void* dfeat, float* dwa, float* dba, float* dwb, float* dbb,
                         void* scratch, size_t scratch_bytes, int B, int H, int W, int dil, void* ws, size_t ws_bytes, hipStream_t stream) {
  return frl_smooth_heads_bwd_masked(d_smoothed, x, feat, wa, ba, wb, bb, dx_add, dx, dfeat, dwa, dba, dwb, dbb, scratch, scratch_bytes, B, H, W, dil, 0,
                                     ws, ws_bytes, stream);
}

// The same with dfeat_relu != 0: dfeat is returned already multiplied by [feat > 0] (feat is the output of a ReLU convolution, mix_backbone
// of spatial.py:258-261), so that convolution's backward-data / backward-weight calls take it without their own activation mask.
int frl_smooth_heads_bwd_masked(const void* d_smoothed, const void* x, const void* feat, const float* wa, const float* ba, const float* wb,
                                const float* bb, const void* dx_add, void* dx, void* dfeat, float* dwa, float* dba, float* dwb, float* dbb,
                                void* scratch, size_t scratch_bytes, int B, int H, int W, int dil, int dfeat_relu, void* ws, size_t ws_bytes,
                                hipStream_t stream) {
  const int64_t npix = (int64_t)B * H * W;
  if (npix <= 0) return frl_fail(-2, "smooth_heads_bwd: empty input");
  if (npix * SH_NB >= (int64_t)1 << 31) return frl_fail(-2, "smooth_heads_bwd: too many pixels for 32-bit element offsets");
  if (dil < 1) return frl_fail(-2, "smooth_heads_bwd: dilation must be positive");
  if (ws_bytes < frl_smooth_heads_workspace_bytes(npix)) return frl_fail(-4, "smooth_heads_bwd: workspace too small");
  if (scratch_bytes < (size_t)npix * (SH_NB + SH_NA) * sizeof(bf16)) return frl_fail(-4, "smooth_heads_bwd: scratch too small");
  char* w = (char*)ws;
  const size_t pkb = ((size_t)(SH_FR_B + SH_FR_A + SH_FR_BT + SH_FR_AT) * sizeof(frag8) + 255) / 256 * 256;
  const frag8* pk = sh_packed(wa, wb, 1, reinterpret_cast<frag8*>(w), stream);
  float* slab = reinterpret_cast<float*>(w + pkb);
  bf16* u = (bf16*)scratch;
  bf16* as = u + npix * SH_NB;
  int64_t g = (npix + SHB_R - 1) / SHB_R;
  if (g > 256) g = 256;
  const size_t lds = (size_t)(SH_FR_B + SH_FR_A + SH_FR_BT + SH_FR_AT) * sizeof(frag8) + (SH_NB + SH_NA + SHB_R * SH_PA) * sizeof(float) +
                     (size_t)SHB_R * (SHB_PLB + SHB_PF) * sizeof(bf16);
  FRL_HIP(hipFuncSetAttribute((const void*)smooth_heads_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH(smooth_heads_bwd_kernel, dim3((unsigned)g), dim3(64 * SHB_NW), lds, stream, (const bf16*)x, (const bf16*)feat, (const bf16*)d_smoothed,
             pk, ba, bb, (bf16*)dfeat, u, as, slab, B, H, W, dil, dfeat_relu);
  launch_slab_reduce_deferrable<float, ShEpi>((const float*)slab, (int)g, (int64_t)SHB_SLAB, ShEpi{dwb, dwa, dbb, dba, SH_NB, SH_NA, SH_HID}, stream);
  // kernel 2: tiled through the LDS when full-width bands of TH rows (+ dil halo rows each side) make whole waves and fit, else gathers
  int TH = 0;
  if (W % 16 == 0 && W <= 512 && g_sh_force_gather == 0) {
    for (int th = 16; th >= 1; th >>= 1)
      if ((th * W) % 64 == 0 && th * W <= 512 && (size_t)(th + 2 * dil) * W * 128 <= 72 * 1024) { TH = th; break; }
  }
  if (TH > 0) {
    const int nband = (H + TH - 1) / TH, tile_px_max = (TH + 2 * dil) * W;
    const size_t lds2 = (size_t)tile_px_max * 128;
    FRL_HIP(hipFuncSetAttribute((const void*)smooth_dx_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    FRL_LAUNCH(smooth_dx_tiled_kernel, dim3((unsigned)(B * nband)), dim3((unsigned)(TH * W)), lds2, stream, (const bf16*)d_smoothed, (const bf16*)u,
               (const bf16*)as, (const bf16*)dx_add, (bf16*)dx, B, H, W, dil, TH, tile_px_max);
  } else {
    int64_t g2 = (npix + 31) / 32;
    if (g2 > 2048) g2 = 2048;
    FRL_LAUNCH(smooth_dx_kernel, dim3((unsigned)g2), dim3(256), 0, stream, (const bf16*)d_smoothed, (const bf16*)u, (const bf16*)as,
               (const bf16*)dx_add, (bf16*)dx, B, H, W, dil);
  }
  return frl_check_launch("smooth_heads_bwd");
}

size_t frl_smooth_heads_bwd_scratch_bytes(int64_t npix) { return (size_t)npix * (SH_NB + SH_NA) * sizeof(bf16); }

}  // extern "C"
