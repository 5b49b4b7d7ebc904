// Fused mixing heads + softmaxes + directional filter bank of EdgeAwareSmoothingConv2D for the hot configuration
// (bf16, 64 channels, 64 hidden features, 4 directions x 2 scales, rank 4):
//   A   = softmax_k(W_A feat + b_A)   viewed [K = 8][R = 4]          frl/models/spatial.py:262,300-302   (K7)
//   B   = softmax_r(W_B feat + b_B)   viewed [C = 64][R = 4]         spatial.py:263,305-307              (K8)
//   smoothed[c] = sum_k (sum_r A[k][r] B[c][r]) * f_k[c],  f_k = 3-tap line average of x along direction k / scale (K9, K10)
//   residual = x - smoothed                                           spatial.py:314-331
// Forward (smooth_heads_fwd_kernel): ONE pass over (feat, x): both 1x1 heads run on the matrix cores with the wave's 16-pixel
//   lane-quarter image as the B operand, the [P,32] + [P,256] logits, their soft-maxed copies and the 8 filtered copies of x live in
//   registers only (the modular chain writes and re-reads ~0.6 GB of them per step at cfg2).  Head-A rows are REPLICATED over the four
//   lane quarters (8 blocks instead of 2: 12 extra MFMAs per 16 pixels) so that every lane holds all 32 direction weights of its
//   pixel and the rank contraction needs no cross-lane traffic.
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

// fragment counts of the packed weight images
#define SH_FR_B (16 * 2 * 64)   // W_B  forward:    MB = 16, NF = 2
#define SH_FR_A (8 * 2 * 64)    // W_A  replicated: MB = 8,  NF = 2
#define SH_FR_BT (4 * 8 * 64)   // W_B^T: d feat <- d logit B, MB = 4, NF = 8
#define SH_FR_AT (4 * 1 * 64)   // W_A^T: d feat <- d logit A, MB = 4, NF = 1

// pack_weights_lds with the rows of every 16-row block REPLICATED over the four lane quarters: row r of block mb <-> oc = 4 * mb + (r & 3)
__device__ __forceinline__ void sh_pack_rep(frag8* __restrict__ dst, const float* __restrict__ W, int Cout, int Cin, int MB, int tid, int nthreads) {
  const int total = MB * 2 * 64;
  for (int i = tid; i < total; i += nthreads) {
    const int lane = i & 63, fs = i >> 6;
    const int s = fs & 1, mb = fs >> 1;
    const int r = lane & 15, kc = lane >> 4;
    const int oc = 4 * mb + (r & 3);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ic = 16 * kc + 8 * s + e;
      v[e] = (oc < Cout && ic < Cin) ? (bf16)W[oc * Cin + ic] : (bf16)0.f;
    }
    dst[i] = v;
  }
}

__global__ void sh_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ WA, const float* __restrict__ WB, int bwd) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  pack_weights_lds<bf16, 2>(dst, WB, SH_NB, SH_HID, 16, SH_HID, 1, tid, nt);                       // logit B[o] = sum_i W_B[o][i] feat[i]
  sh_pack_rep(dst + SH_FR_B, WA, SH_NA, SH_HID, 8, tid, nt);
  if (!bwd) return;
  pack_weights_lds<bf16, 8>(dst + SH_FR_B + SH_FR_A, WB, SH_HID, SH_NB, 4, 1, SH_HID, tid, nt);    // d feat[o] = sum_l W_B[l][o] d logit B[l]
  pack_weights_lds<bf16, 1>(dst + SH_FR_B + SH_FR_A + SH_FR_BT, WA, SH_HID, SH_NA, 4, 1, SH_HID, tid, nt);
}

static const frag8* sh_packed(const float* wa, const float* wb, int bwd, frag8* ws_pk, hipStream_t st) {
  FrlPackJob jobs[4];
  size_t off = 0;
  jobs[0] = frl_pack_job_pw(wb, off, FRL_BF16, 2, SH_NB, SH_HID, 16, SH_HID, 1);
  off += (size_t)SH_FR_B * sizeof(frag8);
  jobs[1] = frl_pack_job_pw(wa, off, FRL_BF16, 2, SH_NA, SH_HID, 8, SH_HID, 1);
  jobs[1].kind = FRL_PACK_PW_REP;
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

// Both heads + both softmaxes of one 16-pixel tile.  Out: A[k][r] (every lane: all 32 of its pixel) and Bw[m][r] for the lane's 16
// channels c = 16 * kc + m.  wB / wA: packed images in LDS; tb: b_B[256] | b_A[32] in LDS.
__device__ __forceinline__ void sh_heads(f32x4 (&A)[SH_K], f32x4 (&Bw)[16], const LQTile<bf16, 2>& ft, const frag8* __restrict__ wB,
                                         const frag8* __restrict__ wA, const float* __restrict__ tb, int lane, int kc) {
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(tb + 64 * kc + 4 * m);
    acc = mfma16(wB[(m * 2 + 0) * 64 + lane], ft.f[0], acc);
    Bw[m] = mfma16(wB[(m * 2 + 1) * 64 + lane], ft.f[1], acc);
  }
#pragma unroll
  for (int k = 0; k < SH_K; ++k) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(tb + SH_NB + 4 * k);
    acc = mfma16(wA[(k * 2 + 0) * 64 + lane], ft.f[0], acc);
    A[k] = mfma16(wA[(k * 2 + 1) * 64 + lane], ft.f[1], acc);
  }
  // softmax over the 8 filters for each rank slot (spatial.py:300-302)
#pragma unroll
  for (int r = 0; r < SH_R; ++r) {
    float m = A[0][r];
#pragma unroll
    for (int k = 1; k < SH_K; ++k) m = fmaxf(m, A[k][r]);
    const float ms = m * SH_LOG2E;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SH_K; ++k) { A[k][r] = __builtin_amdgcn_exp2f(fmaf(A[k][r], SH_LOG2E, -ms)); s += A[k][r]; }
    const float inv = __builtin_amdgcn_rcpf(s);
#pragma unroll
    for (int k = 0; k < SH_K; ++k) A[k][r] *= inv;
  }
  // softmax over the 4 rank slots for each channel (spatial.py:305-307)
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const float mx = fmaxf(fmaxf(Bw[m][0], Bw[m][1]), fmaxf(Bw[m][2], Bw[m][3])) * SH_LOG2E;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < SH_R; ++r) { Bw[m][r] = __builtin_amdgcn_exp2f(fmaf(Bw[m][r], SH_LOG2E, -mx)); s += Bw[m][r]; }
    const float inv = __builtin_amdgcn_rcpf(s);
#pragma unroll
    for (int r = 0; r < SH_R; ++r) Bw[m][r] *= inv;
  }
}

__device__ __forceinline__ void sh_unpack8(const bf16x8 v, float* o) {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void smooth_heads_fwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ FEAT,
                                                                  const frag8* __restrict__ Wpk, const float* __restrict__ bA,
                                                                  const float* __restrict__ bB, bf16* __restrict__ SM, bf16* __restrict__ RES,
                                                                  int B, int H, int W, int dil) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* wB = reinterpret_cast<frag8*>(smem);
  frag8* wA = wB + SH_FR_B;
  float* tb = reinterpret_cast<float*>(wA + SH_FR_A);         // b_B[256] | b_A[32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<bf16>(wB, Wpk, SH_FR_B + SH_FR_A, tid, 256);
  for (int i = tid; i < SH_NB + SH_NA; i += 256) tb[i] = i < SH_NB ? bB[i] : bA[i - SH_NB];
  __syncthreads();
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
    const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(xp), c1 = *reinterpret_cast<const bf16x8*>(xp + 8);
    f32x4 A[SH_K], Bw[16];
    int lw = lane;
    asm volatile("" : "+v"(lw));                               // (opaque per tile: the weight fragments stay in LDS instead of hoisted registers)
    sh_heads(A, Bw, ft, wB, wA, tb, lw, kc);
    float ctr[16], sm[16];
    sh_unpack8(c0, ctr);
    sh_unpack8(c1, ctr + 8);
#pragma unroll
    for (int e = 0; e < 16; ++e) sm[e] = 0.f;
#pragma unroll
    for (int k = 0; k < SH_K; ++k) {
      int dy, dx;
      sh_dir(k, dil, dy, dx);
      float f[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] = ctr[e];
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int qy = y + sgn * dy, qx = x + sgn * dx;
        const bool ok = (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
        const bf16* qp = ok ? xp + (int64_t)(sgn * (dy * W + dx)) * SH_C : xp;     // out of the image: zero padding (weight 0 on a safe address)
        const float wq = ok ? 1.f : 0.f;
        float v[16];
        sh_unpack8(*reinterpret_cast<const bf16x8*>(qp), v);
        sh_unpack8(*reinterpret_cast<const bf16x8*>(qp + 8), v + 8);
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = fmaf(wq, v[e], f[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float w = A[k][0] * Bw[e][0];
        w = fmaf(A[k][1], Bw[e][1], w);
        w = fmaf(A[k][2], Bw[e][2], w);
        w = fmaf(A[k][3], Bw[e][3], w);
        sm[e] = fmaf(w, f[e], sm[e]);
      }
    }
    if (inb) {
      bf16x8 s0, s1, r0, r1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = third * sm[e], b = third * sm[8 + e];
        s0[e] = (bf16)a; s1[e] = (bf16)b;
        r0[e] = (bf16)(ctr[e] - a); r1[e] = (bf16)(ctr[8 + e] - b);
      }
      bf16* so = SM + (int64_t)p * SH_C + 16 * kc;
      bf16* ro = RES + (int64_t)p * SH_C + 16 * kc;
      *reinterpret_cast<bf16x8*>(so) = s0;
      *reinterpret_cast<bf16x8*>(so + 8) = s1;
      *reinterpret_cast<bf16x8*>(ro) = r0;
      *reinterpret_cast<bf16x8*>(ro + 8) = r1;
    }
  }
}

static unsigned sh_fwd_grid(int64_t npix) {
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 1024) g = 1024;
  return (unsigned)(g < 1 ? 1 : g);
}

extern "C" {

int frl_smooth_heads_supported(int C, int hidden, int rank, int dtype) {
  return (dtype == FRL_BF16 && C == SH_C && hidden == SH_HID && rank == SH_R) ? 1 : 0;
}

size_t frl_smooth_heads_workspace_bytes(int64_t npix) {
  (void)npix;
  const size_t pk = (size_t)(SH_FR_B + SH_FR_A + SH_FR_BT + SH_FR_AT) * sizeof(frag8);
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
  const size_t lds = (size_t)(SH_FR_B + SH_FR_A) * sizeof(frag8) + (SH_NB + SH_NA) * sizeof(float);
  FRL_HIP(hipFuncSetAttribute((const void*)smooth_heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  FRL_LAUNCH(smooth_heads_fwd_kernel, dim3(sh_fwd_grid(npix)), dim3(256), lds, stream, (const bf16*)x, (const bf16*)feat, pk, ba, bb,
             (bf16*)smoothed, (bf16*)residual, B, H, W, dil);
  return frl_check_launch("smooth_heads_fwd");
}

}  // extern "C"
