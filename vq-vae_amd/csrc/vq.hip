// Vector-quantisation step: nearest-codebook assignment, straight-through / commitment gradients,
// EMA codebook statistics.  The reference no longer ships a quantizer; the definition implemented here
// is SURVEY.md section 8a row a11 (constants: frl/config/frl_model_v0.yaml:29-35,
// frl/config/frl_bindings_v0.yaml:887-891, scripts/train_vqvae.py:410-436).
//
// vq_assign: for every row z_n find argmin_k ||z_n - e_k||^2 (first index on ties), bit-exact against a
// float64 evaluation, while running the -2 z.e term on the matrix cores:
//   * codebook chunk (<= 512 codes) staged once per workgroup in LDS as packed MFMA A fragments holding
//     -2*e, plus ||e||^2; z streams from HBM as lane-quarter B fragments (16 vectors per tile, 16 B / lane);
//   * accumulator initialised with ||e_k||^2 + ||z_n||^2 (+ a per-vector positive bias) so that the MFMA
//     result is the (positive) squared distance;  key = (f32 bits & ~511) | local index, so that ONE
//     v_min_u32 tracks the running arg-min and ONE v_med3_u32 the runner-up, wave-level min-reduce over
//     the 4 lane groups by __shfl_xor at the end;
//   * rows whose runner-up is within the rigorous rounding + truncation bound of the minimum are flagged
//     (idx = -1 - provisional) and re-evaluated exactly in float64 (<1 % of rows): inside the kernel when the codebook is resident,
//     by vq_fixup_tile_kernel (16 rows per wave, candidates screened on the matrix cores) on the multi-chunk path;
//   * z_q gather, squared-error partial sums and the per-workgroup code histogram are fused in.
// Roofline: HBM for K <= 1024 (algorithmic bytes/vector = 2*d*s + 4), MFMA for K = 8192.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_reduce.hpp"
#include <math.h>
#include <stdlib.h>

#define VQ_IDX_BITS 7
#define VQ_IDX_MASK 127u        // keys carry a 7-bit index inside groups of 128 codes (8 MFMA row blocks)
#define VQ_MAX_CHUNK 512

struct VqHeader {           // lives at the start of the workspace (zeroed by hipMemsetAsync every call)
  unsigned enmax_bits;      // (unused; kept for the layout)
  int namb;                 // number of rows flagged for exact re-evaluation (append counter of amb_list / total of the fused path)
  double sq_fix;            // sum of squared errors of the re-evaluated rows (multi-chunk path)
  int done;                 // fused path: workgroups that have published their partial results (arrival ticket)
  int pad;
};

// prep + pack in ONE launch (blocks [0, npack) write fragments, the rest the norms): the "prepared codebook" image
// [en: kpad floats][packed fragments] that frl_vq_prepare hands to the caller, who keeps it until the codebook changes
template <typename T, int NF>
__global__ void vq_prepare_kernel(const float* __restrict__ E, int K, int d, float* __restrict__ en, int kpad,
                                  typename DT<T>::frag_t* __restrict__ pk, int total, int npack_blocks, int* __restrict__ ctl) {
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE;
  if ((int)blockIdx.x >= npack_blocks) {
    const int k = ((int)blockIdx.x - npack_blocks) * blockDim.x + threadIdx.x;
    if (k >= kpad) return;
    if (k < 64) ctl[k] = 0;                                  // control block of the resident kernel: {done, namb, ...}, counts_acc[K]
    if (k < K) ctl[64 + k] = 0;
    float s = 3.0e38f;                                     // codes beyond K never win
    if (k < K) {
      s = 0.f;
      for (int j = 0; j < d; ++j) {
        const float v = to_f32(from_f32<T>(E[(int64_t)k * d + j]));
        s = fmaf(v, v, s);
      }
    }
    en[k] = s;
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ln = i & 63, fs = i >> 6;
  const int s = fs % NF, mb = fs / NF;
  const int code = mb * 16 + (ln & 15), kq = ln >> 4;
  if constexpr (FE == 8) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = q * kq + 8 * s + e;
      v[e] = (code < K && ch < d) ? (bf16)(-2.f * (float)(bf16)E[(int64_t)code * d + ch]) : (bf16)0.f;
    }
    pk[i] = v;
  } else {
    const int ch = q * kq + s;
    pk[i] = (code < K && ch < d) ? -2.f * E[(int64_t)code * d + ch] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// main assignment kernel
// ---------------------------------------------------------------------------------------------
// NW = waves per workgroup: 16 (one workgroup per CU, ONE LDS copy of the codebook chunk shared by four waves per SIMD) when the
// batch is large enough, else 4.
#ifdef VQ_STAMPS
__device__ unsigned long long* vq_dbg;      // diagnostic build only (tools/diag/vq_stamps.hip)
#endif
// exact squared distances of row n to the codes k0 and k1 (k1 < 0: only k0) in float64; operands rounded to T first, as everywhere
// in this file.  16-byte loads, four chunks in flight: the caller is one lane per candidate pair, the latency of the loads is all
// there is to hide.
template <typename T, int U = 4>
__device__ __forceinline__ void vq_exact_pair(const T* __restrict__ Z, const float* __restrict__ E, int64_t n, int k0, int k1, int d,
                                              double& d0, double& d1) {
  const T* zr = Z + n * (int64_t)d;
  const float* e0 = E + (int64_t)k0 * d;
  const float* e1 = E + (int64_t)(k1 >= 0 ? k1 : k0) * d;
  double s0 = 0.0, s1 = 0.0;
  if ((d & 7) == 0) {
#pragma unroll U
    for (int j = 0; j < d; j += 8) {
      float zv[8];
      if constexpr (sizeof(T) == 2) {
        Vec<T>::load(zr + j, zv);
      } else {
        Vec<T>::load(zr + j, zv);
        Vec<T>::load(zr + j + 4, zv + 4);
      }
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(e0 + j), a1 = *reinterpret_cast<const f32x4*>(e0 + j + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(e1 + j), b1 = *reinterpret_cast<const f32x4*>(e1 + j + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const double zz = (double)zv[e];
        const double da = zz - (double)to_f32(from_f32<T>(e < 4 ? a0[e & 3] : a1[e & 3]));
        const double db = zz - (double)to_f32(from_f32<T>(e < 4 ? b0[e & 3] : b1[e & 3]));
        s0 = __builtin_fma(da, da, s0);
        s1 = __builtin_fma(db, db, s1);
      }
    }
  } else {
    for (int j = 0; j < d; ++j) {
      const double zz = (double)to_f32(zr[j]);
      const double da = zz - (double)to_f32(from_f32<T>(e0[j])), db = zz - (double)to_f32(from_f32<T>(e1[j]));
      s0 = __builtin_fma(da, da, s0);
      s1 = __builtin_fma(db, db, s1);
    }
  }
  d0 = s0;
  d1 = s1;
}

// Multi-chunk assignment (codebook image larger than one LDS chunk: K = 8192, d = 128 of BASELINE configs[3]).  The packed image is
// streamed through TWO LDS chunk buffers by LDS-DMA (global_load_lds_dwordx4: the image is contiguous, no registers involved): chunk
// i + 1 arrives while chunk i is scored, one barrier per chunk, the chunk sequence runs on across the workgroup's batches.  Keys are the
// float-ordered ones of the resident kernel (no batch bias, per-row threshold), flagged rows go to a global list with their limits
// (vq_fixup_tile_kernel), the histogram to integer atomics on the zeroed accumulator.
template <typename T, int NF, int NT, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 2)) void vq_assign_kernel(
    const T* __restrict__ Z, const float* __restrict__ E, const float* __restrict__ en_g, int64_t N, int K, int d, int Kc,
    int32_t* __restrict__ idx_out, T* __restrict__ zq_out, float* __restrict__ partial /*[grid*NW]*/, int32_t* __restrict__ counts_acc,
    VqHeader* __restrict__ hdr_w, int32_t* __restrict__ amb_list, float* __restrict__ amb_lim, const typename DT<T>::frag_t* __restrict__ pk) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE;                 // channels per lane quarter
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int chunk_bytes = (Kc / 16) * NF * 64 * (int)sizeof(frag_t);  // a multiple of 1 KB
  const int buf_bytes = chunk_bytes + Kc * 4;                         // fragments | ||e||^2 of the chunk
  float* cbw = reinterpret_cast<float*>(smem + 2 * buf_bytes);         // [NW] scratch of the prologue
  // histogram: small codebooks are hit by many rows per code, so the workgroup counts in LDS and adds each used code once at the end;
  // large ones (K > 2048: few rows per code, 32 KB of LDS) go straight to the global integer atomics
  int* hist = reinterpret_cast<int*>(smem + 2 * buf_bytes + 64);
  const bool lds_hist = K <= 2048;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vx = lane & 15, kc = lane >> 4;
  const int nchunks = (K + Kc - 1) / Kc;
  const bool fast = (d == 4 * q);
  float enmax;
  {                                                                    // max ||e||^2 over the real codes, by the workgroup itself
    float m = 0.f;
    for (int k = tid; k < K; k += NW * 64) m = fmaxf(m, en_g[k]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) cbw[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, cbw[w]);
    enmax = mm;
  }
  const float err_rel = (float)(4 * q + 8) * 1.1920929e-7f;            // (d_pad+8) * 2^-23: f32 accumulation of exact products
  const float thr_rel = (3.0517578125e-5f + 2.f * err_rel) * 1.001953125f;   // 2^-15: key truncation of both scores, + 2*err, + margin
  const float enroot = sqrtf(enmax);
  float sq_acc = 0.f;
  if (lds_hist)
    for (int k = tid; k < K; k += NW * 64) hist[k] = 0;            // (visible behind the first chunk barrier)

  // chunk c of the image (fragments, then its norms) into buffer `buf`: 1 KB pieces dealt round-robin to the waves
  const int npiece = chunk_bytes >> 10;
  auto dma_chunk = [&](int c, int buf) {
    const char* src = reinterpret_cast<const char*>(pk) + (size_t)c * chunk_bytes;
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    for (int p = wave; p < npiece; p += NW) {                      // (wave-uniform trip count)
      const char* sp = src + p * 1024 + lane_ * 16;
      const int ldst = buf * buf_bytes + p * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(sp), "s"(ldst) : "memory");
    }
    for (int p = wave; p < (Kc >> 6); p += NW) {                   // the chunk's norms, 64 floats per piece (en_g holds kpad entries: 3e38 beyond K)
      const float* sp = en_g + c * Kc + p * 64 + lane_;
      const int ldst = buf * buf_bytes + chunk_bytes + p * 256;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(sp), "s"(ldst) : "memory");
    }
  };

  const int64_t vec_per_batch = NW * NT * 16;
  const int64_t nbatch = (N + vec_per_batch - 1) / vec_per_batch;
  const int64_t my_batches = (int64_t)blockIdx.x < nbatch ? (nbatch - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total_it = my_batches * nchunks;                  // chunk iterations of this workgroup, across its batches
  int64_t it = 0;
  if (total_it > 0) dma_chunk(0, 0);
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int64_t v0 = (batch * NW + wave) * (NT * 16);
    LQTile<T, NF> zt[NT];
    float thr[NT];
    unsigned g1[NT], g2[NT];
    int gc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int64_t row = v0 + t * 16 + vx;
      if (row >= N) row = N - 1;
      lq_load<T, NF>(zt[t], Z, row, d, kc, fast);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float zn = 0.f;
#pragma unroll
      for (int s = 0; s < NF; ++s)
#pragma unroll
        for (int e = 0; e < FE; ++e) { const float v = lq_get<T, NF>(zt[t], s, e); zn = fmaf(v, v, zn); }
      zn += __shfl_xor(zn, 16, 64);
      zn += __shfl_xor(zn, 32, 64);
      const float sroot = sqrtf(zn) + enroot;                      // |score| and every partial sum <= (||z|| + ||e||max)^2
      thr[t] = sroot * sroot * thr_rel + 1e-37f;
#pragma unroll
      for (int s = 0; s < NF; ++s) asm volatile("" : "+v"(zt[t].f[s]));   // (else the unpacked floats stay live through the chunk loop)
      g1[t] = 0x7F800000u; g2[t] = 0x7F800000u; gc[t] = 0;         // +inf
    }
    unsigned maskv = ~VQ_IDX_MASK;
    asm volatile("" : "+v"(maskv));                                // (opaque and in a register: (score & mask) | index then selects as ONE v_and_or_b32)
    for (int c = 0; c < nchunks; ++c, ++it) {
      const int buf = (int)(it & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of chunk `it` (and its norm loads) have landed
      __syncthreads();                                           // ... everybody's; everybody is done with the other buffer
      if (it + 1 < total_it) dma_chunk((c + 1 == nchunks) ? 0 : c + 1, buf ^ 1);
      const frag_t* wl = reinterpret_cast<const frag_t*>(smem + buf * buf_bytes);
      const float* enl = reinterpret_cast<const float*>(smem + buf * buf_bytes + chunk_bytes);
      const int nmb = Kc / 16;
      for (int g0 = 0; g0 < nmb; g0 += 8) {
        unsigned c1[NT], c2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { c1[t] = 0x7F800000u; c2[t] = 0x7F800000u; }
        const int g1e = (g0 + 8) < nmb ? (g0 + 8) : nmb;
        for (int mb = g0; mb < g1e; ++mb) {
          const f32x4 en4 = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kc);
          frag_t a[NF];
#pragma unroll
          for (int s = 0; s < NF; ++s) a[s] = wl[(mb * NF + s) * 64 + lane];
          const unsigned lidx = (unsigned)((mb - g0) * 4);            // (wave-uniform)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            f32x4 acc = en4;
#pragma unroll
            for (int s = 0; s < NF; ++s) acc = mfma16(a[s], zt[t].f[s], acc);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              // key index = (block in group) * 4 + accumulator row (the lane adds its quarter when it decodes); two keys per update:
              // with c1 <= c2 the second smallest of {c1, c2, k0, k1} is min(c2, median(c1, k0, k1)); 2.5 vector operations per score
              unsigned i0 = lidx + r, i1 = lidx + r + 1;
              asm("" : "+s"(i0), "+s"(i1));                            // (opaque scalars: else the odd index becomes v_and + v_or3)
              const unsigned k0 = (__float_as_uint(acc[r]) & maskv) | i0, k1 = (__float_as_uint(acc[r + 1]) & maskv) | i1;
              unsigned m;
              asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(c1[t]), "v"(k0), "v"(k1));
              asm("v_min_f32 %0, %1, %2" : "=v"(c2[t]) : "v"(c2[t]), "v"(m));
              asm("v_min3_f32 %0, %1, %2, %3" : "=v"(c1[t]) : "v"(c1[t]), "v"(k0), "v"(k1));
            }
          }
        }
        const int gid = c * (Kc / 16) + g0;                     // group base in units of 16 codes
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float a1 = __uint_as_float(g1[t]), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(c1[t]), b2 = __uint_as_float(c2[t]);
          const float hi = fmaxf(a1, b1), lo2 = fminf(a2, b2);
          g2[t] = __float_as_uint(fminf(hi, lo2));
          if (b1 < a1) { g1[t] = c1[t]; gc[t] = gid; }
        }
      }
    }
    // ---- min-reduce over the 4 lane groups that share a row; ambiguity test; z_q, squared error, histogram of the settled rows ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int cd1 = (gc[t] + (int)((g1[t] & VQ_IDX_MASK) >> 2)) * 16 + 4 * kc + (int)(g1[t] & 3u);   // the lane's own best code
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const unsigned o1 = __shfl_xor(g1[t], off, 64), o2 = __shfl_xor(g2[t], off, 64);
        const int oc = __shfl_xor(cd1, off, 64);
        const float a1 = __uint_as_float(g1[t] & ~VQ_IDX_MASK), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(o1 & ~VQ_IDX_MASK), b2 = __uint_as_float(o2);
        const float hi = fmaxf(__uint_as_float(g1[t]), __uint_as_float(o1)), lo2 = fminf(a2, b2);
        g2[t] = __float_as_uint(fminf(hi, lo2));
        if (b1 < a1 || (b1 == a1 && oc < cd1)) { g1[t] = o1; cd1 = oc; }
      }
      const int64_t row = v0 + t * 16 + vx;
      const int code = cd1;
      const float s1 = __uint_as_float(g1[t] & ~VQ_IDX_MASK), s2 = __uint_as_float(g2[t] & ~VQ_IDX_MASK);
      const bool amb = !((s2 - s1) > thr[t]) || (unsigned)code >= (unsigned)K;   // also catches NaN / inf rows
      if (row < N) {
        if (kc == 0) {
          idx_out[row] = amb ? 0 : code;
          if (amb) {
            const int pos = atomicAdd(&hdr_w->namb, 1);
            amb_list[pos] = (int32_t)row;
            amb_lim[pos] = s1 + thr[t] + 2.3841858e-7f * fabsf(s1);   // every code whose f32 score is <= this may be the float64 arg-min
          }
        }
        if (!amb) {
          // gather z_q (rounded to T) for this lane's channel quarter, accumulate squared error
          const float* er = E + (int64_t)code * d + q * kc;
          T* zo = zq_out + row * (int64_t)d + q * kc;
          if (fast) {
#pragma unroll
            for (int s = 0; s < NF * FE; s += DT<T>::VEC) {
              float ev[DT<T>::VEC];
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) ev[e] = to_f32(from_f32<T>(er[s + e]));
              Vec<T>::store(zo + s, ev);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) {
                const float df = lq_get<T, NF>(zt[t], (s + e) / FE, (s + e) % FE) - ev[e];
                sq_acc = fmaf(df, df, sq_acc);
              }
            }
          } else {
#pragma unroll
            for (int s = 0; s < NF * FE; ++s) {
              if (q * kc + s < d) {
                const float ev = to_f32(from_f32<T>(er[s]));
                zo[s] = from_f32<T>(ev);
                const float df = lq_get<T, NF>(zt[t], s / FE, s % FE) - ev;
                sq_acc = fmaf(df, df, sq_acc);
              }
            }
          }
          if (kc == 0) atomicAdd(lds_hist ? &hist[code] : &counts_acc[code], 1);   // integer sums: order-independent, bit-reproducible
        }
      }
    }
  }
  // ---- per-workgroup output: wave partials of the squared error; LDS histogram -> one global add per used code ----
  const float ws_ = wave_sum(sq_acc);
  if (lane == 0) partial[blockIdx.x * NW + wave] = ws_;
  if (lds_hist) {
    __syncthreads();
    for (int k = tid; k < K; k += NW * 64) {
      const int h = hist[k];
      if (h) atomicAdd(&counts_acc[k], h);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Resident-codebook assignment: the whole codebook image fits one LDS chunk (<= 64 KB: K <= 512 at d = 64 bf16), and ONE launch does
// everything, with nothing to zero beforehand:
//   * NO batch bias: score = ||e||^2 - 2 z.e straight from the accumulator; the key (f32 bits & ~127) | index is ordered with the FLOAT
//     v_min_f32 / v_med3_f32, which order positive and negative keys alike (among equal truncated negative scores the larger index
//     wins: irrelevant, equal scores are re-evaluated exactly anyway; NaN keys are ignored and leave the row at +inf = ambiguous);
//     the ambiguity threshold is per row, from the row's own norm;
//   * rows whose two best scores are within the bound are parked in an LDS list in ROW ORDER (ballot ranks + a prefix over the waves:
//     the slot of a row does not depend on timing, so the squared-error sum is bit-reproducible) and resolved by the whole workgroup
//     while the codebook is resident: every wave re-scores the parked rows against its share of the code blocks on the matrix cores and
//     appends the codes under the row's limit to a candidate list; one LANE per candidate evaluates the float64 distance (vq_exact_one:
//     same summation order for every candidate, so duplicate codes tie exactly), LDS integer atomics keep the minimum and then the
//     smallest code among the minima (first index wins);  more than VQ_FAST_ROWS parked rows or more candidates than threads (constructed
//     inputs: every row a tie) take the sequential per-wave path with the same arithmetic;
//   * z_q comes from the LDS image (-0.5 x the packed -2e is exact) on the bf16 fast path;
//   * histogram: integer atomics into the control block of the prepared image; the workgroup that arrives last (ticket behind an
//     agent-scope release / acquire) folds squared-error partials and perplexity, writes stats / counts and ZEROES the control block
//     again for the next call (the first zeroing is vq_prepare_kernel's).  One call in flight per prepared image.
// ---------------------------------------------------------------------------------------------
#define VQ_FAST_ROWS 128     // parked rows per batch the all-wave path handles (typical: < 1 % of the batch's rows)
struct VqCtl { int done; int namb; int pad[62]; };       // 256 bytes, followed by counts_acc[K]

template <typename T>
__device__ __forceinline__ double vq_exact_one(const T* __restrict__ Z, const float* __restrict__ E, int64_t n, int k, int d) {
  double d0, d1;
  vq_exact_pair<T, 4>(Z, E, n, k, -1, d, d0, d1);        // (k1 < 0: the second row folds away; two 16-byte chunks in flight)
  return d0;
}

// The same float64 distance with both operands taken from LDS (bf16 rows with d = 4 q): z from the parked-row copy, e from the packed
// image (e rounded to bf16 = -0.5 x the packed -2e, exact).  Same channel order and arithmetic as vq_exact_pair: same bits.
template <int NF>
__device__ __forceinline__ double vq_exact_lds(const bf16* __restrict__ zrow, const bf16x8* __restrict__ wl, int code, int d) {
  constexpr int q = NF * 8;
  double s0 = 0.0;
  for (int j = 0; j < d; j += 8) {
    const bf16x8 zv = *reinterpret_cast<const bf16x8*>(zrow + j);
    const bf16x8 pv = wl[((code >> 4) * NF + ((j % q) >> 3)) * 64 + (code & 15) + 16 * (j / q)];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double da = (double)(float)zv[e] - (double)(-0.5f * (float)pv[e]);
      s0 = __builtin_fma(da, da, s0);
    }
  }
  return s0;
}

// Workgroup barrier of the batch loop: orders LDS traffic only.  __syncthreads() also drains the vector-memory queue, i.e. the first
// barrier behind the epilogue waited for every z_q / index store of the batch to reach memory before the parked rows were looked at.
#define VQ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define VQ_PZ_CAP 16         // parked rows per batch whose z row is kept in LDS (the usual case: ~0.3-0.7 % of a batch's rows are parked)

template <typename T, int NF, int NT, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 1)) void vq_assign_resident_kernel(
    const T* __restrict__ Z, const float* __restrict__ E, const float* __restrict__ en_g, int64_t N, int K, int d, int Kc,
    int32_t* __restrict__ idx_out, T* __restrict__ zq_out, float* __restrict__ partial /*[grid*NW]*/,
    const typename DT<T>::frag_t* __restrict__ pk, VqCtl* __restrict__ ctl, int32_t* __restrict__ counts_out, float* __restrict__ stats_out) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE;                 // channels per lane quarter
  constexpr int BATCH = NW * NT * 16;        // rows per workgroup batch = capacity of the parked list
  constexpr int CAND_CAP = NW * 64;          // one candidate per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag_t* wl = reinterpret_cast<frag_t*>(smem);                                               // [Kc/16][NF][64]
  float* enl = reinterpret_cast<float*>(smem + (size_t)(Kc / 16) * NF * 64 * sizeof(frag_t)); // [Kc] ||e||^2 (3e38 beyond K)
  int* hist = reinterpret_cast<int*>(enl + Kc);                                               // [K]
  unsigned long long* best = reinterpret_cast<unsigned long long*>(hist + ((K + 1) & ~1));    // [VQ_FAST_ROWS] float64 bits of the minimum
  int* bestk = reinterpret_cast<int*>(best + VQ_FAST_ROWS);                                   // [VQ_FAST_ROWS]
  int* park = bestk + VQ_FAST_ROWS;                                                           // [BATCH] x {row - batch base, limit (f32 bits)}
  int* cand = park + 2 * BATCH;                                                               // [CAND_CAP] (slot << 16) | code
  int* misc = cand + CAND_CAP;                                                                // [0..NW) parked per wave, [16] candidates, [17] last flag, [18..18+NW) f32 maxima
  T* park_z = reinterpret_cast<T*>(misc + 32);                                                // [VQ_PZ_CAP][d] rows of the parked vectors (bf16 fast path only)
  int32_t* counts_acc = reinterpret_cast<int32_t*>(ctl + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 15, kc = lane >> 4;
  const bool fast = (d == 4 * q);
  const int nmb = Kc / 16;
#ifdef VQ_STAMPS
  __shared__ unsigned long long vq_ts[16][8];
  if (tid < 128) (&vq_ts[0][0])[tid] = 0ull;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_prev;
  const unsigned long long w_begin = wall_clock64();               // constant 100 MHz counter: calibrates the s_memtime ticks
#define VQ_ST(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) vq_ts[wave][i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define VQ_ST(i) do { } while (0)
#endif

  // ---- the first batch's rows are requested before the codebook image so that both latencies overlap ----
  const int64_t nbatch = (N + BATCH - 1) / BATCH;
  LQTile<T, NF> zt[NT];
  auto load_batch = [&](int64_t b, int tid_) {
    const int64_t r0 = b * BATCH + (int64_t)(tid_ >> 6) * (NT * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int64_t row = r0 + t * 16 + (tid_ & 15);
      if (row >= N) row = N - 1;
      lq_load<T, NF>(zt[t], Z, row, d, (tid_ >> 4) & 3, fast);
    }
  };
  if ((int64_t)blockIdx.x < nbatch) load_batch(blockIdx.x, tid);
  // ---- once per workgroup: codebook image, norms, their maximum, cleared histogram / lists ----
  copy_frags_lds<T>(wl, pk, nmb * NF * 64, tid, NW * 64);
  float enmax;
  {
    float m = 0.f;
    for (int i = tid; i < Kc; i += NW * 64) {
      const float v = i < K ? en_g[i] : 3.0e38f;
      enl[i] = v;
      if (i < K) m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) misc[18 + wave] = __float_as_int(m);
  }
  for (int k = tid; k < K; k += NW * 64) hist[k] = 0;
  if (tid < VQ_FAST_ROWS) { best[tid] = ~0ull; bestk[tid] = 0x7fffffff; }
  if (tid == 0) misc[16] = 0;
  __syncthreads();
  {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, __int_as_float(misc[18 + w]));
    enmax = m;
  }
  const float err_rel = (float)(4 * q + 8) * 1.1920929e-7f;            // (d_pad+8) * 2^-23: f32 accumulation of exact products
  const float thr_rel = (3.0517578125e-5f + 2.f * err_rel) * 1.001953125f;   // 2^-15: key truncation of both scores, + 2*err, + margin
  const float enroot = sqrtf(enmax);
  float sq_acc = 0.f;
  int n_resolved = 0;
  VQ_ST(2);

  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int64_t base = batch * BATCH;
    const int64_t v0 = base + (int64_t)wave * (NT * 16);
    if (batch != (int64_t)blockIdx.x) {                            // (the first batch is already in flight)
      int tidl = tid;
      asm volatile("" : "+v"(tidl));                               // (no row addresses carried around the loop)
      load_batch(batch, tidl);
    }
    float thr[NT];
    unsigned g1[NT], g2[NT];
    int gc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float zn = 0.f;
#pragma unroll
      for (int s = 0; s < NF; ++s)
#pragma unroll
        for (int e = 0; e < FE; ++e) { const float v = lq_get<T, NF>(zt[t], s, e); zn = fmaf(v, v, zn); }
      zn += __shfl_xor(zn, 16, 64);
      zn += __shfl_xor(zn, 32, 64);
      const float sroot = sqrtf(zn) + enroot;                      // |score| and every partial sum <= (||z|| + ||e||max)^2
      thr[t] = sroot * sroot * thr_rel + 1e-37f;
#pragma unroll
      for (int s = 0; s < NF; ++s) asm volatile("" : "+v"(zt[t].f[s]));   // (else the unpacked floats stay live through the main loop)
      g1[t] = 0x7F800000u; g2[t] = 0x7F800000u; gc[t] = 0;         // +inf
    }
    VQ_ST(0);                                                      // z loads + norms (includes the load latency)
    for (int g0 = 0; g0 < nmb; g0 += 8) {
      unsigned c1[NT], c2[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { c1[t] = 0x7F800000u; c2[t] = 0x7F800000u; }
      const int g1e = (g0 + 8) < nmb ? (g0 + 8) : nmb;
      for (int mb = g0; mb < g1e; ++mb) {
        const f32x4 en4 = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kc);
        frag_t a[NF];
#pragma unroll
        for (int s = 0; s < NF; ++s) a[s] = wl[(mb * NF + s) * 64 + lane];
        const unsigned lidx = (unsigned)((mb - g0) * 16 + 4 * kc);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x4 acc = en4;
#pragma unroll
          for (int s = 0; s < NF; ++s) acc = mfma16(a[s], zt[t].f[s], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned key = (__float_as_uint(acc[r]) & ~VQ_IDX_MASK) | (lidx + r);
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(c2[t]) : "v"(c1[t]), "v"(c2[t]), "v"(key));   // runner-up (c1 <= c2)
            asm("v_min_f32 %0, %1, %2" : "=v"(c1[t]) : "v"(c1[t]), "v"(key));
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {                               // fold the group of 128 codes into the running (best, runner-up, group)
        const float a1 = __uint_as_float(g1[t]), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(c1[t]), b2 = __uint_as_float(c2[t]);
        const float hi = fmaxf(a1, b1), lo2 = fminf(a2, b2);
        g2[t] = __float_as_uint(fminf(hi, lo2));
        if (b1 < a1) { g1[t] = c1[t]; gc[t] = g0; }
      }
    }
    VQ_ST(3);                                                      // MFMA + key / min / med3 main loop
    // ---- min-reduce over the 4 lane groups that share a row; ambiguity test; z_q, squared error, histogram of the settled rows ----
    // (opaque copies of the lane coordinates: keeps the epilogue's address arithmetic out of the main loop's register budget)
    int tide = tid;
    asm volatile("" : "+v"(tide));
    const int vxe = tide & 15, kce = (tide >> 4) & 3;
    const int64_t v0e = base + (int64_t)(tide >> 6) * (NT * 16);
    float lim[NT];
    unsigned ambm = 0u;                                            // bit t: row (t, vxe) is parked (valid in the kce == 0 lanes)
    int nparked = 0, rank[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const unsigned o1 = __shfl_xor(g1[t], off, 64), o2 = __shfl_xor(g2[t], off, 64);
        const int oc = __shfl_xor(gc[t], off, 64);
        const float a1 = __uint_as_float(g1[t]), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(o1), b2 = __uint_as_float(o2);
        const float hi = fmaxf(a1, b1), lo2 = fminf(a2, b2);
        g2[t] = __float_as_uint(fminf(hi, lo2));
        if (b1 < a1 || (b1 == a1 && (oc < gc[t] || (oc == gc[t] && o1 < g1[t])))) { g1[t] = o1; gc[t] = oc; }
      }
      const int64_t row = v0e + t * 16 + vxe;
      const int code = gc[t] * 16 + (int)(g1[t] & VQ_IDX_MASK);
      const float s1 = __uint_as_float(g1[t] & ~VQ_IDX_MASK), s2 = __uint_as_float(g2[t] & ~VQ_IDX_MASK);
      const bool amb = !((s2 - s1) > thr[t]) || (unsigned)code >= (unsigned)K;   // also catches NaN / inf rows
      lim[t] = s1 + thr[t] + 2.3841858e-7f * fabsf(s1);
      const unsigned long long pm = __builtin_amdgcn_ballot_w64(amb && kce == 0 && row < N);
      rank[t] = nparked + __builtin_popcountll(pm & ((1ull << vxe) - 1ull));
      nparked += __builtin_popcountll(pm);
      if (amb) ambm |= 1u << t;
      if (row < N && !amb) {
        T* zo = zq_out + row * (int64_t)d + q * kce;
        if (fast) {
          if constexpr (FE == 8) {                                 // e (rounded to bf16) = -0.5 * the packed -2e: exact
#pragma unroll
            for (int s = 0; s < NF; ++s) {
              const bf16x8 pv = wl[((code >> 4) * NF + s) * 64 + (code & 15) + 16 * kce];
              bf16x8 ev;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float f = -0.5f * (float)pv[e];
                ev[e] = (bf16)f;
                const float df = (float)zt[t].f[s][e] - f;
                sq_acc = fmaf(df, df, sq_acc);
              }
              *reinterpret_cast<bf16x8*>(zo + 8 * s) = ev;
            }
          } else {
            const float* er = E + (int64_t)code * d + q * kce;
#pragma unroll
            for (int s = 0; s < NF * FE; s += DT<T>::VEC) {
              float ev[DT<T>::VEC];
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) ev[e] = to_f32(from_f32<T>(er[s + e]));
              Vec<T>::store(zo + s, ev);
#pragma unroll
              for (int e = 0; e < DT<T>::VEC; ++e) {
                const float df = lq_get<T, NF>(zt[t], (s + e) / FE, (s + e) % FE) - ev[e];
                sq_acc = fmaf(df, df, sq_acc);
              }
            }
          }
        } else {
          const float* er = E + (int64_t)code * d + q * kce;
#pragma unroll
          for (int s = 0; s < NF * FE; ++s) {
            if (q * kce + s < d) {
              const float ev = to_f32(from_f32<T>(er[s]));
              zo[s] = from_f32<T>(ev);
              const float df = lq_get<T, NF>(zt[t], s / FE, s % FE) - ev;
              sq_acc = fmaf(df, df, sq_acc);
            }
          }
        }
        if (kce == 0) { idx_out[row] = code; atomicAdd(&hist[code], 1); }
      }
    }
    VQ_ST(4);                                                      // reduce, ambiguity test, z_q, histogram
    // ---- park the ambiguous rows in row order: slot = (rows parked by the waves before this one) + rank inside the wave ----
    if ((tide & 63) == 0) misc[tide >> 6] = nparked;
    VQ_LDS_BARRIER();
    int pbase = 0, n = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = misc[w]; if (w < (tide >> 6)) pbase += c; n += c; }
    if (n == 0) { VQ_LDS_BARRIER(); continue; }                     // (uniform) nothing to resolve in this batch
    if (kce == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (((ambm >> t) & 1u) && v0e + t * 16 + vxe < N) {
          park[2 * (pbase + rank[t])] = (tide >> 6) * (NT * 16) + t * 16 + vxe;
          park[2 * (pbase + rank[t]) + 1] = __float_as_int(lim[t]);
        }
    }
    // bf16 rows, few of them parked (the usual case): their vectors go to LDS as well, straight from the registers of the four lanes
    // that hold them; the screen, the float64 distances and the outputs below then touch no global memory but the final stores (the
    // global-memory form of this pass was a chain of dependent L2 / HBM latencies behind barriers: ~9 us per batch)
    bool zlds = false;
    if constexpr (FE == 8) {
      zlds = fast && n <= VQ_PZ_CAP;
      if (zlds) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (((ambm >> t) & 1u) && v0e + t * 16 + vxe < N) {
#pragma unroll
            for (int s = 0; s < NF; ++s) *reinterpret_cast<bf16x8*>(park_z + (pbase + rank[t]) * d + q * kce + 8 * s) = zt[t].f[s];
          }
      }
    }
    VQ_LDS_BARRIER();
    bool all_waves = n <= VQ_FAST_ROWS;
    int tidr = tid;
    asm volatile("" : "+v"(tidr));
    if (all_waves) {
      // ---- screen: wave w re-scores every tile of 16 parked rows against its share of the code blocks ----
      const int waver = tidr >> 6, laner = tidr & 63, vxr = tidr & 15, kcr = (tidr >> 4) & 3;
      const int mb0 = (waver * nmb) / NW, mb1 = ((waver + 1) * nmb) / NW;
      for (int t0 = 0; t0 < n; t0 += 16) {
        const int li = t0 + vxr;
        const bool valid = li < n;
        const int64_t row = base + park[2 * (valid ? li : t0)];
        const float lm = __int_as_float(park[2 * (valid ? li : t0) + 1]);
        LQTile<T, NF> zr;
        bool from_lds = false;
        if constexpr (FE == 8) {
          if (zlds) {
            from_lds = true;
#pragma unroll
            for (int s = 0; s < NF; ++s) zr.f[s] = *reinterpret_cast<const bf16x8*>(park_z + (valid ? li : t0) * d + q * kcr + 8 * s);
          }
        }
        if (!from_lds) lq_load<T, NF>(zr, Z, row, d, kcr, fast);
        for (int mb = mb0; mb < mb1; ++mb) {
          f32x4 acc = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kcr);
#pragma unroll
          for (int s = 0; s < NF; ++s) acc = mfma16(wl[(mb * NF + s) * 64 + laner], zr.f[s], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int code = mb * 16 + 4 * kcr + r;
            if (valid && acc[r] <= lm && code < K) {
              const int pos = atomicAdd(&misc[16], 1);
              if (pos < CAND_CAP) cand[pos] = (li << 16) | code;
            }
          }
        }
      }
      VQ_LDS_BARRIER();
      const int ncand = misc[16];
      all_waves = ncand <= CAND_CAP;                               // (uniform)
      if (all_waves) {
        // ---- one lane per candidate: float64 distance; minimum per row, then the smallest code among the minima ----
        const int c = laner * NW + waver;                            // spread the few candidates over the waves
        int li = 0, code = 0;
        unsigned long long bits = ~0ull;
        if (c < ncand) {
          const int cv = cand[c];
          li = cv >> 16; code = cv & 0xffff;
          double dist;
          bool done_ = false;
          if constexpr (FE == 8) {
            if (zlds) { dist = vq_exact_lds<NF>(reinterpret_cast<const bf16*>(park_z) + li * d, reinterpret_cast<const bf16x8*>(wl), code, d); done_ = true; }
          }
          if (!done_) dist = vq_exact_one<T>(Z, E, base + park[2 * li], code, d);
          bits = (unsigned long long)__double_as_longlong(dist);
          __hip_atomic_fetch_min(&best[li], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        VQ_LDS_BARRIER();
        if (c < ncand && bits == best[li]) __hip_atomic_fetch_min(&bestk[li], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        VQ_LDS_BARRIER();
        // ---- outputs of the resolved rows: 16 lanes per row ----
        const int j = tidr & 15;
        for (int li2 = tidr >> 4; li2 < n; li2 += NW * 4) {
          int bk = bestk[li2];
          if ((unsigned)bk >= (unsigned)K) bk = 0;                 // no candidate at all (NaN row): argmin of an all-NaN row is 0
          const int64_t row = base + park[2 * li2];
          for (int ch = j; ch < d; ch += 16) {
            float ev, zv;
            bool done_ = false;
            if constexpr (FE == 8) {
              if (zlds) {
                ev = -0.5f * (float)reinterpret_cast<const bf16x8*>(wl)[((bk >> 4) * NF + ((ch % q) >> 3)) * 64 + (bk & 15) + 16 * (ch / q)][ch & 7];
                zv = (float)reinterpret_cast<const bf16*>(park_z)[li2 * d + ch];
                done_ = true;
              }
            }
            if (!done_) {
              ev = to_f32(from_f32<T>(E[(int64_t)bk * d + ch]));
              zv = to_f32(Z[row * (int64_t)d + ch]);
            }
            zq_out[row * (int64_t)d + ch] = from_f32<T>(ev);
            const float df = zv - ev;
            sq_acc = fmaf(df, df, sq_acc);
          }
          if (j == 0) { idx_out[row] = bk; atomicAdd(&hist[bk], 1); }
        }
        VQ_LDS_BARRIER();
        if (tidr < VQ_FAST_ROWS && tidr < n) {
          int m1 = -1;
          asm volatile("" : "+v"(m1));                              // (materialised here, not carried through the batch loop)
          best[tidr] = ((unsigned long long)(unsigned)m1 << 32) | (unsigned)m1;
          bestk[tidr] = 0x7fffffff;
        }
      }
    }
    if (!all_waves) {
      // ---- many parked rows / candidates (constructed inputs): one wave per tile of 16 parked rows, candidates evaluated in-lane
      // (up to two pending per lane, flushed together); same float64 arithmetic as above ----
      for (int t0 = wave * 16; t0 < n; t0 += NW * 16) {
        const int li = t0 + vx;
        const bool valid = li < n;
        const int64_t row = base + park[2 * (valid ? li : t0)];
        const float lm = __int_as_float(park[2 * (valid ? li : t0) + 1]);
        LQTile<T, NF> zr;
        lq_load<T, NF>(zr, Z, row, d, kc, fast);
        double bestd = 1.0e300;
        int bk = 0x7fffffff, p0 = -1, p1 = -1;
        auto flush = [&]() {
          if (p0 >= 0) {
            double d0, d1;
            vq_exact_pair<T, 1>(Z, E, row, p0, p1, d, d0, d1);
            if (d0 < bestd || (d0 == bestd && p0 < bk)) { bestd = d0; bk = p0; }
            if (p1 >= 0 && (d1 < bestd || (d1 == bestd && p1 < bk))) { bestd = d1; bk = p1; }
          }
          p0 = p1 = -1;
        };
#pragma unroll 4
        for (int mb = 0; mb < nmb; ++mb) {
          f32x4 acc = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kc);
#pragma unroll
          for (int s = 0; s < NF; ++s) acc = mfma16(wl[(mb * NF + s) * 64 + lane], zr.f[s], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int code = mb * 16 + 4 * kc + r;
            const bool hit = acc[r] <= lm && code < K;
            if (__builtin_amdgcn_ballot_w64(hit && p1 >= 0) != 0ull) flush();       // a lane with both slots taken: evaluate all pending
            if (hit) { if (p0 < 0) p0 = code; else p1 = code; }
          }
        }
        flush();
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
          const double ob = __shfl_xor(bestd, off, 64);
          const int okk = __shfl_xor(bk, off, 64);
          if (ob < bestd || (ob == bestd && okk < bk)) { bestd = ob; bk = okk; }
        }
        if ((unsigned)bk >= (unsigned)K) bk = 0;
        if (valid) {
          const float* er = E + (int64_t)bk * d + q * kc;
          T* zo = zq_out + row * (int64_t)d + q * kc;
#pragma unroll
          for (int s = 0; s < NF * FE; ++s) {
            if (q * kc + s < d) {
              const float ev = to_f32(from_f32<T>(er[s]));
              zo[s] = from_f32<T>(ev);
              const float df = lq_get<T, NF>(zr, s / FE, s % FE) - ev;
              sq_acc = fmaf(df, df, sq_acc);
            }
          }
          if (kc == 0) { idx_out[row] = bk; atomicAdd(&hist[bk], 1); }
        }
      }
    }
    n_resolved += n;
    VQ_LDS_BARRIER();
    if (tid == 0) misc[16] = 0;
    VQ_ST(5);                                                      // exact re-evaluation of the parked rows
  }
  // ---- per-workgroup outputs: wave partials of the squared error; histogram by integer atomics (order-independent) ----
  int tidt = tid;                                                   // (opaque: the tail's addresses are computed here, not kept in registers / scratch from the prologue on)
  asm volatile("" : "+v"(tidt));
  const float ws_ = wave_sum(sq_acc);
  if ((tidt & 63) == 0) partial[blockIdx.x * NW + (tidt >> 6)] = ws_;
  __syncthreads();
  for (int k = tidt; k < K; k += NW * 64) {
    const int h = hist[k];
    if (h) atomicAdd(&counts_acc[k], h);
  }
  if (tidt == 0 && n_resolved) atomicAdd(&ctl->namb, n_resolved);
  // publish (partials by plain stores, counts by atomics), then take a ticket; the last workgroup folds everything
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tidt == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(&ctl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    misc[17] = (ticket == (int)gridDim.x - 1) ? 1 : 0;
    if (ticket == (int)gridDim.x - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const bool last = misc[17] != 0;
  __syncthreads();                                                  // (red below may overlap misc when the image is tiny)
  if (last) {
    double* red = reinterpret_cast<double*>(smem);                  // (the codebook fragments are no longer needed)
    const int npartial = (int)gridDim.x * NW;
    double sp = 0.0;
    for (int i = tidt; i < npartial; i += NW * 64) sp += (double)__hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double hp = 0.0;
    for (int k = tidt; k < K; k += NW * 64) {
      const int c = __hip_atomic_load(&counts_acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      counts_out[k] = c;
      __hip_atomic_store(&counts_acc[k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // clean for the next call
      const double p = (double)c / (double)N;
      hp += p * log(p + 1e-10);
    }
    sp = wave_sum_d(sp);
    hp = wave_sum_d(hp);
    if ((tidt & 63) == 0) { red[tidt >> 6] = sp; red[NW + (tidt >> 6)] = hp; }
    __syncthreads();
    if (tidt == 0) {
      for (int w = 1; w < NW; ++w) { red[0] += red[w]; red[NW] += red[NW + w]; }     // fixed order
    }
    if (tidt == 0) {
      stats_out[0] = (float)red[0];
      stats_out[1] = (float)exp(-red[NW]);
      stats_out[2] = (float)__hip_atomic_load(&ctl->namb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stats_out[3] = (float)(red[0] / ((double)N * (double)d));   // mean squared error (the two VQ loss terms)
      __hip_atomic_store(&ctl->namb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef VQ_STAMPS
  VQ_ST(6);                                                        // partials, histogram atomics, ticket, (last workgroup) statistics
  __syncthreads();
  if (tid < 128) vq_dbg[(size_t)blockIdx.x * 128 + tid] = (&vq_ts[0][0])[tid];
  if (tid == 0) { vq_dbg[(size_t)gridDim.x * 128 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_begin; vq_dbg[(size_t)gridDim.x * 128 + 2 * blockIdx.x + 1] = wall_clock64() - w_begin; }
#endif
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the resident-codebook assignment for the measured shape (bf16 rows of d = 64 channels, K <= 512): ONE 16-wave
// workgroup per CU keeps ONE copy of the codebook image and loops over batches of 16 * NT * 16 rows.  Nothing inside the loop waits
// for the workgroup: every wave requests its rows of the NEXT batch before it scores the current ones, its z_q / index stores drain
// behind the next batch's scoring, and a tile that holds ambiguous rows is resolved by the wave that owns it (the rows go to a
// wave-private LDS scratch, the tile is re-scored against the whole image, codes under the row's limit are evaluated in float64
// in-lane: the arithmetic of the resident kernel's sequential path, so indices and z_q are the same bit for bit).  Row results and the
// per-wave squared-error partials depend on the row-to-wave mapping only, never on timing.
// ---------------------------------------------------------------------------------------------
#define VQS_NW 16
#define VQS_ZP 72            // pitch (bf16) of a scratch row: 144 B keeps the 16-byte reads of different rows on different banks
#define VQS_GRP 16           // code blocks per key group: the 7-bit index is (block in group) * 4 + accumulator row, the lane adds its quarter

// scores CNT (FULL: VQS_GRP, compile-time) code blocks from mb0 on against NT row tiles; (c1, c2) = smallest / second smallest key per tile
template <int NT, bool FULL>
__device__ __forceinline__ void vqs_score_group(const bf16x8* __restrict__ wl, const float* __restrict__ enl, int lane, int kc, int mb0, int cnt,
                                                const LQTile<bf16, 2> (&z)[NT], unsigned (&c1)[NT], unsigned (&c2)[NT]) {
  const unsigned maskv = ~VQ_IDX_MASK;
  // (one base per group + constant element offsets: the block's three LDS reads then carry immediate offsets instead of an address
  // computation each)
  const bf16x8* __restrict__ wb = wl + mb0 * 2 * 64 + lane;
  const float* __restrict__ eb = enl + mb0 * 16 + 4 * kc;
  auto block = [&](int mb, int j) {
    (void)mb;
    const f32x4 en4 = *reinterpret_cast<const f32x4*>(eb + j * 16);
    const bf16x8 a0 = wb[j * 2 * 64], a1 = wb[j * 2 * 64 + 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 acc = mfma16(a0, z[t].f[0], en4);
      acc = mfma16(a1, z[t].f[1], acc);
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        // (the keys are made in C, not in the inline assembly: the first reader of a matrix-core result must be an instruction the
        // compiler sees, or the wait states between v_mfma and the read are not inserted)
        const unsigned k0 = (__float_as_uint(acc[r]) & maskv) | (unsigned)(4 * j + r);
        const unsigned k1 = (__float_as_uint(acc[r + 1]) & maskv) | (unsigned)(4 * j + r + 1);
        // two keys per update, 2.5 vector operations per score: with c1 <= c2 the second smallest of {c1, c2, k0, k1} is
        // min(c2, median(c1, k0, k1)); then c1 = min3(c1, k0, k1)
        unsigned m;
        asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(c1[t]), "v"(k0), "v"(k1));
        asm("v_min_f32 %0, %1, %2" : "=v"(c2[t]) : "v"(c2[t]), "v"(m));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(c1[t]) : "v"(c1[t]), "v"(k0), "v"(k1));
      }
    }
  };
  if constexpr (FULL) {
#pragma unroll
    for (int j = 0; j < VQS_GRP; ++j) block(mb0 + j, j);
  } else {
    for (int j = 0; j < cnt; ++j) block(mb0 + j, j);
  }
}

// Row loads of the streaming kernel are issued from inline assembly, so that the compiler's s_waitcnt insertion does not see them: on
// gfx9 loads and stores share vmcnt, and across the batch loop's back edge the compiler's count for "the rows of this batch have
// arrived" also waited for the z_q / index stores the previous batch had just issued (store latency exposed once per batch).  The wait
// is placed by hand instead (vqs_wait): the two loads of a tile retire in issue order, and exactly CNT vector-memory operations (the
// stores of the batch in between) were issued behind the rows being waited for.  The registers are in / out operands of the wait, so no
// use of them can be scheduled above it.
__device__ __forceinline__ void vqs_load_row(LQTile<bf16, 2>& t, const bf16* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t.f[0]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(t.f[1]) : "v"(p) : "memory");
}
template <int NT, int CNT>
__device__ __forceinline__ void vqs_wait(LQTile<bf16, 2> (&b)[NT]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0].f[0]), "+v"(b[0].f[1]) : "n"(CNT) : "memory");
#pragma unroll
  for (int t = 1; t < NT; ++t) asm volatile("" : "+v"(b[t].f[0]), "+v"(b[t].f[1]) : : "memory");
}

// exact arg-min of the rows of one 16-row tile by the wave that holds it: the tile (lane-quarter fragments zt; this lane's row also lies
// in LDS at zrow) is re-scored against the whole image, every code whose score is <= the row's limit is evaluated in float64 in-lane
// (up to two pending per lane, flushed together), smallest code among the minima.  The sequential path of the resident kernel.
__device__ __forceinline__ int vqs_resolve_inlane(const LQTile<bf16, 2>& zt, const bf16* __restrict__ zrow, const bf16x8* __restrict__ wl,
                                                  const float* __restrict__ enl, int nmb, int K, int lane, int kc, bool mine, float lm) {
  double bestd = 1.0e300;
  int bk = 0x7fffffff, p0 = -1, p1 = -1;
  auto flush = [&]() {
    if (p0 >= 0) {
      const double d0 = vq_exact_lds<2>(zrow, wl, p0, 64);
      if (d0 < bestd || (d0 == bestd && p0 < bk)) { bestd = d0; bk = p0; }
      if (p1 >= 0) {
        const double d1 = vq_exact_lds<2>(zrow, wl, p1, 64);
        if (d1 < bestd || (d1 == bestd && p1 < bk)) { bestd = d1; bk = p1; }
      }
    }
    p0 = p1 = -1;
  };
  for (int mb = 0; mb < nmb; ++mb) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kc);
    acc = mfma16(wl[(mb * 2 + 0) * 64 + lane], zt.f[0], acc);
    acc = mfma16(wl[(mb * 2 + 1) * 64 + lane], zt.f[1], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cd = mb * 16 + 4 * kc + r;
      const bool hit = mine && acc[r] <= lm && cd < K;
      if (__builtin_amdgcn_ballot_w64(hit && p1 >= 0) != 0ull) flush();       // a lane with both slots taken: evaluate all pending
      if (hit) { if (p0 < 0) p0 = cd; else p1 = cd; }
    }
  }
  flush();
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const double ob = __shfl_xor(bestd, off, 64);
    const int okk = __shfl_xor(bk, off, 64);
    if (ob < bestd || (ob == bestd && okk < bk)) { bestd = ob; bk = okk; }
  }
  if ((unsigned)bk >= (unsigned)K) bk = 0;                         // no candidate at all (NaN row): argmin of an all-NaN row is 0
  return bk;
}

#define VQS_CAP 8            // ambiguous rows per wave whose vectors are kept in LDS for the joint pass behind the batch loop (16 * 8 = VQ_FAST_ROWS)
#define VQS_CAND (VQS_NW * 64)
// z_q / index stores of the batch loop.  Plain stores: the two 64-byte halves of a row's line come from two instructions and are merged
// in the L2.  Non-temporal stores (-DVQS_NT_STORES) drain while later batches are scored and were 0.7 us faster (44.6 against 45.3 us),
// but each half then travels to HBM on its own: WRITE_SIZE 60.4 MB per launch against 41.7 MB (algorithmic 34.6 MB) -- not kept.
#ifdef VQS_NT_STORES
#define VQS_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define VQS_STORE(p, v) (*(p) = (v))
#endif
template <int NT>
__global__ __launch_bounds__(VQS_NW * 64, 1) void vq_assign_stream_kernel(
    const bf16* __restrict__ Z, const float* __restrict__ en_g, int64_t N, int K, int Kc, int32_t* __restrict__ idx_out,
    bf16* __restrict__ zq_out, float* __restrict__ partial /*[grid*NW]*/, const bf16x8* __restrict__ pk, VqCtl* __restrict__ ctl,
    int32_t* __restrict__ counts_out, float* __restrict__ stats_out, unsigned* __restrict__ ovf_row /*[N]*/, float* __restrict__ ovf_lim /*[N]*/) {
  constexpr int NF = 2, q = 16, d = 64, NW = VQS_NW;
  constexpr int BATCH = NW * NT * 16;
  static_assert(NW * VQS_CAP <= VQ_FAST_ROWS, "parked rows of a workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem);                                               // [Kc/16][NF][64]
  float* enl = reinterpret_cast<float*>(smem + (size_t)(Kc / 16) * NF * 64 * sizeof(bf16x8)); // [Kc] ||e||^2 (3e38 beyond K)
  int* hist = reinterpret_cast<int*>(enl + Kc);                                               // [K]
  int* misc = hist + ((K + 3) & ~3);          // [0] rows resolved, [1] candidates, [17] last flag, [18..18+NW) f32 maxima, [34..34+NW) rows parked per wave
  unsigned long long* best = reinterpret_cast<unsigned long long*>(misc + 64);                // [VQ_FAST_ROWS] float64 bits of the minimum
  int* bestk = reinterpret_cast<int*>(best + VQ_FAST_ROWS);                                   // [VQ_FAST_ROWS]
  int* plist = bestk + VQ_FAST_ROWS;                                                          // [VQ_FAST_ROWS] parked rows in (wave, slot) order -> entry
  unsigned* prow = reinterpret_cast<unsigned*>(plist + VQ_FAST_ROWS);                         // [NW * CAP] row
  float* plim = reinterpret_cast<float*>(prow + NW * VQS_CAP);                                // [NW * CAP] limit
  int* cand = reinterpret_cast<int*>(plim + NW * VQS_CAP);                                    // [VQS_CAND] (list index << 16) | code
  bf16* zdef = reinterpret_cast<bf16*>(cand + VQS_CAND);                                      // [NW * CAP][VQS_ZP] vectors of the parked rows
  bf16* zpark = zdef + NW * VQS_CAP * VQS_ZP;                                                 // [NW][16][VQS_ZP] wave-private tile scratch (overflow path)
  int32_t* counts_acc = reinterpret_cast<int32_t*>(ctl + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 15, kc = lane >> 4;
  const int nmb = Kc / 16;
  const int64_t nbatch = N / BATCH;                                // (the host checks N % BATCH == 0 and launches at most nbatch workgroups)
#ifdef VQ_STAMPS
  __shared__ unsigned long long vq_ts[16][8];                       // (the VQ_ST macro of the resident kernel: phase durations per wave)
  if (tid < 128) (&vq_ts[0][0])[tid] = 0ull;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_prev;
  const unsigned long long w_begin = wall_clock64();
#endif
  LQTile<bf16, NF> bufA[NT], bufB[NT];
  auto load_batch = [&](LQTile<bf16, NF> (&dst)[NT], int64_t b, bool real = true) {
    const int64_t r0 = b * BATCH + (int64_t)wave * (NT * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) vqs_load_row(dst[t], Z + (r0 + (real ? t * 16 + vx : 0)) * d + q * kc);   // (!real: one cached line, see step)
  };
  load_batch(bufA, blockIdx.x);
  // ---- once per workgroup: codebook image, norms, their maximum, cleared histogram / lists ----
  const float env = tid < K ? en_g[tid] : 3.0e38f;                 // (requested before the image so that both arrive in one round trip)
  for (int i0 = 0; i0 < nmb * NF * 64; i0 += 4 * NW * 64) {        // four 16-byte loads in flight per thread (64 KB per pass of the workgroup)
    bf16x8 tmp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * NW * 64 + tid; if (i < nmb * NF * 64) tmp[u] = pk[i]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * NW * 64 + tid; if (i < nmb * NF * 64) wl[i] = tmp[u]; }
  }
  float enmax;
  {
    float m = 0.f;
    if (tid < Kc) {                                                  // (Kc <= VQ_MAX_CHUNK <= threads: one norm per thread, loaded above)
      enl[tid] = env;
      if (tid < K) m = env;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) misc[18 + wave] = __float_as_int(m);
  }
  for (int k = tid; k < K; k += NW * 64) hist[k] = 0;
  if (tid < VQ_FAST_ROWS) { best[tid] = ~0ull; bestk[tid] = 0x7fffffff; }
  if (tid < 2) misc[tid] = 0;
  __syncthreads();
  {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, __int_as_float(misc[18 + w]));
    enmax = m;
  }
  const float err_rel = (float)(4 * q + 8) * 1.1920929e-7f;            // (d_pad+8) * 2^-23: f32 accumulation of exact products
  const float thr_rel = (3.0517578125e-5f + 2.f * err_rel) * 1.001953125f;   // 2^-15: key truncation of both scores, + 2*err, + margin
  const float enroot = sqrtf(enmax);
  float sq_acc = 0.f;
  int n_resolved = 0, cnt_w = 0, ovf_cnt = 0;                      // (wave-uniform) rows re-evaluated / parked in the wave's LDS segment / in its overflow list
  // overflow list of the wave (rows that found the LDS segment full; constructed inputs only): entry k lives in the slot of the wave's k-th
  // row, so a wave touches nothing but the slots of its own rows and the list needs no room beyond the N entries of the workspace
  auto ovf_slot = [&](int k) -> int64_t {
    return ((int64_t)blockIdx.x + (int64_t)(k / (NT * 16)) * gridDim.x) * BATCH + (int64_t)wave * (NT * 16) + (k % (NT * 16));
  };
  bf16* zp = zpark + wave * 16 * VQS_ZP;
  int nstep = 0;                                                   // (wave-uniform) batches behind this wave
  VQ_ST(2);                                                        // image, norms, lists

  // one batch: `cur` holds its rows (requested one batch earlier), `nxt` receives the rows of the batch after it
  auto step = [&](LQTile<bf16, NF> (&cur)[NT], LQTile<bf16, NF> (&nxt)[NT], int64_t batch) {
    const int64_t v0 = batch * BATCH + (int64_t)wave * (NT * 16);
    // the rows of this batch were requested a whole batch ago; behind them only the 3 * NT stores of the previous batch were issued
    // (every vector-memory operation of a step is issued unconditionally; the last step requests one line of its own rows again, a cache hit)
    // issue priority falls with the batches a wave has behind it: the arbiter otherwise favours the oldest wave of a SIMD throughout, the
    // four waves finish a quarter of the loop apart and the last one runs alone, unable to hide its own MFMA / LDS latencies
    { if (nstep == 0) __builtin_amdgcn_s_setprio(3); else if (nstep == 1) __builtin_amdgcn_s_setprio(2); else if (nstep == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    ++nstep;
    vqs_wait<NT, 3 * NT>(cur);
    VQ_ST(0);                                                      // waiting for this batch's rows
    const bool has_next = batch + (int64_t)gridDim.x < nbatch;
    load_batch(nxt, has_next ? batch + gridDim.x : batch, has_next);
    float thr[NT];
    unsigned g1[NT], g2[NT];
    int gc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float zn = 0.f;
#pragma unroll
      for (int s = 0; s < NF; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = (float)cur[t].f[s][e]; zn = fmaf(v, v, zn); }
      zn += __shfl_xor(zn, 16, 64);
      zn += __shfl_xor(zn, 32, 64);
      const float sroot = sqrtf(zn) + enroot;                      // |score| and every partial sum <= (||z|| + ||e||max)^2
      thr[t] = sroot * sroot * thr_rel + 1e-37f;
      g1[t] = 0x7F800000u; g2[t] = 0x7F800000u; gc[t] = 0;         // +inf
    }
    for (int g0 = 0; g0 < nmb; g0 += VQS_GRP) {
      unsigned c1[NT], c2[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { c1[t] = 0x7F800000u; c2[t] = 0x7F800000u; }
      if (g0 + VQS_GRP <= nmb) vqs_score_group<NT, true>(wl, enl, lane, kc, g0, VQS_GRP, cur, c1, c2);
      else vqs_score_group<NT, false>(wl, enl, lane, kc, g0, nmb - g0, cur, c1, c2);
#pragma unroll
      for (int t = 0; t < NT; ++t) {                               // fold the group into the running (best, runner-up, group)
        const float a1 = __uint_as_float(g1[t]), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(c1[t]), b2 = __uint_as_float(c2[t]);
        const float hi = fmaxf(a1, b1), lo2 = fminf(a2, b2);
        g2[t] = __float_as_uint(fminf(hi, lo2));
        if (b1 < a1) { g1[t] = c1[t]; gc[t] = g0; }
      }
    }
    VQ_ST(3);                                                      // norms + scoring
    // ---- min-reduce over the 4 lane groups that share a row (code = block * 16 + 4 * quarter + accumulator row, the quarter is the
    // lane's own); ambiguity test; z_q, squared error, histogram ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int cd1 = (gc[t] + (int)((g1[t] & VQ_IDX_MASK) >> 2)) * 16 + 4 * kc + (int)(g1[t] & 3u);
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const unsigned o1 = __shfl_xor(g1[t], off, 64), o2 = __shfl_xor(g2[t], off, 64);
        const int oc = __shfl_xor(cd1, off, 64);
        const float a1 = __uint_as_float(g1[t] & ~VQ_IDX_MASK), a2 = __uint_as_float(g2[t]), b1 = __uint_as_float(o1 & ~VQ_IDX_MASK), b2 = __uint_as_float(o2);
        const float hi = fmaxf(__uint_as_float(g1[t]), __uint_as_float(o1)), lo2 = fminf(a2, b2);
        g2[t] = __float_as_uint(fminf(hi, lo2));
        if (b1 < a1 || (b1 == a1 && oc < cd1)) { g1[t] = o1; cd1 = oc; }
      }
      const int64_t row = v0 + t * 16 + vx;
      int code = cd1;
      const float s1 = __uint_as_float(g1[t] & ~VQ_IDX_MASK), s2 = __uint_as_float(g2[t] & ~VQ_IDX_MASK);
      const bool amb = !((s2 - s1) > thr[t]) || (unsigned)code >= (unsigned)K;   // also catches NaN / inf rows
      const float lm = s1 + thr[t] + 2.3841858e-7f * fabsf(s1);
      const unsigned long long pm = __builtin_amdgcn_ballot_w64(amb && kc == 0);
      bool parked = false;
      if (pm != 0ull) {                                            // (wave-uniform) this tile holds rows to re-evaluate exactly
        const int np = __builtin_popcountll(pm);
        const bool fits = cnt_w + np <= VQS_CAP;                   // (wave-uniform)
        if (fits) {
          // the usual case (0.3-0.7 % of the rows): vector, row and limit go to the wave's LDS segment in row order, the whole workgroup
          // resolves them behind the batch loop; the provisional z_q / index below are stored all the same (the store count per
          // step stays fixed) and overwritten there
          if (amb) {
            const int e = wave * VQS_CAP + cnt_w + __builtin_popcountll(pm & ((1ull << vx) - 1ull));
            *reinterpret_cast<bf16x8*>(zdef + e * VQS_ZP + q * kc) = cur[t].f[0];
            *reinterpret_cast<bf16x8*>(zdef + e * VQS_ZP + q * kc + 8) = cur[t].f[1];
            if (kc == 0) { prow[e] = (unsigned)row; plim[e] = lm; }
          }
        } else {
          // segment full (constructed inputs: every row a tie): row and limit go to the wave's overflow list in global memory and the wave
          // resolves them on its own behind the loop (the float64 path stays out of the loop's register budget; the two extra stores only
          // add to the vector-memory operations issued behind the pending row loads, which the wait above tolerates)
          if (amb && kc == 0) {
            const int64_t sl_ = ovf_slot(ovf_cnt + __builtin_popcountll(pm & ((1ull << vx) - 1ull)));
            ovf_row[sl_] = (unsigned)row;
            ovf_lim[sl_] = lm;
          }
        }
        cnt_w += fits ? np : 0;                                    // (plain arithmetic on both counters: a store through a selected
        ovf_cnt += fits ? 0 : np;                                  //  address would move them to scratch memory)
        parked = amb;
        n_resolved += np;
      }
      {
        bf16* zo = zq_out + row * (int64_t)d + q * kc;
        float sl = 0.f;
#pragma unroll
        for (int s = 0; s < NF; ++s) {                             // e (rounded to bf16) = -0.5 * the packed -2e: exact
          const bf16x8 pv = wl[((code >> 4) * NF + s) * 64 + (code & 15) + 16 * kc];
          bf16x8 ev;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = -0.5f * (float)pv[e];
            ev[e] = (bf16)f;
            const float df = (float)cur[t].f[s][e] - f;
            sl = fmaf(df, df, sl);
          }
          VQS_STORE(reinterpret_cast<bf16x8*>(zo + 8 * s), ev);
        }
        VQS_STORE(idx_out + row, code);                            // (the four lanes of a row store the same word)
        if (!parked) {
          sq_acc += sl;
          if (kc == 0) atomicAdd(&hist[code], 1);
        }
      }
    }
    VQ_ST(4);                                                      // reduce, ambiguity test, z_q, histogram
  };
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += 2 * (int64_t)gridDim.x) {
    step(bufA, bufB, batch);
    if (batch + (int64_t)gridDim.x < nbatch) step(bufB, bufA, batch + gridDim.x);
  }
  // the last step's (redundant) row loads are still in flight and the compiler does not know it: both buffers stay live up to this wait,
  // so their registers cannot be handed to the tail's variables before the loads have landed.  It also orders the provisional stores of
  // the parked rows before the final ones below.
  __builtin_amdgcn_s_setprio(0);
  vqs_wait<NT, 0>(bufA);
  vqs_wait<NT, 0>(bufB);
  VQ_ST(1);                                                        // store drain behind the last batch
  // ---- the parked rows of the whole workgroup (<= 8 per wave, in (wave, slot) = row order): every wave re-scores them against its share
  // of the code blocks, one lane per candidate evaluates the float64 distance, LDS integer atomics keep the minimum and then the
  // smallest code among the minima -- the all-wave pass of the resident kernel ----
  if (lane == 0) misc[34 + wave] = cnt_w;
  __syncthreads();
  int pbase = 0, npark = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const int c = misc[34 + w]; if (w < wave) pbase += c; npark += c; }
  if (npark > 0) {                                                  // (uniform)
    if (lane < cnt_w) plist[pbase + lane] = wave * VQS_CAP + lane;
    __syncthreads();
    const int mb0 = (wave * nmb) / NW, mb1 = ((wave + 1) * nmb) / NW;
    for (int t0 = 0; t0 < npark; t0 += 16) {
      const int li = t0 + vx;
      const bool valid = li < npark;
      const int e = plist[valid ? li : t0];
      const float lm = plim[e];
      LQTile<bf16, NF> zr;
      zr.f[0] = *reinterpret_cast<const bf16x8*>(zdef + e * VQS_ZP + q * kc);
      zr.f[1] = *reinterpret_cast<const bf16x8*>(zdef + e * VQS_ZP + q * kc + 8);
      for (int mb = mb0; mb < mb1; ++mb) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(enl + mb * 16 + 4 * kc);
        acc = mfma16(wl[(mb * NF + 0) * 64 + lane], zr.f[0], acc);
        acc = mfma16(wl[(mb * NF + 1) * 64 + lane], zr.f[1], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cd = mb * 16 + 4 * kc + r;
          if (valid && acc[r] <= lm && cd < K) {
            const int pos = atomicAdd(&misc[1], 1);
            if (pos < VQS_CAND) cand[pos] = (li << 16) | cd;
          }
        }
      }
    }
    __syncthreads();
    const int ncand = misc[1];
    if (ncand <= VQS_CAND) {                                         // (uniform)
      int li = 0, cd = 0;
      unsigned long long bits = ~0ull;
      if (tid < ncand) {
        const int cv = cand[tid];
        li = cv >> 16; cd = cv & 0xffff;
        bits = (unsigned long long)__double_as_longlong(vq_exact_lds<NF>(zdef + plist[li] * VQS_ZP, wl, cd, d));
        __hip_atomic_fetch_min(&best[li], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
      if (tid < ncand && bits == best[li]) __hip_atomic_fetch_min(&bestk[li], cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
      const int j = tid & 15;                                        // outputs of the resolved rows: 16 lanes per row
      for (int li2 = tid >> 4; li2 < npark; li2 += NW * 4) {
        int bk = bestk[li2];
        if ((unsigned)bk >= (unsigned)K) bk = 0;                     // no candidate at all (NaN row): argmin of an all-NaN row is 0
        const int e = plist[li2];
        const int64_t row = (int64_t)prow[e];
        for (int ch = j; ch < d; ch += 16) {
          const float ev = -0.5f * (float)wl[((bk >> 4) * NF + ((ch % q) >> 3)) * 64 + (bk & 15) + 16 * (ch / q)][ch & 7];
          const float zv = (float)zdef[e * VQS_ZP + ch];
          zq_out[row * (int64_t)d + ch] = (bf16)ev;
          const float df = zv - ev;
          sq_acc = fmaf(df, df, sq_acc);
        }
        if (j == 0) { idx_out[row] = bk; atomicAdd(&hist[bk], 1); }
      }
    } else {
      // more candidates than threads (constructed inputs): one wave per tile of 16 parked rows, candidates evaluated in-lane
      for (int t0 = wave * 16; t0 < npark; t0 += NW * 16) {
        const int li = t0 + vx;
        const bool valid = li < npark;
        const int e = plist[valid ? li : t0];
        LQTile<bf16, NF> zr;
        zr.f[0] = *reinterpret_cast<const bf16x8*>(zdef + e * VQS_ZP + q * kc);
        zr.f[1] = *reinterpret_cast<const bf16x8*>(zdef + e * VQS_ZP + q * kc + 8);
        const int bk = vqs_resolve_inlane(zr, zdef + e * VQS_ZP, wl, enl, nmb, K, lane, kc, valid, plim[e]);
        if (valid) {
          const int64_t row = (int64_t)prow[e];
          bf16* zo = zq_out + row * (int64_t)d + q * kc;
#pragma unroll
          for (int s = 0; s < NF; ++s) {
            const bf16x8 pv = wl[((bk >> 4) * NF + s) * 64 + (bk & 15) + 16 * kc];
            bf16x8 ev;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) {
              const float f = -0.5f * (float)pv[e2];
              ev[e2] = (bf16)f;
              const float df = (float)zr.f[s][e2] - f;
              sq_acc = fmaf(df, df, sq_acc);
            }
            *reinterpret_cast<bf16x8*>(zo + 8 * s) = ev;
          }
          if (kc == 0) { idx_out[row] = bk; atomicAdd(&hist[bk], 1); }
        }
      }
    }
  }
  // ---- the wave's overflow list, 16 rows at a time: vectors back from global memory (and into the wave's LDS scratch for the float64
  // distances), the tile re-scored against the whole image, candidates evaluated in-lane ----
  for (int t0 = 0; t0 < ovf_cnt; t0 += 16) {                        // (wave-uniform trip count)
    const int kq = t0 + vx;
    const bool valid = kq < ovf_cnt;
    const int64_t sl_ = ovf_slot(valid ? kq : t0);
    const int64_t row = (int64_t)__hip_atomic_load(&ovf_row[sl_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by this wave; past the L1)
    const float lm = __hip_atomic_load(&ovf_lim[sl_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    LQTile<bf16, NF> zr;
    zr.f[0] = *reinterpret_cast<const bf16x8*>(Z + row * d + q * kc);
    zr.f[1] = *reinterpret_cast<const bf16x8*>(Z + row * d + q * kc + 8);
    *reinterpret_cast<bf16x8*>(zp + vx * VQS_ZP + q * kc) = zr.f[0];
    *reinterpret_cast<bf16x8*>(zp + vx * VQS_ZP + q * kc + 8) = zr.f[1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-private scratch: LDS operations of one wave complete in order
    const int bk = vqs_resolve_inlane(zr, zp + vx * VQS_ZP, wl, enl, nmb, K, lane, kc, valid, lm);
    if (valid) {
      bf16* zo = zq_out + row * (int64_t)d + q * kc;
#pragma unroll
      for (int s = 0; s < NF; ++s) {
        const bf16x8 pv = wl[((bk >> 4) * NF + s) * 64 + (bk & 15) + 16 * kc];
        bf16x8 ev;
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const float f = -0.5f * (float)pv[e2];
          ev[e2] = (bf16)f;
          const float df = (float)zr.f[s][e2] - f;
          sq_acc = fmaf(df, df, sq_acc);
        }
        *reinterpret_cast<bf16x8*>(zo + 8 * s) = ev;
      }
      if (kc == 0) { idx_out[row] = bk; atomicAdd(&hist[bk], 1); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the scratch is rewritten by the next tile)
  }
  VQ_ST(5);                                                        // joint pass over the parked rows, overflow list
  // ---- per-workgroup outputs: wave partials of the squared error; histogram by integer atomics (order-independent) ----
  const float ws_ = wave_sum(sq_acc);
  if (lane == 0) {
    // (device-scope store: written through, so the arrival ticket below needs no L2 write-back in front of it)
    __hip_atomic_store(&partial[blockIdx.x * NW + wave], ws_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_resolved) atomicAdd(&misc[0], n_resolved);
  }
  __syncthreads();
  for (int k = tid; k < K; k += NW * 64) {
    const int h = hist[k];
    if (h) atomicAdd(&counts_acc[k], h);
  }
  if (tid == 0 && misc[0]) atomicAdd(&ctl->namb, misc[0]);
  // publish, then take a ticket; the last workgroup folds everything.  What the last workgroup reads -- the partials and the histogram
  // accumulator -- was written by device-scope stores / atomics and every wave has waited for their completion in front of the barrier,
  // so the ticket is not preceded by a release fence (an L2 write-back of all the z_q lines the XCD still holds: one more device-scope
  // round trip on the tail of every workgroup); the outputs themselves become visible at the end of the kernel as always
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(&ctl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    misc[17] = (ticket == (int)gridDim.x - 1) ? 1 : 0;
    if (ticket == (int)gridDim.x - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const bool last = misc[17] != 0;
  __syncthreads();
  if (last) {
    double* red = reinterpret_cast<double*>(smem);                  // (the codebook fragments are no longer needed)
    const int npartial = (int)gridDim.x * NW;
    double sp = 0.0;
    for (int i = tid; i < npartial; i += NW * 64) sp += (double)__hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double hp = 0.0;
    for (int k = tid; k < K; k += NW * 64) {
      const int c = __hip_atomic_load(&counts_acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      counts_out[k] = c;
      __hip_atomic_store(&counts_acc[k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // clean for the next call
      const double p = (double)c / (double)N;
      hp += p * log(p + 1e-10);
    }
    sp = wave_sum_d(sp);
    hp = wave_sum_d(hp);
    if (lane == 0) { red[wave] = sp; red[NW + wave] = hp; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NW; ++w) { red[0] += red[w]; red[NW] += red[NW + w]; }     // fixed order
      stats_out[0] = (float)red[0];
      stats_out[1] = (float)exp(-red[NW]);
      stats_out[2] = (float)__hip_atomic_load(&ctl->namb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stats_out[3] = (float)(red[0] / ((double)N * (double)d));   // mean squared error (the two VQ loss terms)
      __hip_atomic_store(&ctl->namb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef VQ_STAMPS
  VQ_ST(6);                                                        // partials, histogram atomics, ticket, (last workgroup) statistics
  __syncthreads();
  if (tid < 128) vq_dbg[(size_t)blockIdx.x * 128 + tid] = (&vq_ts[0][0])[tid];
  if (tid == 0) { vq_dbg[(size_t)gridDim.x * 128 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_begin; vq_dbg[(size_t)gridDim.x * 128 + 2 * blockIdx.x + 1] = wall_clock64() - w_begin; }
#endif
}

// ---------------------------------------------------------------------------------------------
// Exact re-evaluation of the flagged rows of the multi-chunk path, 16 rows per workgroup (8 waves, each with its share of the code
// blocks, merged in wave order): the rows are re-scored against the WHOLE packed
// image on the matrix cores (fragments streamed from L2: 2 MB per tile at K = 8192, d = 128, instead of the 4 MB float32 codebook per
// ROW of the former one-wave-per-row kernels -- 17 ms at 1 % flagged rows), codes under the row's limit are evaluated in float64
// in-lane (up to two pending per lane, flushed together), first index wins ties.  Row results do not depend on the list order; the
// squared-error sum uses a float64 atomic, the histogram integer atomics.
// ---------------------------------------------------------------------------------------------
#define VQ_FIXT_WAVES 8      // waves per tile of 16 flagged rows: each scores its share of the code blocks
template <typename T, int NF>
__global__ __launch_bounds__(64 * VQ_FIXT_WAVES) void vq_fixup_tile_kernel(const T* __restrict__ Z, const float* __restrict__ E, const float* __restrict__ en_g,
                                                            const typename DT<T>::frag_t* __restrict__ pk, int K, int kpad, int d,
                                                            const int32_t* __restrict__ amb_list, const float* __restrict__ amb_lim,
                                                            VqHeader* __restrict__ hdr, int32_t* __restrict__ idx_out, T* __restrict__ zq_out,
                                                            int32_t* __restrict__ counts_fix) {
  constexpr int FE = DT<T>::FE;
  constexpr int q = NF * FE;
  __shared__ double mrg_d[VQ_FIXT_WAVES][16];
  __shared__ int mrg_k[VQ_FIXT_WAVES][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vx = lane & 15, kc = lane >> 4;
  const bool fast = (d == 4 * q);
  const int namb = hdr->namb;
  const int ntile = (namb + 15) >> 4, nmb = kpad >> 4;
  const int mb0 = (wave * nmb) / VQ_FIXT_WAVES, mb1 = ((wave + 1) * nmb) / VQ_FIXT_WAVES;
  float sq_acc = 0.f;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int li = tile * 16 + vx;
    const bool valid = li < namb;
    const int64_t row = amb_list[valid ? li : tile * 16];
    const float lm = amb_lim[valid ? li : tile * 16];
    LQTile<T, NF> zr;
    lq_load<T, NF>(zr, Z, row, d, kc, fast);
    double bestd = 1.0e300;
    int bk = 0x7fffffff, p0 = -1, p1 = -1;
    auto flush = [&]() {
      if (p0 >= 0) {
        double d0, d1;
        vq_exact_pair<T, 1>(Z, E, row, p0, p1, d, d0, d1);
        if (d0 < bestd || (d0 == bestd && p0 < bk)) { bestd = d0; bk = p0; }
        if (p1 >= 0 && (d1 < bestd || (d1 == bestd && p1 < bk))) { bestd = d1; bk = p1; }
      }
      p0 = p1 = -1;
    };
#pragma unroll 2
    for (int mb = mb0; mb < mb1; ++mb) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(en_g + mb * 16 + 4 * kc);
#pragma unroll
      for (int s = 0; s < NF; ++s) acc = mfma16(pk[(size_t)(mb * NF + s) * 64 + lane], zr.f[s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int code = mb * 16 + 4 * kc + r;
        const bool hit = valid && acc[r] <= lm && code < K;
        if (__builtin_amdgcn_ballot_w64(hit && p1 >= 0) != 0ull) flush();       // a lane with both slots taken: evaluate all pending
        if (hit) { if (p0 < 0) p0 = code; else p1 = code; }
      }
    }
    flush();
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const double ob = __shfl_xor(bestd, off, 64);
      const int okk = __shfl_xor(bk, off, 64);
      if (ob < bestd || (ob == bestd && okk < bk)) { bestd = ob; bk = okk; }
    }
    // the waves' shares are merged in wave order: smallest distance, then smallest code (= first index over the whole codebook)
    if (kc == 0) { mrg_d[wave][vx] = bestd; mrg_k[wave][vx] = bk; }
    __syncthreads();
    if (wave == 0) {
      bestd = mrg_d[0][vx]; bk = mrg_k[0][vx];
#pragma unroll
      for (int w = 1; w < VQ_FIXT_WAVES; ++w) {
        const double ob = mrg_d[w][vx];
        const int okk = mrg_k[w][vx];
        if (ob < bestd || (ob == bestd && okk < bk)) { bestd = ob; bk = okk; }
      }
      if ((unsigned)bk >= (unsigned)K) bk = 0;                       // no candidate at all (NaN row): argmin of an all-NaN row is 0
      if (valid) {
        const float* er = E + (int64_t)bk * d + q * kc;
        T* zo = zq_out + row * (int64_t)d + q * kc;
#pragma unroll
        for (int s = 0; s < NF * FE; ++s) {
          if (q * kc + s < d) {
            const float ev = to_f32(from_f32<T>(er[s]));
            zo[s] = from_f32<T>(ev);
            const float df = lq_get<T, NF>(zr, s / FE, s % FE) - ev;
            sq_acc = fmaf(df, df, sq_acc);
          }
        }
        if (kc == 0) { idx_out[row] = bk; atomicAdd(&counts_fix[bk], 1); }
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
    const float ws_ = wave_sum(sq_acc);
    if (lane == 0 && ws_ != 0.f) atomicAdd(&hdr->sq_fix, (double)ws_);
  }
}

// stats = {sqerr_sum, perplexity, n_re-evaluated, 0}
__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ partial, int npartial, const VqHeader* __restrict__ hdr,
                                                          const int32_t* __restrict__ counts_acc, int32_t* __restrict__ counts, int K, int64_t N,
                                                          int d, float* __restrict__ stats) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < npartial; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const double sq = red[0] + hdr->sq_fix;
  __syncthreads();
  double h = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    const int cnt = counts_acc[k];
    counts[k] = cnt;
    const double p = (double)cnt / (double)N;
    h += p * log(p + 1e-10);
  }
  red[threadIdx.x] = h;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    stats[0] = (float)sq;
    stats[1] = (float)exp(-red[0]);
    stats[2] = (float)hdr->namb;
    stats[3] = (float)(sq / ((double)N * (double)d));
  }
}

// ---------------------------------------------------------------------------------------------
// backward: g_z = g_out + cz * (z - e_idx) ; code sums S_k = sum_{n: idx=k} z_n (LDS f32 accumulation)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vq_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ Z,
                                                     const float* __restrict__ E, const int32_t* __restrict__ idx,
                                                     const float* __restrict__ gscale, float cz_base, int64_t N, int K, int d,
                                                     int Kc, int64_t rows_per_wg, T* __restrict__ gz,
                                                     float* __restrict__ slab /*[grid][K][d]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = reinterpret_cast<float*>(smem);        // [Kc][d]
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  int64_t r1 = r0 + rows_per_wg;
  if (r1 > N) r1 = N;
  const float cz = cz_base * (gscale ? gscale[0] : 1.f);
  constexpr int VEC = DT<T>::VEC;
  const bool fast = (d % VEC) == 0;
  const int vpr = (d + VEC - 1) / VEC;                // vectors per row
  const int nchunks = (K + Kc - 1) / Kc;
  for (int c = 0; c < nchunks; ++c) {
    const int kbase = c * Kc;
    __syncthreads();
    for (int i = tid; i < Kc * d; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int64_t nvec = (r1 > r0 ? (r1 - r0) : 0) * vpr;
    for (int64_t i = tid; i < nvec; i += 256) {
      const int64_t n = r0 + i / vpr;
      const int c0 = (int)(i % vpr) * VEC;
      const int k = idx[n];
      const bool mine = (k >= kbase && k < kbase + Kc);
      if (c == 0 || mine) {
        float zv[VEC];
        if (fast) Vec<T>::load(Z + n * (int64_t)d + c0, zv);
        else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) zv[e] = (c0 + e < d) ? to_f32(Z[n * (int64_t)d + c0 + e]) : 0.f;
        }
        if (mine) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (c0 + e < d) atomicAdd(&acc[(k - kbase) * d + c0 + e], zv[e]);
        }
        if (c == 0 && gz != nullptr) {
          float gv[VEC], ov[VEC];
          if (gout != nullptr) {
            if (fast) Vec<T>::load(gout + n * (int64_t)d + c0, gv);
            else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) gv[e] = (c0 + e < d) ? to_f32(gout[n * (int64_t)d + c0 + e]) : 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) gv[e] = 0.f;
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float ev = (c0 + e < d) ? to_f32(from_f32<T>(E[(int64_t)k * d + c0 + e])) : 0.f;
            ov[e] = gv[e] + cz * (zv[e] - ev);
          }
          if (fast) Vec<T>::store(gz + n * (int64_t)d + c0, ov);
          else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
              if (c0 + e < d) gz[n * (int64_t)d + c0 + e] = from_f32<T>(ov[e]);
          }
        }
      }
    }
    __syncthreads();
    const int kn = (K - kbase) < Kc ? (K - kbase) : Kc;
    float* dst = slab + ((int64_t)blockIdx.x * K + kbase) * d;
    for (int i = tid; i < kn * d; i += 256) dst[i] = acc[i];
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 backward on the matrix cores: code sums S[k][:] = sum_{n: idx_n = k} z_n as the GEMM  onehot(idx)^T * Z  with K = rows.
// The one-hot A fragments are generated in registers from the staged indices (exact 0/1 in bf16), the z tile is fetched
// k-strided with ds_read_b64_tr_b16, products are exact and accumulated in f32 in a fixed order -> bit-reproducible,
// no LDS float atomics.  g_z = g_out + cz (z - e_idx) is fused on the staged tile.
// Each of the NWV = 8 waves (two per SIMD: LDS and matrix-pipe latencies of one hide behind the other) owns RB row blocks (16 codes
// each) x CB column blocks (16 channels) of the [Kc x d] chunk; chunks of the
// codebook are an outer loop (z is re-read once per chunk; one chunk covers K <= 512 at d <= 64).
// ---------------------------------------------------------------------------------------------
template <int RB, int CB, int NWV>
__global__ __launch_bounds__(64 * NWV) void vq_bwd_mfma_kernel(const bf16* __restrict__ gout, const bf16* __restrict__ Z, const bf16* __restrict__ ZQ,
                                                          const float* __restrict__ E,
                                                          const int32_t* __restrict__ idx, const float* __restrict__ gscale, float cz_base,
                                                          int64_t N, int K, int d, int64_t rows_per_wg, bf16* __restrict__ gz,
                                                          float* __restrict__ slab /*[grid][K][d]*/) {
  constexpr int DP = CB * 16, PITCH = DP + 8, KC = RB * NWV * 16, NTH = 64 * NWV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* zt = reinterpret_cast<bf16*>(smem);                      // [64][PITCH]
  int* it = reinterpret_cast<int*>(zt + 64 * PITCH);             // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kc = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  int64_t r1 = r0 + rows_per_wg;
  if (r1 > N) r1 = N;
  const float cz = cz_base * (gscale ? gscale[0] : 1.f);
  const bool fast = (d % 8) == 0;
  const int vpr = DP / 8;
  const bf16 one = (bf16)1.f, zero = (bf16)0.f;
  for (int kbase = 0; kbase < K; kbase += KC) {
    f32x4 acc[RB][CB];
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // register image of the NEXT 64-row tile (16-byte channel vectors): z, and on the first chunk g_out and z_q for the fused
    // g_z = g_out + cz (z - z_q); requested behind the MFMAs of the current tile
    constexpr int NV = (64 * (DP / 8) + NTH - 1) / NTH;
    const bool pre = fast && (ZQ != nullptr || gz == nullptr || kbase != 0);      // prefetch path: no codebook gather needed
    const bool want_gz = (kbase == 0 && gz != nullptr);
    bf16x8 rz[NV], rg[NV], rq[NV];
    auto fetch = [&](int64_t p0) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = tid + u * NTH;
        const int row = i / vpr, c0 = (i % vpr) * 8;
        const int64_t n = p0 + row;
        const bool ok = i < 64 * vpr && n < r1 && c0 < d;
        const int64_t off = ok ? n * (int64_t)d + c0 : 0;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(Z + off);
        if (!ok) v = bf16x8{};
        rz[u] = v;
        if (want_gz) {
          rg[u] = gout != nullptr ? *reinterpret_cast<const bf16x8*>(gout + off) : bf16x8{};
          rq[u] = *reinterpret_cast<const bf16x8*>(ZQ + off);
        }
      }
    };
    if (pre && r0 < r1) fetch(r0);
    for (int64_t p0 = r0; p0 < r1; p0 += 64) {
      __syncthreads();
      if (pre) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const int i = tid + u * NTH;
          if (i >= 64 * vpr) continue;
          const int row = i / vpr, c0 = (i % vpr) * 8;
          const int64_t n = p0 + row;
          if (want_gz && n < r1 && c0 < d) {
            float ov[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (float)rg[u][e] + cz * ((float)rz[u][e] - (float)rq[u][e]);
            Vec<bf16>::store(gz + n * (int64_t)d + c0, ov);
          }
          *reinterpret_cast<bf16x8*>(zt + row * PITCH + c0) = rz[u];
        }
      } else {
      // stage 64 rows of z (zero padded) and their indices; first chunk also writes g_z
      for (int i = tid; i < 64 * vpr; i += NTH) {
        const int row = i / vpr, c0 = (i % vpr) * 8;
        const int64_t n = p0 + row;
        float zv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) zv[e] = 0.f;
        if (n < r1 && c0 < d) {
          if (fast) Vec<bf16>::load(Z + n * (int64_t)d + c0, zv);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e < d) zv[e] = (float)Z[n * (int64_t)d + c0 + e];
          }
          if (kbase == 0 && gz != nullptr) {
            const int k = idx[n];
            float gv[8], ov[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = 0.f;
            if (gout != nullptr) {
              if (fast) Vec<bf16>::load(gout + n * (int64_t)d + c0, gv);
              else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (c0 + e < d) gv[e] = (float)gout[n * (int64_t)d + c0 + e];
              }
            }
            float qv[8];
            if (ZQ != nullptr && fast) Vec<bf16>::load(ZQ + n * (int64_t)d + c0, qv);     // coalesced z_q instead of a codebook gather
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) qv[e] = (c0 + e < d) ? (float)(bf16)E[(int64_t)k * d + c0 + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = gv[e] + cz * (zv[e] - qv[e]);
            if (fast) Vec<bf16>::store(gz + n * (int64_t)d + c0, ov);
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (c0 + e < d) gz[n * (int64_t)d + c0 + e] = (bf16)ov[e];
            }
          }
        }
        Vec<bf16>::store(zt + row * PITCH + c0, zv);
      }
      }
      if (tid < 64) it[tid] = (p0 + tid < r1) ? idx[p0 + tid] - kbase : -1;
      __syncthreads();
      if (pre && p0 + 64 < r1) fetch(p0 + 64);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pix0 = ks * 32 + 8 * kc;
        int id8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) id8[j] = it[pix0 + j];
        bf16x8 bfr[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
          const bf16* a0 = zt + (pix0 + (r16 >> 2)) * PITCH + b * 16 + 4 * (r16 & 3);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * PITCH));
          bfr[b] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int a = 0; a < RB; ++a) {
          const int code = (wave * RB + a) * 16 + r16;            // chunk-local code of this lane's A row
          bf16x8 af;
#pragma unroll
          for (int j = 0; j < 8; ++j) af[j] = (id8[j] == code) ? one : zero;
#pragma unroll
          for (int b = 0; b < CB; ++b) acc[a][b] = mfma16(af, bfr[b], acc[a][b]);
        }
      }
    }
    float* dst = slab + (int64_t)blockIdx.x * K * d;
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kbase + (wave * RB + a) * 16 + kc * 4 + r, c = b * 16 + r16;
          if (k < K && c < d) dst[(int64_t)k * d + c] = acc[a][b][r];
        }
  }
}


// g_E[k][j] = ce * gscale * (n_k * e_k[j] - sum_wg slab[wg][k][j]);  also exports code sums when asked
template <typename T>
__global__ void vq_code_reduce_kernel(const float* __restrict__ slab, int nslab, const float* __restrict__ E,
                                      const int32_t* __restrict__ counts, const float* __restrict__ gscale, float ce_base,
                                      int K, int d, float* __restrict__ gE, float* __restrict__ sums_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)K * d) return;
  float s = 0.f;
  for (int w = 0; w < nslab; ++w) s += slab[(int64_t)w * K * d + i];
  if (sums_out) sums_out[i] = s;
  if (gE) {
    const float ce = ce_base * (gscale ? gscale[1] : 1.f);
    const float ev = to_f32(from_f32<T>(E[i]));
    gE[i] = ce * ((float)counts[i / d] * ev - s);
  }
}


// EMA update (scripts/train_vqvae.py:412-414): N_k, m_k moving averages + Laplace-smoothed codebook
// ok (optional device float): <= 0 leaves the running averages and the codebook untouched (isfinite guard of the train step)
__global__ __launch_bounds__(256) void vq_ema_kernel(const float* __restrict__ sums, const int32_t* __restrict__ counts, int K, int d,
                                                     float decay, float eps, float* __restrict__ ema_count,
                                                     float* __restrict__ ema_sum, float* __restrict__ E, const float* __restrict__ ok) {
  __shared__ double red[256];
  if (ok != nullptr && !(ok[0] > 0.f)) return;
  double s = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float nc = decay * ema_count[k] + (1.f - decay) * (float)counts[k];
    ema_count[k] = nc;
    s += (double)nc;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float n = (float)red[0];
  for (int i = threadIdx.x; i < K * d; i += 256) {
    const int k = i / d;
    const float ms = decay * ema_sum[i] + (1.f - decay) * sums[i];
    ema_sum[i] = ms;
    const float smoothed = (ema_count[k] + eps) / (n + (float)K * eps) * n;
    E[i] = ms / smoothed;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int vq_chunk(int K, int d_pad, size_t esize, int nw) {
  (void)nw;                                                  // <= 64 KB of codes so that two workgroups fit a CU
  const size_t cap = 64 * 1024;
  int kc = VQ_MAX_CHUNK;
  while (kc > 16 && (size_t)kc * d_pad * esize > cap) kc >>= 1;
  const int kpad = (K + 15) / 16 * 16;
  if (kc > kpad) kc = kpad;
  return kc;
}
// Large batches (d <= 64): 8-wave workgroups, two per CU (four waves per SIMD, two 16-vector tiles per wave inside the 128-register
// budget); the two workgroups of a CU drift apart, so one streams z / writes z_q while the other is in its MFMA loop.  Else 4 waves x 4 tiles.
static int vq_waves(int64_t N, int d) { return (d <= 64 && N >= (int64_t)256 * 8 * 2 * 16) ? 8 : 4; }
static int vq_grid(int64_t N, int d) {
  const int nw = vq_waves(N, d), NT = nw == 8 ? 2 : 4;
  const int64_t nb = (N + nw * NT * 16 - 1) / (nw * NT * 16);
  const int cap = 512;
  return (int)(nb < cap ? (nb < 1 ? 1 : nb) : cap);
}
#define VQ_FIX_WAVES 1024
#define VQ_BWD_WGS 256

// "prepared codebook": [||e||^2: kpad floats (3e38 beyond K)][packed -2e fragments of kpad codes]; kpad = K rounded up to whole chunks
struct VqPrep { size_t ctl, en, pack, total; int Kc, kpad, npk; };
template <typename T, int NF>
static VqPrep vq_prep_layout(int64_t N, int K, int d) {
  VqPrep P;
  const int d_pad = NF * DT<T>::FE * 4;
  P.Kc = vq_chunk(K, d_pad, sizeof(T), vq_waves(N, d));
  P.kpad = (K + P.Kc - 1) / P.Kc * P.Kc;
  P.npk = (P.kpad / 16) * NF * 64;
  P.ctl = 0;                                                        // VqCtl + counts_acc[K] (kpad >= 64 always holds the 64 header ints)
  P.en = (sizeof(VqCtl) + (size_t)K * 4 + 255) / 256 * 256;
  P.pack = P.en + ((size_t)P.kpad * 4 + 255) / 256 * 256;
  P.total = P.pack + (size_t)P.npk * sizeof(typename DT<T>::frag_t);
  return P;
}
static size_t vq_prep_bytes_max(int K) {
  return (sizeof(VqCtl) + (size_t)K * 4 + 255) / 256 * 256 + ((size_t)(K + VQ_MAX_CHUNK) * 4 + 255) / 256 * 256 +
         (size_t)((K + 15) / 16 + 64) * 16 * 128 * 4;
}

struct VqLayout { size_t hdr, counts_fix, partial, amb, amb_lim, prep, total; int grid; };
static VqLayout vq_layout(int64_t N, int K, int d) {
  VqLayout L;
  L.grid = vq_grid(N, d);
  size_t o = 0;
  L.hdr = o; o += 256;
  L.counts_fix = o; o += ((size_t)K * 4 + 255) / 256 * 256;       // [hdr, counts_fix / counts_acc] are zeroed every call
  L.partial = o; o += ((size_t)L.grid * 16 * 4 + 255) / 256 * 256;
  L.amb = o; o += ((size_t)N * 4 + 255) / 256 * 256;
  L.amb_lim = o; o += ((size_t)N * 4 + 255) / 256 * 256;
  L.prep = o; o += vq_prep_bytes_max(K);                           // prepared codebook of the one-call entry point
  L.total = o;
  return L;
}

template <typename T, int NF>
static int launch_vq_prepare(const float* E, int64_t N, int K, int d, char* prep, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  const VqPrep P = vq_prep_layout<T, NF>(N, K, d);
  const int npack_blocks = (P.npk + 255) / 256;
  const int kz = P.kpad > 64 ? P.kpad : 64;                         // the norms blocks also zero the 64 header ints of the control block
  FRL_LAUNCH((vq_prepare_kernel<T, NF>), dim3(npack_blocks + (kz + 255) / 256), dim3(256), 0, st, E, K, d, (float*)(prep + P.en), P.kpad,
             (frag_t*)(prep + P.pack), P.npk, npack_blocks, (int*)(prep + P.ctl));
  return frl_check_launch("vq_prepare");
}

static int g_vq_stream_tiles = -1;    // frl_vq_stream_tiles(): -1 = the FRL_VQ_STREAM environment variable, else VQ_STREAM_DEFAULT
static int vq_stream_tiles();
#ifndef VQ_STREAM_DEFAULT
#define VQ_STREAM_DEFAULT 2   // tiles per wave and batch of the streaming kernel when FRL_VQ_STREAM is unset (0: resident kernel)
#endif
static int vq_stream_tiles() {
  if (g_vq_stream_tiles >= 0) return g_vq_stream_tiles;
  static const int env = [] { const char* e = getenv("FRL_VQ_STREAM"); return e ? atoi(e) : VQ_STREAM_DEFAULT; }();
  return env;
}
#ifndef VQ_RES_NT
#define VQ_RES_NT 4        // 16-row tiles per wave and batch of the resident kernel
#endif
template <typename T, int NF>
static int launch_vq(const void* z, const float* E, char* prep, int64_t N, int K, int d, int32_t* idx, void* zq, float* stats,
                     int32_t* counts, char* ws, hipStream_t st) {
  typedef typename DT<T>::frag_t frag_t;
  const VqLayout L = vq_layout(N, K, d);
  const VqPrep P = vq_prep_layout<T, NF>(N, K, d);
  const int nw = vq_waves(N, d);
  const int Kc = P.Kc;
  VqHeader* hdr = (VqHeader*)(ws + L.hdr);
  const bool resident = P.kpad == Kc;                                // the whole codebook is one LDS chunk
  if (!resident) FRL_HIP(hipMemsetAsync(ws, 0, L.partial, st));      // header + counts_fix of the multi-chunk path
  if (prep == nullptr) {                                             // one-call entry point: prepare into the workspace first
    const int rc = launch_vq_prepare<T, NF>(E, N, K, d, ws + L.prep, st);
    if (rc) return rc;
    prep = ws + L.prep;
  }
  const float* en = (const float*)(prep + P.en);
  const frag_t* pk = (const frag_t*)(prep + P.pack);
  if constexpr (sizeof(T) == 2 && NF == 2) {
    // streaming form (one 16-wave workgroup per CU, no workgroup barrier in the batch loop); FRL_VQ_STREAM=0 selects the older kernel,
    // FRL_VQ_STREAM=1 / 2 / 4 the 16-row tiles per wave and batch (A/B hook, read once)
    const int stream_nt = vq_stream_tiles();
    int nt = stream_nt >= 4 ? 4 : (stream_nt >= 2 ? 2 : 1);
    while (nt > 1 && N % (VQS_NW * nt * 16) != 0) nt >>= 1;          // whole batches only: fewer tiles per wave when the row count asks for it
    const int batch = VQS_NW * nt * 16;
    if (resident && stream_nt > 0 && d == 64 && (Kc & 15) == 0 && N % batch == 0 && N < ((int64_t)1 << 32)) {   // whole batches only: no row guards in the kernel
      const int64_t nb = (N + batch - 1) / batch;
      const int grid_s = (int)(nb < 256 ? nb : 256);
      const size_t lds = (size_t)(Kc / 16) * NF * 64 * sizeof(frag_t) + (size_t)Kc * 4 + (size_t)((K + 3) & ~3) * 4 + 64 * 4 +
                         (size_t)VQ_FAST_ROWS * 16 + (size_t)VQS_NW * VQS_CAP * 8 + (size_t)VQS_CAND * 4 +
                         (size_t)VQS_NW * (VQS_CAP + 16) * VQS_ZP * sizeof(T);
      if (lds > 160 * 1024) return frl_fail(-3, "vq_assign: LDS budget exceeded");
#define VQ_GOS(NT_)                                                                                                                \
  do {                                                                                                                             \
    auto kern = vq_assign_stream_kernel<NT_>;                                                                                      \
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    FRL_LAUNCH_AS("vq_assign_kernel", kern, dim3(grid_s), dim3(64 * VQS_NW), lds, st, (const bf16*)z, en, N, K, Kc, idx, (bf16*)zq,  \
                  (float*)(ws + L.partial), (const bf16x8*)pk, (VqCtl*)(prep + P.ctl), counts, stats, (unsigned*)(ws + L.amb),      \
                  (float*)(ws + L.amb_lim));                                                                                       \
  } while (0)
      if (nt == 4) VQ_GOS(4); else if (nt == 2) VQ_GOS(2); else VQ_GOS(1);
#undef VQ_GOS
      return frl_check_launch("vq_assign");
    }
  }
  if (resident) {
    // image | norms | histogram | per-row minima | parked rows | candidates | misc; the statistics fold of the last workgroup reuses the front
    const int nwr = (nw == 8 && NF <= (sizeof(T) == 2 ? 2 : 4)) ? 8 : 4;  // 8 waves x 4 tiles only where the tiles fit the 128-register budget
    const int batch = nwr * VQ_RES_NT * 16;
    const int64_t nb = (N + batch - 1) / batch;
    const int grid_r = (int)(nb < L.grid ? nb : L.grid);
    size_t lds = (size_t)(Kc / 16) * NF * 64 * sizeof(frag_t) + (size_t)Kc * 4 + (size_t)((K + 1) & ~1) * 4 + (size_t)VQ_FAST_ROWS * 12 +
                 (size_t)2 * batch * 4 + (size_t)nwr * 64 * 4 + 32 * 4 + (size_t)VQ_PZ_CAP * d * sizeof(T);
    const size_t fold = (size_t)2 * nwr * 64 * sizeof(double);
    if (lds < fold) lds = fold;
    if (lds > 160 * 1024) return frl_fail(-3, "vq_assign: LDS budget exceeded");
#define VQ_GO(NW_)                                                                                                                 \
  do {                                                                                                                             \
    auto kern = vq_assign_resident_kernel<T, NF, VQ_RES_NT, NW_>;                                                                          \
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    FRL_LAUNCH_AS("vq_assign_kernel", kern, dim3(grid_r), dim3(64 * NW_), lds, st, (const T*)z, E, en, N, K, d, Kc, idx, (T*)zq,   \
                  (float*)(ws + L.partial), pk, (VqCtl*)(prep + P.ctl), counts, stats);                                            \
  } while (0)
    if constexpr (NF <= (sizeof(T) == 2 ? 2 : 4)) {
      if (nwr == 8) VQ_GO(8); else VQ_GO(4);
    } else {
      VQ_GO(4);
    }
#undef VQ_GO
    return frl_check_launch("vq_assign");
  }
  // multi-chunk: two half-size chunk buffers (fragments + norms each), so that two workgroups still fit a CU
  const int Kc2 = Kc >= 32 ? Kc / 2 : Kc;
  const size_t lds = 2 * ((size_t)(Kc2 / 16) * NF * 64 * sizeof(frag_t) + (size_t)Kc2 * 4) + 64 + (K <= 2048 ? (size_t)K * 4 : 0);
  if (((Kc2 / 16) * NF * 64 * sizeof(frag_t)) % 1024 != 0 || (Kc2 & 63) != 0) return frl_fail(-3, "vq_assign: chunk is not a whole number of DMA pieces");
#define VQ_GO(NT_, NW_)                                                                                                            \
  do {                                                                                                                             \
    auto kern = vq_assign_kernel<T, NF, NT_, NW_>;                                                                                 \
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    FRL_LAUNCH_AS("vq_assign_kernel", kern, dim3(L.grid), dim3(64 * NW_), lds, st, (const T*)z, E, en, N, K, d, Kc2, idx, (T*)zq,   \
                  (float*)(ws + L.partial), (int32_t*)(ws + L.counts_fix), hdr, (int32_t*)(ws + L.amb), (float*)(ws + L.amb_lim), pk); \
  } while (0)
  if (nw == 8) VQ_GO(2, 8); else VQ_GO(4, 4);
#undef VQ_GO
  FRL_LAUNCH_AS("vq_fixup_tile_kernel", (vq_fixup_tile_kernel<T, NF>), dim3(1024), dim3(64 * VQ_FIXT_WAVES), 0, st, (const T*)z, E, en, pk, K, P.kpad, d,
                (const int32_t*)(ws + L.amb), (const float*)(ws + L.amb_lim), hdr, idx, (T*)zq, (int32_t*)(ws + L.counts_fix));
  FRL_LAUNCH(vq_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)(ws + L.partial), L.grid * nw, (const VqHeader*)hdr,
             (const int32_t*)(ws + L.counts_fix), counts, K, N, d, stats);
  return frl_check_launch("vq_assign");
}

static int vq_bwd_chunk(int K, int d) {
  int kc = K;
  while ((size_t)kc * d * 4 > 96 * 1024 && kc > 1) kc = (kc + 1) / 2;
  return kc;
}


// ---------------------------------------------------------------------------------------------------------------
// Dead-code revival (CodebookManager of the legacy trainer, scripts/train_vqvae.py:92,196-198; the manager module itself is not
// in the reference tree -- build definition): a code whose usage count over the manager's window is below `min_count` is re-seeded
// with an encoder output row of the current batch, row index = splitmix64(seed + k) mod N; its AdamW moments are cleared so the
// stale momentum does not drag the new vector back.  One workgroup per code, no host round trip; `revived` counts them.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long vq_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <typename T>
__global__ __launch_bounds__(64) void vq_revive_kernel(float* __restrict__ E, const long long* __restrict__ window_counts, long long min_count,
                                                       const T* __restrict__ z, long long N, int K, int d, unsigned long long seed,
                                                       float* __restrict__ m, float* __restrict__ v, int* __restrict__ revived) {
  const int k = blockIdx.x;
  if (k >= K || window_counts[k] >= min_count) return;
  const unsigned long long r = vq_splitmix64(seed + (unsigned long long)k) % (unsigned long long)N;
  for (int j = threadIdx.x; j < d; j += 64) {
    E[(size_t)k * d + j] = to_f32(z[(size_t)r * d + j]);
    if (m != nullptr) m[(size_t)k * d + j] = 0.f;
    if (v != nullptr) v[(size_t)k * d + j] = 0.f;
  }
  if (threadIdx.x == 0) atomicAdd(revived, 1);
}

extern "C" {

// A/B hook: 16-row tiles per wave and batch of the streaming assignment kernel (1, 2 or 4), 0 = the resident kernel of round 2,
// -1 = back to the default (FRL_VQ_STREAM, else the built-in choice).  Returns the previous setting.  Results are identical either way.
int frl_vq_stream_tiles(int nt) {
  const int prev = g_vq_stream_tiles;
  g_vq_stream_tiles = nt < 0 ? -1 : nt;
  return prev;
}

size_t frl_vq_workspace_bytes(int64_t N, int K, int d) {
  const VqLayout L = vq_layout(N, K, d);
  const size_t bwd = (size_t)VQ_BWD_WGS * K * d * 4 + (size_t)K * d * 4;
  return L.total > bwd ? L.total : bwd;
}

// Dispatch on (dtype, d) to the fragment count NF of the kernels.
#define VQ_DISPATCH(dtype, d, CALL)                                              \
  do {                                                                          \
    if ((dtype) == FRL_F32) {                                                   \
      if ((d) <= 16) return CALL(float, 4);                                     \
      if ((d) <= 32) return CALL(float, 8);                                     \
      if ((d) <= 64) return CALL(float, 16);                                    \
      return CALL(float, 32);                                                   \
    } else if ((dtype) == FRL_BF16) {                                           \
      if ((d) <= 32) return CALL(bf16, 1);                                      \
      if ((d) <= 64) return CALL(bf16, 2);                                      \
      return CALL(bf16, 4);                                                     \
    }                                                                           \
    return frl_fail(-2, "vq: bad dtype");                                       \
  } while (0)

// Prepared codebook: ||e||^2 + the packed MFMA fragment image of -2 e.  It depends on the codebook only, so a caller that keeps the
// codebook fixed over several assignments (inference), or knows when it changes (once per optimizer step), builds it once with
// frl_vq_prepare and hands it to frl_vq_assign_fwd_prepared: the assignment is then ONE kernel launch (plus a memset node).
// N only selects the kernel variant (workgroup shape); pass the row count the assignments will use.
size_t frl_vq_prepared_bytes(int K, int d) { return (K > 0 && d > 0 && d <= 128) ? vq_prep_bytes_max(K) : 0; }

int frl_vq_prepare(const float* E, int64_t N, int K, int d, int dtype, void* prep, size_t prep_bytes, hipStream_t stream) {
  if (N <= 0 || K <= 0 || d <= 0) return frl_fail(-2, "vq_prepare: empty input");
  if (d > 128) return frl_fail(-2, "vq_prepare: d > 128 unsupported");
  if (prep == nullptr || prep_bytes < frl_vq_prepared_bytes(K, d)) return frl_fail(-4, "vq_prepare: buffer too small");
#define VQ_CALL_PREP(T_, NF_) launch_vq_prepare<T_, NF_>(E, N, K, d, (char*)prep, stream)
  VQ_DISPATCH(dtype, d, VQ_CALL_PREP);
#undef VQ_CALL_PREP
}

// z [N][d] (dtype), E [K][d] f32 master codebook.  Outputs: idx_out [N] int32, zq_out [N][d] (dtype, the
// codebook rows rounded to dtype), stats_out [4] f32 = {sum ||z - z_q||^2, perplexity, #rows re-evaluated in
// float64, mean squared error = sum / (N d)}, counts_out [K] int32 code usage.  In bf16 mode distances are taken to the bf16-rounded codebook.
// prep: NULL, or the image frl_vq_prepare wrote for THIS codebook content, dtype and row count class.  The image carries the kernel's
// arrival counter and histogram accumulator (zeroed by frl_vq_prepare, left zeroed by every call): one call in flight per image.
int frl_vq_assign_fwd_prepared(const void* z, const float* E, void* prep, int64_t N, int K, int d, int32_t* idx_out, void* zq_out,
                               float* stats_out, int32_t* counts_out, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (N <= 0 || K <= 0 || d <= 0) return frl_fail(-2, "vq_assign: empty input");
  if (d > 128) return frl_fail(-2, "vq_assign: d > 128 unsupported");
  if (ws_bytes < frl_vq_workspace_bytes(N, K, d)) return frl_fail(-4, "vq_assign: workspace too small");
#define VQ_CALL_FWD(T_, NF_) launch_vq<T_, NF_>(z, E, (char*)prep, N, K, d, idx_out, zq_out, stats_out, counts_out, (char*)ws, stream)
  VQ_DISPATCH(dtype, d, VQ_CALL_FWD);
#undef VQ_CALL_FWD
}

int frl_vq_assign_fwd(const void* z, const float* E, int64_t N, int K, int d, int32_t* idx_out, void* zq_out,
                      float* stats_out, int32_t* counts_out, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  return frl_vq_assign_fwd_prepared(z, E, nullptr, N, K, d, idx_out, zq_out, stats_out, counts_out, dtype, ws, ws_bytes, stream);
}

// g_z = g_out + gscale[0] * beta * 2/(N d) * (z - e_idx);  g_E[k] = gscale[1] * 2/(N d) * (n_k e_k - sum_{idx=k} z)
// gscale: device float[2] = upstream gradients of {L_commit, L_codebook}, may be null (= {1, 1}).
// zq (optional): the forward's z_q rows; when given the bf16 path reads it instead of gathering codebook rows.  g_out may be null (=0).
// g_z / g_E may be null to skip.  sums_out (optional, [K][d] f32) receives the per-code sums of z.
int frl_vq_bwd(const void* g_out, const void* z, const void* zq, const float* E, const int32_t* idx, const int32_t* counts,
               const float* gscale, float beta, int64_t N, int K, int d, void* g_z_out, float* g_E_out,
               float* sums_out, int dtype, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (N <= 0) return frl_fail(-2, "vq_bwd: empty input");
  if (ws_bytes < frl_vq_workspace_bytes(N, K, d)) return frl_fail(-4, "vq_bwd: workspace too small");
  const int Kc = vq_bwd_chunk(K, d);
  const size_t lds = (size_t)Kc * d * 4;
  const int64_t rows = (N + VQ_BWD_WGS - 1) / VQ_BWD_WGS;
  const float cz = beta * 2.f / ((float)N * (float)d), ce = 2.f / ((float)N * (float)d);
  float* slab = (float*)ws;
  if (dtype == FRL_F32) {
    auto kern = vq_bwd_kernel<float>;
    if (lds > 64 * 1024) FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FRL_LAUNCH_AS("vq_bwd_kernel", kern, dim3(VQ_BWD_WGS), dim3(256), lds, stream, (const float*)g_out, (const float*)z, E, idx, gscale,
                       cz, N, K, d, Kc, rows, (float*)g_z_out, slab);
    launch_slab_reduce_deferrable<float, CodeEpi>((const float*)slab, VQ_BWD_WGS, (int64_t)K * d, CodeEpi{E, counts, gscale, ce, d, 0, g_E_out, sums_out}, stream, sums_out == nullptr);
  } else if (dtype == FRL_BF16) {
    const int64_t rows64 = (rows + 63) / 64 * 64;
    if (d <= 32) {
      FRL_LAUNCH((vq_bwd_mfma_kernel<4, 2, 8>), dim3(VQ_BWD_WGS), dim3(512), (size_t)64 * 40 * 2 + 256, stream, (const bf16*)g_out, (const bf16*)z, (const bf16*)zq, E, idx,
                 gscale, cz, N, K, d, rows64, (bf16*)g_z_out, slab);
    } else if (d <= 64) {
      FRL_LAUNCH((vq_bwd_mfma_kernel<4, 4, 8>), dim3(VQ_BWD_WGS), dim3(512), (size_t)64 * 72 * 2 + 256, stream, (const bf16*)g_out, (const bf16*)z, (const bf16*)zq, E, idx,
                 gscale, cz, N, K, d, rows64, (bf16*)g_z_out, slab);
    } else {
      FRL_LAUNCH((vq_bwd_mfma_kernel<2, 8, 8>), dim3(VQ_BWD_WGS), dim3(512), (size_t)64 * 136 * 2 + 256, stream, (const bf16*)g_out, (const bf16*)z, (const bf16*)zq, E, idx,
                 gscale, cz, N, K, d, rows64, (bf16*)g_z_out, slab);
    }
    launch_slab_reduce_deferrable<float, CodeEpi>((const float*)slab, VQ_BWD_WGS, (int64_t)K * d, CodeEpi{E, counts, gscale, ce, d, 1, g_E_out, sums_out}, stream, sums_out == nullptr);
  } else return frl_fail(-2, "vq_bwd: bad dtype");
  return frl_check_launch("vq_bwd");
}

// EMA codebook update from this batch's assignments: counts [K] int32 and per-code sums [K][d] f32
// (both produced by frl_vq_assign_fwd / frl_vq_bwd(sums_out)).  Updates ema_count, ema_sum, E in place.
int frl_vq_ema_update(const float* sums, const int32_t* counts, int K, int d, float decay, float eps, float* ema_count,
                      float* ema_sum, float* E, const float* ok, hipStream_t stream) {
  FRL_LAUNCH(vq_ema_kernel, dim3(1), dim3(256), 0, stream, sums, counts, K, d, decay, eps, ema_count, ema_sum, E, ok);
  return frl_check_launch("vq_ema_update");
}

// E [K][d] float32 codebook (updated in place), window_counts [K] int64, z [N][d] rows of `dtype`, m / v optional AdamW moment
// rows of the codebook, revived: device int32 incremented by the number of re-seeded codes.
int frl_vq_revive_dead_codes(float* E, const int64_t* window_counts, int64_t min_count, const void* z, int64_t N, int K, int d,
                             uint64_t seed, float* m, float* v, int32_t* revived, int dtype, hipStream_t stream) {
  if (K <= 0 || d <= 0) return frl_fail(-2, "vq_revive_dead_codes: bad codebook shape");
  if (N <= 0) return frl_fail(-2, "vq_revive_dead_codes: no candidate rows");
  if (dtype == FRL_F32)
    FRL_LAUNCH((vq_revive_kernel<float>), dim3(K), dim3(64), 0, stream, E, (const long long*)window_counts, (long long)min_count,
               (const float*)z, (long long)N, K, d, (unsigned long long)seed, m, v, revived);
  else if (dtype == FRL_BF16)
    FRL_LAUNCH((vq_revive_kernel<bf16>), dim3(K), dim3(64), 0, stream, E, (const long long*)window_counts, (long long)min_count,
               (const bf16*)z, (long long)N, K, d, (unsigned long long)seed, m, v, revived);
  else return frl_fail(-2, "vq_revive_dead_codes: dtype must be FRL_F32 or FRL_BF16");
  return frl_check_launch("vq_revive_dead_codes");
}

}  // extern "C"
