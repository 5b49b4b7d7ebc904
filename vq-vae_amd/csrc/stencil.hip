// Fixed-filter stencils of EdgeAwareSmoothingConv2D (frl/models/spatial.py):
//   * depthwise Sobel/4 gradients  (spatial.py:240-249,295-296)  -> cat[dx,dy] written as one [P][2C] tensor (K5)
//   * directional bank: 4 orientations x {dilation 1, dilation 3} 3-tap line averages (taps 1/3, zero padding,
//     spatial.py:224-237,315-325), mixed per channel with the rank-R factored weights
//     w[k][c] = sum_r softmax_k(A)[k][r] * softmax_r(B)[c][r]  (spatial.py:300-307,324-328), residual = x - smoothed (K9,K10).
//   The [B,C,R,H,W] slot tensor and the 8 filtered copies of x are never materialised: one thread owns a pixel's
//   16-byte channel vector, gathers the 17 neighbours it needs (L1/L2 resident) and mixes in registers.
//   The softmaxed A / B maps are written once (they are the only saved state the backward needs).
// Roofline: HBM (about 17 cached neighbour reads, ~9 C bytes of compulsory traffic per pixel).
#include "frl_common.hpp"
#include "frl_host.hpp"

#define ST_ND 4

__device__ __forceinline__ void st_dir(int i, int& dy, int& dx) {
  // templates of spatial.py:224-229: E-W, N-S, diagonal (\), anti-diagonal (/)
  dy = (i == 0) ? 0 : 1;
  dx = (i == 0) ? 1 : (i == 1) ? 0 : (i == 2) ? 1 : -1;
}

template <typename T, int V>
__device__ __forceinline__ bool st_load(const T* __restrict__ X, int b, int y, int x, int H, int W, int C, int c0, float* v) {
  if (y < 0 || y >= H || x < 0 || x >= W) {
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    return false;
  }
  Vec<T>::load(X + ((((int64_t)b * H + y) * W + x) * C + c0), v);
  return true;
}

// (pixel, channel vector) of flat element i: three 32-bit divisions where the tensor has fewer than 2^31 vectors (always, in practice) instead
// of five 64-bit ones -- the index arithmetic was most of the instruction stream of these streaming kernels
struct StIdx { int b, y, x, c0; int64_t p; };
__device__ __forceinline__ StIdx st_index(int64_t i, int vpr, int V, int H, int W, bool small) {
  StIdx r;
  if (small) {
    const unsigned u = (unsigned)i, p = u / (unsigned)vpr, row = p / (unsigned)W, b = row / (unsigned)H;
    r.c0 = (int)(u - p * (unsigned)vpr) * V;
    r.x = (int)(p - row * (unsigned)W);
    r.y = (int)(row - b * (unsigned)H);
    r.b = (int)b;
    r.p = (int64_t)p;
  } else {
    r.c0 = (int)(i % vpr) * V;
    r.p = i / vpr;
    r.x = (int)(r.p % W);
    r.y = (int)((r.p / W) % H);
    r.b = (int)(r.p / ((int64_t)W * H));
  }
  return r;
}

// ------------------------------------------------------------------------------------------------ Sobel
template <typename T, int V>
__global__ __launch_bounds__(256) void sobel_fwd_kernel(const T* __restrict__ X, T* __restrict__ G, int B, int H, int W, int C) {
  const int vpr = C / V;
  const int64_t total = (int64_t)B * H * W * vpr;
  const bool small = total < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const StIdx ix = st_index(i, vpr, V, H, W, small);
    const int c0 = ix.c0, x = ix.x, y = ix.y, b = ix.b;
    const int64_t p = ix.p;
    float gx[V], gy[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { gx[e] = 0.f; gy[e] = 0.f; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (ky == 1 && kx == 1) continue;
        const float wx = (float)(kx - 1) * (ky == 1 ? 2.f : 1.f) * 0.25f;   // [[-1,0,1],[-2,0,2],[-1,0,1]]/4
        const float wy = (float)(ky - 1) * (kx == 1 ? 2.f : 1.f) * 0.25f;   // transpose
        float v[V];
        if (!st_load<T, V>(X, b, y + ky - 1, x + kx - 1, H, W, C, c0, v)) continue;
#pragma unroll
        for (int e = 0; e < V; ++e) { gx[e] = fmaf(wx, v[e], gx[e]); gy[e] = fmaf(wy, v[e], gy[e]); }
      }
    Vec<T>::store(G + p * 2 * C + c0, gx);
    Vec<T>::store(G + p * 2 * C + C + c0, gy);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void sobel_bwd_kernel(const T* __restrict__ DG, T* __restrict__ DX, const T* __restrict__ DXADD, int B, int H, int W,
                                                        int C) {
  const int vpr = C / V;
  const int64_t total = (int64_t)B * H * W * vpr;
  const bool small = total < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const StIdx ix = st_index(i, vpr, V, H, W, small);
    const int c0 = ix.c0, x = ix.x, y = ix.y, b = ix.b;
    const int64_t p = ix.p;
    float o[V];
    if (DXADD != nullptr) {                                      // gradient of x through its other consumer, accumulated here
      Vec<T>::load(DXADD + p * C + c0, o);
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (ky == 1 && kx == 1) continue;
        // output pixel q = p - (ky-1, kx-1) read x[p] with weight w[ky][kx]
        const int qy = y - (ky - 1), qx = x - (kx - 1);
        if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
        const float wx = (float)(kx - 1) * (ky == 1 ? 2.f : 1.f) * 0.25f;
        const float wy = (float)(ky - 1) * (kx == 1 ? 2.f : 1.f) * 0.25f;
        const int64_t qp = ((int64_t)b * H + qy) * W + qx;
        float a[V], c[V];
        Vec<T>::load(DG + qp * 2 * C + c0, a);
        Vec<T>::load(DG + qp * 2 * C + C + c0, c);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = fmaf(wx, a[e], fmaf(wy, c[e], o[e]));
      }
    Vec<T>::store(DX + p * C + c0, o);
  }
}

// ------------------------------------------------------------------------------------------------ smoothing
// One workgroup = 256 / TPP pixels; thread (pixel, channel vector).  TPP = power of two >= C / V.
template <typename T, int V, int R>
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const T* __restrict__ X, const T* __restrict__ AL, const T* __restrict__ BL,
                                                         T* __restrict__ SM, T* __restrict__ RES, T* __restrict__ AS, T* __restrict__ BS,
                                                         int B, int H, int W, int C, int dil, int tpp) {
  constexpr int K = 2 * ST_ND;
  const int vpr = C / V;
  const int ppw = 256 / tpp;
  const int64_t npix = (int64_t)B * H * W;
  // XCD-aware block order: the rows of one image (and their +-1 / +-3 row neighbours) stay in one XCD's L2
  for (int64_t p = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * ppw + threadIdx.x / tpp; p < npix; p += (int64_t)gridDim.x * ppw) {
    const int cvi = threadIdx.x % tpp;
    if (cvi >= vpr) continue;
    const int c0 = cvi * V;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
    // softmax over k for each r
    float A[K][R];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r) A[k][r] = to_f32(AL[p * (K * R) + k * R + r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float m = A[0][r];
#pragma unroll
      for (int k = 1; k < K; ++k) m = fmaxf(m, A[k][r]);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { A[k][r] = expf(A[k][r] - m); s += A[k][r]; }
      const float inv = 1.f / s;
#pragma unroll
      for (int k = 0; k < K; ++k) A[k][r] *= inv;
    }
    if (cvi == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r) AS[p * (K * R) + k * R + r] = from_f32<T>(A[k][r]);
    }
    // softmax over r for each channel
    float Bw[V][R];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float m = -3.0e38f;
#pragma unroll
      for (int r = 0; r < R; ++r) { Bw[e][r] = to_f32(BL[p * ((int64_t)C * R) + (c0 + e) * R + r]); m = fmaxf(m, Bw[e][r]); }
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) { Bw[e][r] = expf(Bw[e][r] - m); s += Bw[e][r]; }
      const float inv = 1.f / s;
#pragma unroll
      for (int r = 0; r < R; ++r) { Bw[e][r] *= inv; BS[p * ((int64_t)C * R) + (c0 + e) * R + r] = from_f32<T>(Bw[e][r]); }
    }
    float ctr[V], sm[V];
    st_load<T, V>(X, b, y, x, H, W, C, c0, ctr);
#pragma unroll
    for (int e = 0; e < V; ++e) sm[e] = 0.f;
    const float third = 1.f / 3.f;
#pragma unroll
    for (int i = 0; i < ST_ND; ++i) {
      int dy, dx;
      st_dir(i, dy, dx);
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {
        const int d = sc ? dil : 1;
        float a[V], c[V];
        st_load<T, V>(X, b, y - d * dy, x - d * dx, H, W, C, c0, a);
        st_load<T, V>(X, b, y + d * dy, x + d * dx, H, W, C, c0, c);
        const int k = 2 * i + sc;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float f = third * a[e] + third * ctr[e] + third * c[e];
          float w = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) w = fmaf(Bw[e][r], A[k][r], w);
          sm[e] = fmaf(w, f, sm[e]);
        }
      }
    }
    float res[V];
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] = ctr[e] - sm[e];
    Vec<T>::store(SM + p * C + c0, sm);
    Vec<T>::store(RES + p * C + c0, res);
  }
}

template <typename T, int V, int R>
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const T* __restrict__ DS, const T* __restrict__ X, const T* __restrict__ AS,
                                                         const T* __restrict__ BS, T* __restrict__ DX, T* __restrict__ DAL, T* __restrict__ DBL,
                                                         const T* __restrict__ DXADD, int B, int H, int W, int C, int dil, int tpp) {
  constexpr int K = 2 * ST_ND;
  const int vpr = C / V;
  const int ppw = 256 / tpp;
  const int64_t npix = (int64_t)B * H * W;
  const int64_t nloop = (npix + ppw - 1) / ppw;
  const float third = 1.f / 3.f;
  for (int64_t it = xcd_remap(blockIdx.x, gridDim.x); it < nloop; it += gridDim.x) {
    const int64_t p = it * ppw + threadIdx.x / tpp;
    const int cvi = threadIdx.x % tpp;
    const bool active = (p < npix) && (cvi < vpr);
    const int64_t pc = p < npix ? p : npix - 1;
    const int c0 = (cvi < vpr ? cvi : 0) * V;
    const int x = (int)(pc % W), y = (int)((pc / W) % H), b = (int)(pc / ((int64_t)W * H));
    float A[K][R], Bw[V][R], ds[V], ctr[V];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r) A[k][r] = to_f32(AS[pc * (K * R) + k * R + r]);
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
      for (int r = 0; r < R; ++r) Bw[e][r] = to_f32(BS[pc * ((int64_t)C * R) + (c0 + e) * R + r]);
    Vec<T>::load(DS + pc * C + c0, ds);
    st_load<T, V>(X, b, y, x, H, W, C, c0, ctr);
    if (!active) {
#pragma unroll
      for (int e = 0; e < V; ++e) ds[e] = 0.f;
    }
    float slot[V][R], dA[K][R], dx[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      dx[e] = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) slot[e][r] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r) dA[k][r] = 0.f;
#pragma unroll
    for (int i = 0; i < ST_ND; ++i) {
      int dy, dxo;
      st_dir(i, dy, dxo);
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {
        const int d = sc ? dil : 1;
        const int k = 2 * i + sc;
        float a[V], c[V];
        st_load<T, V>(X, b, y - d * dy, x - d * dxo, H, W, C, c0, a);
        st_load<T, V>(X, b, y + d * dy, x + d * dxo, H, W, C, c0, c);
        float wsum_dummy = 0.f; (void)wsum_dummy;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float f = third * a[e] + third * ctr[e] + third * c[e];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            slot[e][r] = fmaf(A[k][r], f, slot[e][r]);
            dA[k][r] = fmaf(ds[e] * Bw[e][r], f, dA[k][r]);
          }
          // centre-tap contribution of this filter to d x[p]
          float w = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) w = fmaf(Bw[e][r], A[k][r], w);
          dx[e] = fmaf(third * w, ds[e], dx[e]);
        }
        // neighbour contributions: pixels q = p -/+ d*delta whose filter k read x[p]
#pragma unroll
        for (int sgn = -1; sgn <= 1; sgn += 2) {
          const int qy = y + sgn * d * dy, qx = x + sgn * d * dxo;
          if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
          const int64_t q = ((int64_t)b * H + qy) * W + qx;
          float dq[V], aq[R];
          Vec<T>::load(DS + q * C + c0, dq);
#pragma unroll
          for (int r = 0; r < R; ++r) aq[r] = to_f32(AS[q * (K * R) + k * R + r]);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            float w = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) w = fmaf(to_f32(BS[q * ((int64_t)C * R) + (c0 + e) * R + r]), aq[r], w);
            dx[e] = fmaf(third * w, dq[e], dx[e]);
          }
        }
      }
    }
    // d b_logit (softmax over r backward)
    if (active) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) dot = fmaf(Bw[e][r], ds[e] * slot[e][r], dot);
#pragma unroll
        for (int r = 0; r < R; ++r)
          DBL[p * ((int64_t)C * R) + (c0 + e) * R + r] = from_f32<T>(Bw[e][r] * (ds[e] * slot[e][r] - dot));
      }
      if (DXADD != nullptr) {                                   // caller's additive term (the residual branch of EdgeSmoothFn: dx += d_res)
        float ad[V];
        if constexpr (V == 1) ad[0] = to_f32(DXADD[p * C + c0]); else Vec<T>::load(DXADD + p * C + c0, ad);
#pragma unroll
        for (int e = 0; e < V; ++e) dx[e] += ad[e];
      }
      Vec<T>::store(DX + p * C + c0, dx);
    }
    // d a_logit: reduce dA over the pixel's channel threads (tpp contiguous lanes), then softmax-over-k backward
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int r = 0; r < R; ++r)
        for (int off = 1; off < tpp; off <<= 1) dA[k][r] += __shfl_xor(dA[k][r], off, 64);
    if (active && cvi == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) dot = fmaf(A[k][r], dA[k][r], dot);
#pragma unroll
        for (int k = 0; k < K; ++k) DAL[p * (K * R) + k * R + r] = from_f32<T>(A[k][r] * (dA[k][r] - dot));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 / rank-4 specialisation of the smoothing backward (the hot configuration): same math as smooth_bwd_kernel, but
//   * the rank contraction w[c] = sum_r B[c][r] A[k][r] runs on packed bf16 pairs (v_dot2c_f32_bf16: two ops instead of
//     four unpacks + four FMAs), at the pixel itself and at each of the 16 neighbours whose filters read x[p];
//   * 32-bit element offsets with wave-uniform neighbour strides (no 64-bit per-lane address arithmetic);
//   * borders are handled branch-free: out-of-image neighbours are redirected to the pixel itself with weight 0.
// Thread = (pixel, 8 channels); the 8 threads of a pixel are adjacent lanes (tpp = 8, C = 64).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
struct SmPack16 { bf16x8 v; bf2 p[4]; __device__ __forceinline__ void split() { p[0] = __builtin_shufflevector(v, v, 0, 1); p[1] = __builtin_shufflevector(v, v, 2, 3); p[2] = __builtin_shufflevector(v, v, 4, 5); p[3] = __builtin_shufflevector(v, v, 6, 7); } };

__device__ __forceinline__ float sm_dot4(const bf2 (&b)[2], const bf2 (&a)[2]) {
  return __builtin_amdgcn_fdot2_f32_bf16(b[0], a[0], __builtin_amdgcn_fdot2_f32_bf16(b[1], a[1], 0.f, false), false);
}

__global__ __launch_bounds__(256, 2) void smooth_bwd_bf16r4_kernel(const bf16* __restrict__ DS, const bf16* __restrict__ X,
                                                                   const bf16* __restrict__ AS, const bf16* __restrict__ BS,
                                                                   bf16* __restrict__ DX, bf16* __restrict__ DAL, bf16* __restrict__ DBL,
                                                                   const bf16* __restrict__ DXADD, int B, int H, int W, int dil) {
  constexpr int K = 2 * ST_ND, R = 4, V = 8, C = 64, VPR = 8;   // 8 threads per pixel == 8 filters: lane cvi keeps filter cvi's dA
  constexpr int PPW = 256 / VPR;
  const int npix = B * H * W;
  const int nloop = (npix + PPW - 1) / PPW;
  const float third = 1.f / 3.f;
  for (int it = (int)xcd_remap(blockIdx.x, gridDim.x); it < nloop; it += gridDim.x) {
    const int p = it * PPW + (int)threadIdx.x / VPR;
    const int cvi = (int)threadIdx.x % VPR;
    const bool active = p < npix;
    const int pc = active ? p : npix - 1;
    const int c0 = cvi * V;
    const int x = pc % W, y = (pc / W) % H;
    // ---- this pixel: softmaxed A [8][4] (packed pairs), B [8 ch][4] (float), ds, x
    float Bw[V][R];
    {
      const bf16* bp = BS + pc * (C * R) + c0 * R;
#pragma unroll
      for (int i = 0; i < 4; ++i) Vec<bf16>::load(bp + 8 * i, &Bw[2 * i][0]);
    }
    float ds[V], ctr[V];
    Vec<bf16>::load(DS + pc * C + c0, ds);
    Vec<bf16>::load(X + pc * C + c0, ctr);
    if (!active) {
#pragma unroll
      for (int e = 0; e < V; ++e) ds[e] = 0.f;
    }
    // every term carries the tap weight 1/3: accumulate unscaled, scale once at the end
    float slot[V][R], dx[V], myA[R] = {0.f, 0.f, 0.f, 0.f}, Ak[R] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < V; ++e) {
      dx[e] = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) slot[e][r] = 0.f;
    }
    // Rolled loop over the 8 filters (k = 2 * direction + scale): everything a filter needs is fetched inside its iteration,
    // so the live state stays ~100 registers and the compiler cannot interleave the 16 neighbour gathers.
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int i = k >> 1, d = (k & 1) ? dil : 1;
      const int dy = (i == 0) ? 0 : 1;
      const int dxo = (i == 0) ? 1 : (i == 1) ? 0 : (i == 2) ? 1 : -1;
      const int dpix = d * (dy * W + dxo);                 // wave-uniform pixel stride of this filter's outer taps
      bf2 ak[2];                                           // softmaxed A[p][k][0..3]
      {
        const bf2* akp = reinterpret_cast<const bf2*>(AS + pc * (K * R) + k * R);
        ak[0] = akp[0]; ak[1] = akp[1];
      }
      const float Af[R] = {(float)ak[0][0], (float)ak[0][1], (float)ak[1][0], (float)ak[1][1]};
      float f[V];
#pragma unroll
      for (int e = 0; e < V; ++e) f[e] = ctr[e];
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int qy = y + sgn * d * dy, qx = x + sgn * d * dxo;
        const bool inb = (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
        const int q = inb ? pc + sgn * dpix : pc;
        const float xm = inb ? 1.f : 0.f;
        // (1) x[q] feeds this pixel's filter k (zero padding outside the image)
        float xv[V];
        Vec<bf16>::load(X + q * C + c0, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) f[e] = fmaf(xm, xv[e], f[e]);
        // (2) pixel q's filter k read x[p]:  dx[p] += w_k[q] * ds[q]
        float dq[V];
        Vec<bf16>::load(DS + q * C + c0, dq);
        bf2 aq[2];
        {
          const bf2* aqp = reinterpret_cast<const bf2*>(AS + q * (K * R) + k * R);
          aq[0] = aqp[0]; aq[1] = aqp[1];
        }
        const bf16x8* bq = reinterpret_cast<const bf16x8*>(BS + q * (C * R) + c0 * R);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          SmPack16 u; u.v = bq[i2]; u.split();
          const bf2 b0[2] = {u.p[0], u.p[1]}, b1[2] = {u.p[2], u.p[3]};
          dx[2 * i2] = fmaf(sm_dot4(b0, aq), xm * dq[2 * i2], dx[2 * i2]);
          dx[2 * i2 + 1] = fmaf(sm_dot4(b1, aq), xm * dq[2 * i2 + 1], dx[2 * i2 + 1]);
        }
      }
      float dAk[R] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float g = ds[e] * f[e];
        float w = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          slot[e][r] = fmaf(Af[r], f[e], slot[e][r]);
          dAk[r] = fmaf(Bw[e][r], g, dAk[r]);
          w = fmaf(Bw[e][r], Af[r], w);
        }
        dx[e] = fmaf(w, ds[e], dx[e]);                       // centre tap of this pixel's own filter
      }
      // reduce filter k's dA over the pixel's 8 channel threads; lane cvi == k keeps it (and its A[k][:])
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) dAk[r] += __shfl_xor(dAk[r], off, 64);
        myA[r] = (cvi == k) ? dAk[r] : myA[r];
        Ak[r] = (cvi == k) ? Af[r] : Ak[r];
      }
    }
    // ---- d b_logit (softmax over r backward), dx
    if (active) {
      float ob[V * R];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) { slot[e][r] *= third * ds[e]; dot = fmaf(Bw[e][r], slot[e][r], dot); }
#pragma unroll
        for (int r = 0; r < R; ++r) ob[e * R + r] = Bw[e][r] * (slot[e][r] - dot);
        dx[e] *= third;
      }
      bf16* dbp = DBL + p * (C * R) + c0 * R;
#pragma unroll
      for (int i = 0; i < 4; ++i) Vec<bf16>::store(dbp + 8 * i, ob + 8 * i);
      if (DXADD != nullptr) {                                   // caller's additive term (dx += d_res of the residual branch)
        float ad[V];
        Vec<bf16>::load(DXADD + p * C + c0, ad);
#pragma unroll
        for (int e = 0; e < V; ++e) dx[e] += ad[e];
      }
      Vec<bf16>::store(DX + p * C + c0, dx);
    }
    // ---- d a_logit: softmax-over-k backward; lane cvi holds filter k = cvi:  dA_logit[k][r] = A[k][r] (dA[k][r] - sum_k' A dA)
    {
      float prod[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        myA[r] *= third;
        prod[r] = Ak[r] * myA[r];
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) prod[r] += __shfl_xor(prod[r], off, 64);
      }
      if (active) {
        bf16x4 o = {(bf16)(Ak[0] * (myA[0] - prod[0])), (bf16)(Ak[1] * (myA[1] - prod[1])), (bf16)(Ak[2] * (myA[2] - prod[2])),
                    (bf16)(Ak[3] * (myA[3] - prod[3]))};
        *reinterpret_cast<bf16x4*>(DAL + p * (K * R) + cvi * R) = o;
      }
    }
  }
}

// bf16 / rank-4 / 64-channel forward: thread = (pixel, 8 channels), the pixel's 8 threads are adjacent lanes.  Lane cvi owns filter
// k = cvi of the softmax over k (4 exps per lane instead of 32, max / sum by 8-lane shuffles); the filter loop is rolled.
__global__ __launch_bounds__(256, 4) void smooth_fwd_bf16r4_kernel(const bf16* __restrict__ X, const bf16* __restrict__ AL,
                                                                   const bf16* __restrict__ BL, bf16* __restrict__ SM, bf16* __restrict__ RES,
                                                                   bf16* __restrict__ AS, bf16* __restrict__ BS, int B, int H, int W, int dil) {
  constexpr int K = 2 * ST_ND, R = 4, V = 8, C = 64, VPR = 8, PPW = 256 / VPR;
  const int npix = B * H * W;
  const int nloop = (npix + PPW - 1) / PPW;
  const float third = 1.f / 3.f;
  const int lane = (int)threadIdx.x & 63;
  for (int it = (int)xcd_remap(blockIdx.x, gridDim.x); it < nloop; it += gridDim.x) {
    const int p = it * PPW + (int)threadIdx.x / VPR;
    const int cvi = (int)threadIdx.x % VPR;
    const bool active = p < npix;
    const int pc = active ? p : npix - 1;
    const int c0 = cvi * V;
    const int x = pc % W, y = (pc / W) % H;
    // ---- softmax over k (8 lanes) of A[k = cvi][r]
    float myA[R];
    {
      const bf16x4 a4 = *reinterpret_cast<const bf16x4*>(AL + pc * (K * R) + cvi * R);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float v = (float)a4[r];
        float m = v;
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const float e = __expf(v - m);
        float ssum = e;
#pragma unroll
        for (int off = 1; off < VPR; off <<= 1) ssum += __shfl_xor(ssum, off, 64);
        myA[r] = e / ssum;
      }
      if (active) {
        bf16x4 o = {(bf16)myA[0], (bf16)myA[1], (bf16)myA[2], (bf16)myA[3]};
        *reinterpret_cast<bf16x4*>(AS + p * (K * R) + cvi * R) = o;
      }
    }
    // ---- softmax over r for this thread's 8 channels
    float Bw[V][R];
    {
      const bf16* bp = BL + pc * (C * R) + c0 * R;
#pragma unroll
      for (int i = 0; i < 4; ++i) Vec<bf16>::load(bp + 8 * i, &Bw[2 * i][0]);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float m = fmaxf(fmaxf(Bw[e][0], Bw[e][1]), fmaxf(Bw[e][2], Bw[e][3]));
        float ssum = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) { Bw[e][r] = __expf(Bw[e][r] - m); ssum += Bw[e][r]; }
        const float inv = 1.f / ssum;
#pragma unroll
        for (int r = 0; r < R; ++r) Bw[e][r] *= inv;
      }
      if (active) {
        bf16* bo = BS + p * (C * R) + c0 * R;
#pragma unroll
        for (int i = 0; i < 4; ++i) Vec<bf16>::store(bo + 8 * i, &Bw[2 * i][0]);
      }
    }
    float ctr[V], sm[V];
    Vec<bf16>::load(X + pc * C + c0, ctr);
#pragma unroll
    for (int e = 0; e < V; ++e) sm[e] = 0.f;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int i = k >> 1, d = (k & 1) ? dil : 1;
      const int dy = (i == 0) ? 0 : 1;
      const int dxo = (i == 0) ? 1 : (i == 1) ? 0 : (i == 2) ? 1 : -1;
      const int dpix = d * (dy * W + dxo);
      const int src = (lane & ~7) + k;                     // the lane of this pixel that owns filter k
      float Af[R];
#pragma unroll
      for (int r = 0; r < R; ++r) Af[r] = __shfl(myA[r], src, 64);
      float f[V];
#pragma unroll
      for (int e = 0; e < V; ++e) f[e] = ctr[e];
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int qy = y + sgn * d * dy, qx = x + sgn * d * dxo;
        const bool inb = (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
        const int q = inb ? pc + sgn * dpix : pc;
        const float xm = inb ? 1.f : 0.f;
        float xv[V];
        Vec<bf16>::load(X + q * C + c0, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) f[e] = fmaf(xm, xv[e], f[e]);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float w = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) w = fmaf(Bw[e][r], Af[r], w);
        sm[e] = fmaf(w, third * f[e], sm[e]);
      }
    }
    if (active) {
      float res[V];
#pragma unroll
      for (int e = 0; e < V; ++e) res[e] = ctr[e] - sm[e];
      Vec<bf16>::store(SM + p * C + c0, sm);
      Vec<bf16>::store(RES + p * C + c0, res);
    }
  }
}

static int st_tpp(int vpr) { int t = 1; while (t < vpr) t <<= 1; return t; }
static unsigned st_grid(int64_t n) { int64_t g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (unsigned)g; }

template <typename T, int V>
static int smooth_dispatch(bool fwd, const void* a0, const void* a1, const void* a2, const void* a3, void* o0, void* o1, void* o2,
                           void* o3, int B, int H, int W, int C, int R, int dil, hipStream_t st) {
  const int vpr = C / V, tpp = st_tpp(vpr);
  if (tpp > 64) return frl_fail(-2, "edge_smooth: too many channels");
  const int ppw = 256 / tpp;
  const int64_t npix = (int64_t)B * H * W;
  int64_t grid = (npix + ppw - 1) / ppw;
  if (grid > 8192) grid = 8192;
  if constexpr (sizeof(T) == 2 && V == 8) {
    if (fwd && R == 4 && C == 64 && npix * (int64_t)C * R < ((int64_t)1 << 31)) {
      int64_t g2 = (npix + 31) / 32;
      if (g2 > 16384) g2 = 16384;
      FRL_LAUNCH(smooth_fwd_bf16r4_kernel, dim3((unsigned)g2), dim3(256), 0, st, (const bf16*)a0, (const bf16*)a1, (const bf16*)a2,
                 (bf16*)o0, (bf16*)o1, (bf16*)o2, (bf16*)o3, B, H, W, dil);
      return frl_check_launch("edge_smooth_stencil_fwd");
    }
    if (!fwd && R == 4 && C == 64 && npix * (int64_t)C * R < ((int64_t)1 << 31)) {
      int64_t g2 = (npix + 31) / 32;
      if (g2 > 16384) g2 = 16384;
      FRL_LAUNCH(smooth_bwd_bf16r4_kernel, dim3((unsigned)g2), dim3(256), 0, st, (const bf16*)a0, (const bf16*)a1, (const bf16*)a2,
                 (const bf16*)a3, (bf16*)o0, (bf16*)o1, (bf16*)o2, (const bf16*)o3, B, H, W, dil);
      return frl_check_launch("edge_smooth_stencil_bwd");
    }
  }
#define SM_CASE(RR)                                                                                                     \
  if (R == RR) {                                                                                                        \
    if (fwd) FRL_LAUNCH((smooth_fwd_kernel<T, V, RR>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)a0, (const T*)a1, \
                                (const T*)a2, (T*)o0, (T*)o1, (T*)o2, (T*)o3, B, H, W, C, dil, tpp);                     \
    else FRL_LAUNCH((smooth_bwd_kernel<T, V, RR>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)a0, (const T*)a1,      \
                            (const T*)a2, (const T*)a3, (T*)o0, (T*)o1, (T*)o2, (const T*)o3, B, H, W, C, dil, tpp);    \
    return frl_check_launch("edge_smooth_stencil");                                                                     \
  }
  SM_CASE(1) SM_CASE(2) SM_CASE(4)
#undef SM_CASE
  return frl_fail(-2, "edge_smooth: rank must be 1, 2 or 4");
}

extern "C" {

// x [B][H][W][C] -> g [B][H][W][2C] = cat[sobel_x(x), sobel_y(x)]
int frl_sobel_fwd(const void* x, void* g, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && C % 4 == 0)
    FRL_LAUNCH((sobel_fwd_kernel<float, 4>), dim3(st_grid((int64_t)B * H * W * (C / 4))), dim3(256), 0, stream, (const float*)x, (float*)g, B, H, W, C);
  else if (dtype == FRL_BF16 && C % 8 == 0)
    FRL_LAUNCH((sobel_fwd_kernel<bf16, 8>), dim3(st_grid((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream, (const bf16*)x, (bf16*)g, B, H, W, C);
  else return frl_fail(-2, "sobel: C must be a multiple of 8 (bf16) / 4 (f32)");
  return frl_check_launch("sobel_fwd");
}

static int sobel_bwd_launch(const void* dg, void* dx, const void* dx_add, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && C % 4 == 0)
    FRL_LAUNCH((sobel_bwd_kernel<float, 4>), dim3(st_grid((int64_t)B * H * W * (C / 4))), dim3(256), 0, stream, (const float*)dg, (float*)dx,
               (const float*)dx_add, B, H, W, C);
  else if (dtype == FRL_BF16 && C % 8 == 0)
    FRL_LAUNCH((sobel_bwd_kernel<bf16, 8>), dim3(st_grid((int64_t)B * H * W * (C / 8))), dim3(256), 0, stream, (const bf16*)dg, (bf16*)dx,
               (const bf16*)dx_add, B, H, W, C);
  else return frl_fail(-2, "sobel: C must be a multiple of 8 (bf16) / 4 (f32)");
  return frl_check_launch("sobel_bwd");
}

int frl_sobel_bwd(const void* dg, void* dx, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  return sobel_bwd_launch(dg, dx, nullptr, B, H, W, C, dtype, stream);
}

// dx = sobel^T(dg) + dx_add  (dx_add [B][H][W][C], may be null)
int frl_sobel_bwd_add(const void* dg, void* dx, const void* dx_add, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  return sobel_bwd_launch(dg, dx, dx_add, B, H, W, C, dtype, stream);
}

// x [P][C]; a_logit [P][8*R] (channel k*R+r); b_logit [P][C*R] (channel c*R+r).
// Outputs: smoothed, residual [P][C]; a_soft, b_soft (softmaxed maps, saved for backward).  4 directions x 2 scales.
int frl_edge_smooth_stencil_fwd(const void* x, const void* a_logit, const void* b_logit, void* smoothed, void* residual, void* a_soft,
                                void* b_soft, int B, int H, int W, int C, int R, int coarse_dilation, int dtype, hipStream_t stream) {
  if (dtype == FRL_F32 && C % 4 == 0)
    return smooth_dispatch<float, 4>(true, x, a_logit, b_logit, nullptr, smoothed, residual, a_soft, b_soft, B, H, W, C, R, coarse_dilation, stream);
  if (dtype == FRL_BF16 && C % 8 == 0)
    return smooth_dispatch<bf16, 8>(true, x, a_logit, b_logit, nullptr, smoothed, residual, a_soft, b_soft, B, H, W, C, R, coarse_dilation, stream);
  return frl_fail(-2, "edge_smooth: C must be a multiple of 8 (bf16) / 4 (f32)");
}

// d_smoothed -> dx (stencil part only), d a_logit, d b_logit
int frl_edge_smooth_stencil_bwd(const void* d_smoothed, const void* x, const void* a_soft, const void* b_soft, void* dx, void* da_logit,
                                void* db_logit, const void* dx_add, int B, int H, int W, int C, int R, int coarse_dilation, int dtype,
                                hipStream_t stream) {
  void* add = const_cast<void*>(dx_add);                         // optional [B][H][W][C] term added to dx in the kernel's store
  if (dtype == FRL_F32 && C % 4 == 0)
    return smooth_dispatch<float, 4>(false, d_smoothed, x, a_soft, b_soft, dx, da_logit, db_logit, add, B, H, W, C, R, coarse_dilation, stream);
  if (dtype == FRL_BF16 && C % 8 == 0)
    return smooth_dispatch<bf16, 8>(false, d_smoothed, x, a_soft, b_soft, dx, da_logit, db_logit, add, B, H, W, C, R, coarse_dilation, stream);
  return frl_fail(-2, "edge_smooth: C must be a multiple of 8 (bf16) / 4 (f32)");
}

}  // extern "C"
