// Tile ingest: raw (time,y,x,feature) rows as they sit in the chunk store (float16 or float32, NaN = no data) -> normalised
// training rows (bf16 or f32) + per-(t,y,x) validity byte, in ONE pass over HBM.
//
// Follows the reference's per-channel normalisation presets and masking, restated as one record per feature:
//   zscore      (x - mean) / sd        sd  < 1e-8 -> 1        frl/data/loaders/builders/feature_builder.py:504-509
//   robust_iqr  (x - q50) / (q75-q25)  iqr < 1e-8 -> 1        feature_builder.py:511-518
//   minmax      (x - min) / (max-min)                         frl/data/normalization/normalization.py:172-201
//   linear_rescale ((x - in_min) / in_range) * out_range + out_min   feature_builder.py:520-531
//   clamp / none / identity: value unchanged; optional np.clip(min, max) after every preset    feature_builder.py:539-546
//   invalid pixels are set to 0 after normalisation            feature_builder.py:709-737
// The arithmetic is evaluated in float32 in exactly that operation order (true division, separately rounded multiply and add),
// so the f32 output is bit-identical to numpy's float32 evaluation of the reference formulas; the bf16 output is its RNE rounding.
// HBM-bound: F*(s_in + s_out) + 1 (+1) bytes per (t,y,x) row.
#include <hip/hip_fp16.h>

#include "frl_common.hpp"
#include "frl_host.hpp"

struct FrlNormRec {   // mirrors include/frl_hip.h
  float sub, div, mul, add, lo, hi;
  int flags;          // 1: rescale (r*mul + add)   2: clamp below at lo   4: clamp above at hi
  int pad;
};

template <typename TIN>
__device__ __forceinline__ void tp_load8(const TIN* p, float (&v)[8]);
template <>
__device__ __forceinline__ void tp_load8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <>
__device__ __forceinline__ void tp_load8<__half>(const __half* p, float (&v)[8]) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 a = *reinterpret_cast<const h8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <typename TOUT>
__device__ __forceinline__ void tp_store8(TOUT* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void tp_store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <>
__device__ __forceinline__ void tp_store8<bf16>(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// one lane = 8 consecutive features of one row; the LPR = F/8 lanes of a row sit next to each other in the wave, so the row's
// "all features finite" test is an xor-shuffle AND over LPR lanes.  The stride of the grid-stride loop is a multiple of LPR, hence a
// lane keeps its 8 feature records in registers for the whole launch.
//
// GATHER = false: raw holds the output rows in order.  GATHER = true: raw is one whole (time, cy, cx, feature) chunk of the store
// and the output rows are the `tile x tile` patches listed in `desc` ({y0, x0, h, w} per tile, h/w < tile for partial patches at
// the raster edge): the tile cut, the zero padding and the inside-raster mask happen here instead of in host-side strided copies.
struct TpGather {
  const int4* desc;   // [tiles]
  int T, CY, CX, n;   // chunk extent and tile size
  unsigned n_shift;   // log2(n) when n is a power of two, else 0xffffffff
};

template <typename TIN, typename TOUT, bool GATHER>
__global__ __launch_bounds__(256) void normalize_tiles_kernel(const TIN* __restrict__ raw, const uint8_t* __restrict__ valid,
                                                              const FrlNormRec* __restrict__ table, TOUT* __restrict__ out,
                                                              uint8_t* __restrict__ mask_out, int64_t rows, int F, int lpr_shift,
                                                              TpGather g) {
  const int lpr = 1 << lpr_shift;
  const int64_t total = rows << lpr_shift;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int f0 = ((int)i0 & (lpr - 1)) * 8;
  FrlNormRec rec[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    rec[e] = table[f0 + e];
    // disabled stages become exact no-ops so that the row loop is branch-free: x * 1 + (-0) == x bit for bit (also for x = -0),
    // max(x, -inf) == min(x, +inf) == x
    if (!(rec[e].flags & 1)) { rec[e].mul = 1.f; rec[e].add = -0.f; }
    if (!(rec[e].flags & 2)) rec[e].lo = -__builtin_inff();
    if (!(rec[e].flags & 4)) rec[e].hi = __builtin_inff();
  }
  // every lane of a wave runs the same number of iterations (the shuffles below need all of them): pad the bound to whole waves
  const int64_t total_pad = (total + 63) & ~(int64_t)63;
  for (int64_t i = i0; i < total_pad; i += (int64_t)gridDim.x * 256) {
    const bool live = i < total;
    const int64_t row = live ? (i >> lpr_shift) : 0;
    int64_t src = row;
    bool inside = true;
    if constexpr (GATHER) {
      const unsigned r32 = (unsigned)row;                            // rows < 2^31 is checked on the host for this mode
      unsigned x, y, tb;
      if (g.n_shift != 0xffffffffu) { x = r32 & (g.n - 1); y = (r32 >> g.n_shift) & (g.n - 1); tb = r32 >> (2 * g.n_shift); }
      else { x = r32 % (unsigned)g.n; const unsigned q = r32 / (unsigned)g.n; y = q % (unsigned)g.n; tb = q / (unsigned)g.n; }
      const unsigned b = tb / (unsigned)g.T, tt = tb - b * (unsigned)g.T;
      const int4 d = g.desc[b];
      inside = (int)y < d.z && (int)x < d.w;
      src = inside ? ((int64_t)tt * g.CY + d.x + (int)y) * g.CX + d.y + (int)x : 0;
    }
    float v[8];
    if (live && inside) tp_load8<TIN>(raw + src * F + f0, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    int ok = 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) ok &= (__builtin_fabsf(v[e]) <= 3.402823466e38f) ? 1 : 0;      // false for NaN and +-inf
    for (int off = 1; off < lpr; off <<= 1) ok &= __shfl_xor(ok, off, 64);
    if (!inside) ok = 0;
    if (valid != nullptr && live) ok &= valid[row] ? 1 : 0;
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)                                     // numpy rounds the product before the sum: no fma here
      float x = (v[e] - rec[e].sub) / rec[e].div;                  // correctly rounded division (hipcc default)
      const float prod = x * rec[e].mul;
      x = prod + rec[e].add;
      x = __builtin_amdgcn_fmed3f(x, rec[e].lo, rec[e].hi);         // == min(max(x, lo), hi) for lo <= hi (checked on the host)
      r[e] = ok ? x : 0.f;
    }
    if (live) {
      tp_store8<TOUT>(out + row * F + f0, r);
      if (mask_out != nullptr && f0 == 0) mask_out[row] = (uint8_t)ok;
    }
  }
}

template <bool GATHER>
static int tp_launch(const void* raw, int raw_dtype, const uint8_t* valid, const void* table, void* out, int out_dtype, uint8_t* mask_out,
                     int64_t rows, int F, TpGather g, hipStream_t stream) {
  const int lpr = F >> 3;
  if (F < 8 || F > 512 || (F & 7) || (lpr & (lpr - 1)))
    return frl_fail(-2, "normalize_tiles: feature count must be 8, 16, 32, 64, 128, 256 or 512");
  if (rows > (int64_t)1 << 40) return frl_fail(-2, "normalize_tiles: too many rows");
  int lpr_shift = 0;
  while ((1 << lpr_shift) < lpr) ++lpr_shift;
  const int64_t total = rows * lpr;
  int64_t nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  const dim3 grid((unsigned)nb), block(256);
  const FrlNormRec* tb = (const FrlNormRec*)table;
  if (raw_dtype == 0 && out_dtype == FRL_F32)
    FRL_LAUNCH((normalize_tiles_kernel<float, float, GATHER>), grid, block, 0, stream, (const float*)raw, valid, tb, (float*)out, mask_out, rows, F, lpr_shift, g);
  else if (raw_dtype == 0 && out_dtype == FRL_BF16)
    FRL_LAUNCH((normalize_tiles_kernel<float, bf16, GATHER>), grid, block, 0, stream, (const float*)raw, valid, tb, (bf16*)out, mask_out, rows, F, lpr_shift, g);
  else if (raw_dtype == 2 && out_dtype == FRL_F32)
    FRL_LAUNCH((normalize_tiles_kernel<__half, float, GATHER>), grid, block, 0, stream, (const __half*)raw, valid, tb, (float*)out, mask_out, rows, F, lpr_shift, g);
  else if (raw_dtype == 2 && out_dtype == FRL_BF16)
    FRL_LAUNCH((normalize_tiles_kernel<__half, bf16, GATHER>), grid, block, 0, stream, (const __half*)raw, valid, tb, (bf16*)out, mask_out, rows, F, lpr_shift, g);
  else return frl_fail(-2, "normalize_tiles: raw dtype must be FRL_F32 or FRL_F16, output FRL_F32 or FRL_BF16");
  return frl_check_launch("normalize_tiles");
}

extern "C" {

// raw [rows][F] of raw_dtype (FRL_F32 = 0 | FRL_F16 = 2), valid [rows] bytes or null, table [F] device records,
// out [rows][F] of out_dtype (FRL_F32 | FRL_BF16), mask_out [rows] bytes or null.
int frl_normalize_tiles(const void* raw, int raw_dtype, const uint8_t* valid, const void* table, void* out, int out_dtype,
                        uint8_t* mask_out, int64_t rows, int F, hipStream_t stream) {
  if (rows < 0) return frl_fail(-2, "normalize_tiles: negative row count");
  if (rows == 0) return 0;
  return tp_launch<false>(raw, raw_dtype, valid, table, out, out_dtype, mask_out, rows, F, TpGather{}, stream);
}

// chunk [T][CY][CX][F] of raw_dtype resident on the device; desc [ntiles] int32 {y0, x0, h, w} (device): tile b covers chunk rows
// y0..y0+h-1, columns x0..x0+w-1 (h, w <= tile; the remainder of the tile x tile patch is zero padding, mask 0).
// out [ntiles][T][tile][tile][F], mask_out [ntiles][T][tile][tile].
int frl_normalize_chunk_tiles(const void* chunk, int raw_dtype, int T, int CY, int CX, int F, const int32_t* desc, int ntiles, int tile,
                              const void* table, void* out, int out_dtype, uint8_t* mask_out, hipStream_t stream) {
  if (ntiles < 0 || T <= 0 || CY <= 0 || CX <= 0 || tile <= 0) return frl_fail(-2, "normalize_chunk_tiles: bad extents");
  if (ntiles == 0) return 0;
  const int64_t rows = (int64_t)ntiles * T * tile * tile;
  if (rows >= (int64_t)1 << 31) return frl_fail(-2, "normalize_chunk_tiles: more than 2^31 output rows in one call");
  TpGather g;
  g.desc = reinterpret_cast<const int4*>(desc);
  g.T = T; g.CY = CY; g.CX = CX; g.n = tile;
  g.n_shift = 0xffffffffu;
  if ((tile & (tile - 1)) == 0) { g.n_shift = 0; while ((1 << g.n_shift) < tile) ++g.n_shift; }
  return tp_launch<true>(chunk, raw_dtype, nullptr, table, out, out_dtype, mask_out, rows, F, g, stream);
}

}  // extern "C"
