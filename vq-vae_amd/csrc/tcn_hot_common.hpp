// Shared pieces of the hot-configuration GatedResidualBlock kernels (tcn_hot.hip: forward + 8-wave backward with a Dropout1d mask or a
// ragged pixel count; tcn_hot_bwd3.hip: the backward of the measured configuration).  Reference: frl/models/tcn.py:78-111.
#pragma once
#include "tcn_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include "frl_reduce.hpp"

#define TH_T 5
typedef bf16x8 frag8;

template <int DIL> __device__ __forceinline__ constexpr bool th_valid(int t, int k) {
  return t + (k - 1) * DIL >= 0 && t + (k - 1) * DIL < TH_T;
}

__device__ __forceinline__ frag8 th_pack8(const float (&v)[8]) {
  frag8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
  return o;
}

// slab layout per workgroup (floats): [3][64][64] conv taps | [64][64] gate | [64] dbc | [64] dbg | [64] dgamma | [64] dbeta
#define TH_SLAB (4 * 64 * 64 + 4 * 64)

// packed weight image written by tcn_hot_pack_kernel: conv taps [3][4][2][64] | gate [4][2][64] | gate^T | conv^T taps (16-byte fragments)
static constexpr size_t TH_PACK_BYTES = (size_t)64 * 64 * sizeof(frag8);

static inline unsigned th_bwd_grid(int64_t npix) {
  int64_t g = (npix + 63) / 64;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// tcn_hot_bwd3.hip: backward without mask for HW % 64 == 0 (64-pixel tiles never straddle a sample); same slab layout as tcn_hot_bwd2_kernel
bool th_bwd3_supported(int64_t npix, int HW);
int th_bwd3_launch(int dilation, const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg,
                   void* dx, float* slab, unsigned grid, int64_t npix, int HW, float eps, hipStream_t st);
// tcn_hot_bwd4.hip: the same backward cut into two independent 4-wave subgroups per workgroup (32-pixel tiles); same conditions and slabs
bool th_bwd4_supported(int64_t npix, int HW);
int th_bwd4_launch(int dilation, const void* x, const void* dy, const frag8* pk, const float* bc, const float* gw, const float* gb, const float* bg,
                   void* dx, float* slab, unsigned grid, int64_t npix, int HW, float eps, hipStream_t st, const frag8* whp = nullptr, int chd = 0);
