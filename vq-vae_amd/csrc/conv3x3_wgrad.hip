// Weight gradient of the 3x3 convolutions, bf16 band kernel (gfx950):
//   dW[oc][ic][dy][dx] = sum_{b,y,x} (dY .* act'(Y))[b,y,x,oc] * X[b,y+dy-1,x+dx-1,ic],   db[oc] = sum (dY .* act'(Y))[b,y,x,oc]
// Replaces autograd's conv2d weight gradient for spatial.py:247-263,296-313 (mix backbone, gate net); the generic kernel of
// conv3x3.hip stays for float32, ragged shapes and channel counts that are not multiples of 64.
//
// Work item = (8 x 32-pixel band of one image, 64 input channels); a workgroup (8 waves, one per CU, persistent) walks its items with
// everything one item ahead:
//   * the 10 x 34-pixel halo of X arrives by LDS-DMA (global_load_lds_dwordx4: no registers, 43 KB in flight per workgroup); pixels
//     outside the image are fetched from a page of zeros, so the DMA needs no predication;
//   * dY and Y of the band are prefetched into registers behind the MFMA phase of the current item, multiplied (act') and written to
//     the other dY buffer after it: ONE barrier per item;
//   * both LDS tiles are unpadded 128-byte pixel rows; the 32-byte windows of a row are permuted by a function of the pixel's COLUMN
//     (bits 1 and 3), which makes the row-major 16-byte writes, the DMA and the transposing ds_read_b64_tr_b16 fragment reads of all
//     nine taps conflict-free, and keeps every fragment address = lane constant + compile-time offset;
//   * a k-step of the contraction is one 32-pixel image row: per k-step a wave reads 2 dY^T and 9 X fragments for 18 MFMAs
//     (its 16 input channels x 32 output channels x 9 taps stay in 72 accumulator registers for the whole kernel).
// Output: per-workgroup float32 slabs in the layout of the generic kernel (tap-major), summed in a fixed order by slab_reduce_t.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include <type_traits>

#define C3V_TH 8
#define C3V_TW 32
#define C3V_HP 34                           // halo row pitch (pixels)
#define C3V_HPIX (10 * C3V_HP)              // 340 halo pixels
#define C3V_DY_BYTES (C3V_TH * C3V_TW * 128)
#define C3V_NPIECE 43                       // 1 KB DMA pieces per halo (43 x 8 pixels >= 340)
#define C3V_HALO_BYTES (C3V_NPIECE * 1024)
#define C3V_LDS (2 * C3V_DY_BYTES + 2 * C3V_HALO_BYTES)

__device__ uint4 c3v_zero_page[64];         // 1 KB of zeros: source of the halo pixels outside the image

// 32-byte window permutation of a pixel row, from the pixel's column inside its tile / halo row
__device__ __forceinline__ int c3v_f(int col) { return ((col >> 1) & 1) | (((col >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x8 c3v_tr8(const char* smem, int lo, int hi) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(smem + lo));
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(smem + hi));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_band_kernel(const bf16* __restrict__ dY, const bf16* __restrict__ Ymask, int mask_act,
                                                                 const bf16* __restrict__ X, float* __restrict__ slab, int B, int H, int W,
                                                                 int Cin, int Cout, int oc_base, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kc = lane >> 4;
  const int ib = wave & 3, og = wave >> 2;                       // 16 input channels of the 64-chunk; 32 output channels
  const int tiles_x = W / C3V_TW, tiles_y = H / C3V_TH;
  const int ntiles = B * tiles_x * tiles_y;
  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = (t_begin + tiles_per_wg) < ntiles ? (t_begin + tiles_per_wg) : ntiles;
  const int ntl = t_end > t_begin ? t_end - t_begin : 0;
  const int ncks = Cin >> 6;
  const int nitems = ntl * ncks;                                 // item i: chunk i / ntl, tile t_begin + i % ntl
  const int64_t slab_n = (int64_t)64 * Cin * 9 + 64;
  float* my = slab + (int64_t)blockIdx.x * slab_n;

  // ---- lane constants of the fragment reads: column (inside a row) of the lane's first pixel, window swizzle, 8-byte half ----
  int aoff[2][2], boff[3][2];                                    // [output block o][lo / hi], [dx][lo / hi]
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int col = 8 * kc + (r16 >> 2) + 4 * hi;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int c = 2 * (og * 2 + o) + ((r16 & 3) >> 1);
      aoff[o][hi] = col * 128 + ((c ^ (c3v_f(col) << 1)) << 4) + (r16 & 1) * 8;
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hc = col + dx;
      const int c = 2 * ib + ((r16 & 3) >> 1);
      boff[dx][hi] = hc * 128 + ((c ^ (c3v_f(hc) << 1)) << 4) + (r16 & 1) * 8;
    }
  }

  auto item_geom = [&](int i, int& b, int& y0, int& x0, int& ck) {
    const int tile = t_begin + i % ntl;
    ck = (i / ntl) << 6;
    b = tile / (tiles_x * tiles_y);
    const int tyx = tile % (tiles_x * tiles_y);
    y0 = (tyx / tiles_x) * C3V_TH;
    x0 = (tyx % tiles_x) * C3V_TW;
  };
  // ---- halo of one item by LDS-DMA: piece p (8 halo pixels = 1 KB) is issued by wave p % 8 ----
  auto dma_halo = [&](int i, int dst) {
    int b, y0, x0, ck;
    item_geom(i, b, y0, x0, ck);
    const bf16* xb = X + (((int64_t)b * H) * W) * Cin + ck;
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));                              // (opaque: the per-piece lane constants are rebuilt per item, not kept in 40 registers)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int piece = j * 8 + wave;
      if (piece < C3V_NPIECE) {                                  // (wave-uniform)
        const int chunk = piece * 64 + lane_;
        const int hp = chunk >> 3, c = chunk & 7;
        const int hr = (hp * 1928) >> 16, hc = hp - hr * C3V_HP;  // hp / 34 for hp < 344
        const int gy = y0 + hr - 1, gx = x0 + hc - 1;
        const bool ok = hp < C3V_HPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const bf16* src = ok ? xb + ((int64_t)gy * W + gx) * Cin + ((c ^ (c3v_f(hc) << 1)) << 3)
                             : reinterpret_cast<const bf16*>(c3v_zero_page) + (lane_ << 3);
        const int ldst = dst + piece * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
      }
    }
  };
  // ---- dY / Y of one item into registers (4 x 16 bytes each per thread), then masked and written to a dY buffer ----
  bf16x8 ra[4], rm[4];
  auto fetch_dy = [&](int i) {
    int b, y0, x0, ck;
    item_geom(i, b, y0, x0, ck);
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = tid_ + 512 * u;
      const int px = v >> 3, c = v & 7;
      const int64_t off = (((int64_t)b * H + y0 + (px >> 5)) * W + x0 + (px & 31)) * Cout + oc_base + c * 8;
      ra[u] = *reinterpret_cast<const bf16x8*>(dY + off);
      if (Ymask != nullptr) rm[u] = *reinterpret_cast<const bf16x8*>(Ymask + off);
    }
  };
  auto commit_dy = [&](int dst) {
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = tid_ + 512 * u;
      const int px = v >> 3, c = v & 7;
      bf16x8 val = ra[u];
      if (Ymask != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = (bf16)((float)val[e] * act_bwd_from_y((float)rm[u][e], mask_act));
      }
      *reinterpret_cast<bf16x8*>(smem + dst + px * 128 + ((c ^ (c3v_f(px & 31) << 1)) << 4)) = val;
    }
  };

  f32x4 acc[9][2], accb[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) acc[tap][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto write_chunk = [&](int ck) {                               // slab part of one input-channel chunk, then clear the accumulators
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));                              // (slab addresses are built here, not carried through the item loop)
    float* mine = my + (int64_t)(og * 32 + (lane_ >> 4) * 4) * Cin + ck + ib * 16 + (lane_ & 15);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((int64_t)tap * 64 + o * 16 + r) * Cin] = acc[tap][o][r];
        acc[tap][o] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };

  if (nitems > 0) {
    dma_halo(0, 2 * C3V_DY_BYTES);
    fetch_dy(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    commit_dy(0);
  }
  __syncthreads();
  for (int i = 0; i < nitems; ++i) {
    const int cur = i & 1;
    const int dyb = cur * C3V_DY_BYTES, hb = 2 * C3V_DY_BYTES + cur * C3V_HALO_BYTES;
    if (i + 1 < nitems) {                                        // next item: halo by DMA, dY / Y into registers
      dma_halo(i + 1, 2 * C3V_DY_BYTES + (cur ^ 1) * C3V_HALO_BYTES);
      fetch_dy(i + 1);
    }
    // (fully unrolled, the compiler shares the X fragments of a halo row between the taps of three k-steps: 104 instead of 176 LDS
    // reads per item)
    // (two copies of the fully unrolled k loop, so that the bias MFMAs of the (input chunk 0, ib == 0) waves cost the others no branch
    // inside it: a branch per k-step ends the basic block and with it the overlap of one step's LDS reads with the previous MFMAs)
    auto contract = [&](auto with_bias) {
      int r16_ = r16;
      asm volatile("" : "+v"(r16_));
      const bf16 one_ = (r16_ == 0) ? (bf16)1.f : (bf16)0.f;       // B operand that sums the k dimension into output column 0
      const bf16x8 ones = bf16x8{one_, one_, one_, one_, one_, one_, one_, one_};
#pragma unroll
      for (int ks = 0; ks < C3V_TH; ++ks) {
        bf16x8 af[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) af[o] = c3v_tr8(smem, dyb + ks * (C3V_TW * 128) + aoff[o][0], dyb + ks * (C3V_TW * 128) + aoff[o][1]);
        if constexpr (decltype(with_bias)::value) {
#pragma unroll
          for (int o = 0; o < 2; ++o) accb[o] = mfma16(af[o], ones, accb[o]);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3, dx = tap % 3;
          const int ro = hb + (ks + dy) * (C3V_HP * 128);
          const bf16x8 bfr = c3v_tr8(smem, ro + boff[dx][0], ro + boff[dx][1]);
#pragma unroll
          for (int o = 0; o < 2; ++o) acc[tap][o] = mfma16(af[o], bfr, acc[tap][o]);
        }
      }
    };
    if (i < ntl && ib == 0) contract(std::true_type{});           // (uniform) the bias gradient rides with input chunk 0
    else contract(std::false_type{});
    if (i + 1 < nitems) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // dY / Y registers and this wave's DMA pieces have landed
      commit_dy((cur ^ 1) * C3V_DY_BYTES);
    }
    if ((i + 1) % ntl == 0) write_chunk((i / ntl) << 6);         // (uniform) last tile of an input-channel chunk
    __syncthreads();
  }
  if (nitems == 0) {                                             // a workgroup without tiles still owns a slab: zeros
    for (int ck = 0; ck < Cin; ck += 64) write_chunk(ck);
  }
  if (ib == 0 && r16 == 0) {                                     // column 0 of the "ones" product holds the row sums
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) my[(int64_t)64 * Cin * 9 + (og * 2 + o) * 16 + kc * 4 + r] = accb[o][r];
  }
}

// ---- host side (called from frl_conv3x3_bwd_weight, conv3x3.hip) ----
static int g_c3v_off = 0;

bool c3v_supported(int B, int H, int W, int Cin, int Cout, int dtype) {
  return !g_c3v_off && dtype == FRL_BF16 && B > 0 && (W % C3V_TW) == 0 && (H % C3V_TH) == 0 && (Cin % 64) == 0 && (Cout % 64) == 0;
}
int c3v_workgroups(int B, int H, int W) {
  const int tiles = B * (H / C3V_TH) * (W / C3V_TW);
  return tiles < 256 ? tiles : 256;
}
int c3v_launch(const void* dy, const void* ym, int act, const void* x, float* ws, int B, int H, int W, int Cin, int Cout, int oc_base,
               hipStream_t st) {
  const int nwg = c3v_workgroups(B, H, W);
  const int tiles = B * (H / C3V_TH) * (W / C3V_TW);
  const int tpw = (tiles + nwg - 1) / nwg;
  auto kern = conv3x3_wgrad_band_kernel;
  FRL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3V_LDS));
  FRL_LAUNCH_AS("conv3x3_wgrad_kernel", kern, dim3(nwg), dim3(512), C3V_LDS, st, (const bf16*)dy, (const bf16*)ym, act, (const bf16*)x, ws, B, H, W,
                Cin, Cout, oc_base, tpw);
  return frl_check_launch("conv3x3_bwd_weight");
}

extern "C" {
// Test / A-B hook: 1 routes every 3x3 weight gradient through the generic kernel of conv3x3.hip; returns the previous setting.
int frl_conv3x3_wgrad_force_generic(int on) {
  const int was = g_c3v_off;
  g_c3v_off = on ? 1 : 0;
  return was;
}
}
