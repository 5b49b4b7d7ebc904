// FiLM conditioning of the phase path, fused for the hot configuration (bf16, 64 conditioning channels, hidden 32, 12 target channels):
//   gamma = W2g relu(W1g z_type + b1g) + b2g,  beta = W2b relu(W1b z_type + b1b) + b2b      frl/models/conditioning.py:55-67,95-100
//   z_phase[b,t,p,:] = gamma[b,p,:] * h[b,t,p,:] + beta[b,p,:]   (broadcast over T)          frl/models/representation.py:369-372
// Forward: ONE launch instead of four 1x1 convolutions + the modulation; the two hidden layers are one 64-row GEMM on the matrix cores
//   (rows 0-31 gamma net, 32-63 beta net), the output layers one 16-row MFMA each with rows ordered so that a lane holds gamma[c] and
//   beta[c] of the same four channels, which are the four channels of h it loads (8-byte accesses on the 12-channel rows).
// Backward: ONE launch: d h = gamma * d z, d gamma = sum_t d z * h, d beta = sum_t d z in registers; the hidden layer is recomputed,
//   d hidden = W2^T d(gamma | beta) .* relu' on the matrix cores; all eight parameter gradients are contracted over pixels inside the
//   kernel (LDS tiles + ds_read_b64_tr_b16, per-workgroup slabs), as dec_fused.hip does.  z_type is a stop-gradient input
//   (representation.py:350-351): no d z_type.
// Replaces per step: 4 pw_conv + film_fwd launches forward; film_bwd + 2 pw_conv (bwd-data) + 4 pw_wgrad + 4 slab reductions backward.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include "frl_reduce.hpp"

typedef bf16x8 frag8;
#define FF_CZ 64
#define FF_HID 32
#define FF_C 12
#define FF_FR_W1 (4 * 2 * 64)    // [gamma blocks 0,1 | beta blocks 2,3][2 k-steps][64]
#define FF_FR_W2 64              // one 16-row block per net, one k-step (the net's 32 hidden units)
#define FF_FR_W2T (2 * 64)       // per net: 32 hidden rows (2 blocks) x one k-step over the 12 (+4) channels, k-elements e < 4 <-> channel 4 kc + e
#define FF_FR_FWD (FF_FR_W1 + 2 * FF_FR_W2)
#define FF_FR_BWD (FF_FR_FWD + 2 * FF_FR_W2T)
#define FF_SLAB (2 * FF_HID * FF_CZ + 2 * FF_C * FF_HID + 2 * FF_HID + 2 * FF_C)     // 4952 = the layer's parameter count

// transposed output layer: row = hidden unit, k-element e of lane quarter kc <-> channel 4 kc + e (e < 4), zero beyond
__device__ __forceinline__ void ff_pack_k4(frag8* __restrict__ dst, const float* __restrict__ W2, int tid, int nthreads) {
  for (int i = tid; i < FF_FR_W2T; i += nthreads) {
    const int lane = i & 63, mb = i >> 6;
    const int r = lane & 15, kc = lane >> 4;
    const int unit = 8 * (r >> 2) + 4 * mb + (r & 3);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 4 * kc + e;
      v[e] = (e < 4 && c < FF_C) ? (bf16)W2[c * FF_HID + unit] : (bf16)0.f;
    }
    dst[i] = v;
  }
}

__global__ void ff_pack_kernel(frag8* __restrict__ dst, const float* __restrict__ W1g, const float* __restrict__ W1b, const float* __restrict__ W2g,
                               const float* __restrict__ W2b, int bwd) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  pack_weights_lds<bf16, 2>(dst, W1g, FF_HID, FF_CZ, 2, FF_CZ, 1, tid, nt);
  pack_weights_lds<bf16, 2>(dst + 4 * 64, W1b, FF_HID, FF_CZ, 2, FF_CZ, 1, tid, nt);
  pack_weights_lds<bf16, 1>(dst + FF_FR_W1, W2g, FF_C, FF_HID, 1, FF_HID, 1, tid, nt);
  pack_weights_lds<bf16, 1>(dst + FF_FR_W1 + FF_FR_W2, W2b, FF_C, FF_HID, 1, FF_HID, 1, tid, nt);
  if (!bwd) return;
  ff_pack_k4(dst + FF_FR_FWD, W2g, tid, nt);
  ff_pack_k4(dst + FF_FR_FWD + FF_FR_W2T, W2b, tid, nt);
}

static const frag8* ff_packed(const float* w1g, const float* w1b, const float* w2g, const float* w2b, int bwd, frag8* ws_pk, hipStream_t st) {
  FrlPackJob jobs[6];
  size_t off = 0;
  jobs[0] = frl_pack_job_pw(w1g, off, FRL_BF16, 2, FF_HID, FF_CZ, 2, FF_CZ, 1);
  off += (size_t)4 * 64 * sizeof(frag8);
  jobs[1] = frl_pack_job_pw(w1b, off, FRL_BF16, 2, FF_HID, FF_CZ, 2, FF_CZ, 1);
  off += (size_t)4 * 64 * sizeof(frag8);
  jobs[2] = frl_pack_job_pw(w2g, off, FRL_BF16, 1, FF_C, FF_HID, 1, FF_HID, 1);
  off += (size_t)FF_FR_W2 * sizeof(frag8);
  jobs[3] = frl_pack_job_pw(w2b, off, FRL_BF16, 1, FF_C, FF_HID, 1, FF_HID, 1);
  off += (size_t)FF_FR_W2 * sizeof(frag8);
  int n = 4;
  if (bwd) {
    jobs[4] = frl_pack_job_pw(w2g, off, FRL_BF16, 1, FF_HID, FF_C, 2, 1, FF_HID);
    jobs[4].kind = FRL_PACK_PW_K4;
    off += (size_t)FF_FR_W2T * sizeof(frag8);
    jobs[5] = frl_pack_job_pw(w2b, off, FRL_BF16, 1, FF_HID, FF_C, 2, 1, FF_HID);
    jobs[5].kind = FRL_PACK_PW_K4;
    off += (size_t)FF_FR_W2T * sizeof(frag8);
    n = 6;
  }
  bool hit = false;
  frag8* pk = ws_pk;
  if (void* img = frl_pack_cached(jobs, n, off, &hit)) pk = (frag8*)img;
  if (!hit) FRL_LAUNCH(ff_pack_kernel, dim3(8), dim3(256), 0, st, pk, w1g, w1b, w2g, w2b, bwd);
  return pk;
}

// hidden layer of both nets for one 16-pixel tile: ha[mb] pre-activation of units 8 kc + 4 (mb & 1) + reg of net mb >> 1
__device__ __forceinline__ void ff_hidden(f32x4 (&ha)[4], const LQTile<bf16, 2>& zt, const frag8* __restrict__ w1, const float* __restrict__ tb,
                                          int lane, int kc) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(tb + 32 * (mb >> 1) + 8 * kc + 4 * (mb & 1));
    acc = mfma16(w1[(mb * 2 + 0) * 64 + lane], zt.f[0], acc);
    ha[mb] = mfma16(w1[(mb * 2 + 1) * 64 + lane], zt.f[1], acc);
  }
}
__device__ __forceinline__ frag8 ff_relu_frag(const f32x4& a, const f32x4& b) {
  return frag8{(bf16)fmaxf(a[0], 0.f), (bf16)fmaxf(a[1], 0.f), (bf16)fmaxf(a[2], 0.f), (bf16)fmaxf(a[3], 0.f),
               (bf16)fmaxf(b[0], 0.f), (bf16)fmaxf(b[1], 0.f), (bf16)fmaxf(b[2], 0.f), (bf16)fmaxf(b[3], 0.f)};
}

__device__ __forceinline__ void ff_tables(float* tb, const float* b1g, const float* b1b, const float* b2g, const float* b2b, int tid, int nthreads) {
  for (int i = tid; i < 96; i += nthreads) {
    float v = 0.f;
    if (i < 32) v = b1g[i];
    else if (i < 64) v = b1b[i - 32];
    else if (i < 80) v = (i - 64) < FF_C ? b2g[i - 64] : 0.f;
    else v = (i - 80) < FF_C ? b2b[i - 80] : 0.f;
    tb[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void film_fused_fwd_kernel(const bf16* __restrict__ ZT, const bf16* __restrict__ Hh, const frag8* __restrict__ Wpk,
                                                             const float* __restrict__ b1g, const float* __restrict__ b1b,
                                                             const float* __restrict__ b2g, const float* __restrict__ b2b, bf16* __restrict__ Z,
                                                             bf16* __restrict__ GAM, bf16* __restrict__ BET, int64_t npix, int HW, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* w1 = reinterpret_cast<frag8*>(smem);
  frag8* w2g = w1 + FF_FR_W1;
  frag8* w2b = w2g + FF_FR_W2;
  float* tb = reinterpret_cast<float*>(w2b + FF_FR_W2);       // b1g[32] | b1b[32] | b2g[16] | b2b[16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  copy_frags_lds<bf16>(w1, Wpk, FF_FR_FWD, tid, 256);
  ff_tables(tb, b1g, b1b, b2g, b2b, tid, 256);
  __syncthreads();
  const int64_t ntile = (npix + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    const int64_t p = tile * 16 + px;
    const bool inb = p < npix;
    const int64_t pc = inb ? p : npix - 1;
    LQTile<bf16, 2> zt;
    lq_load<bf16, 2>(zt, ZT, pc, FF_CZ, kc, true);
    f32x4 ha[4];
    ff_hidden(ha, zt, w1, tb, lane, kc);
    const frag8 hg = ff_relu_frag(ha[0], ha[1]), hb = ff_relu_frag(ha[2], ha[3]);
    f32x4 g = mfma16(w2g[lane], hg, *reinterpret_cast<const f32x4*>(tb + 64 + 4 * kc));
    f32x4 bt = mfma16(w2b[lane], hb, *reinterpret_cast<const f32x4*>(tb + 80 + 4 * kc));
    if (!inb || kc == 3) continue;                              // (channels 12..15 are padding)
    const bf16x4 gq = bf16x4{(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
    const bf16x4 bq = bf16x4{(bf16)bt[0], (bf16)bt[1], (bf16)bt[2], (bf16)bt[3]};
    *reinterpret_cast<bf16x4*>(GAM + p * FF_C + 4 * kc) = gq;
    *reinterpret_cast<bf16x4*>(BET + p * FF_C + 4 * kc) = bq;
    // the modulation uses gamma / beta as they are stored (bf16): the backward recomputes exactly these
    const int64_t b = p / HW, hw = p - b * HW;
    const bf16* hp = Hh + ((b * T) * HW + hw) * FF_C + 4 * kc;
    bf16* zp = Z + ((b * T) * HW + hw) * FF_C + 4 * kc;
    for (int t = 0; t < T; ++t) {
      const bf16x4 hv = *reinterpret_cast<const bf16x4*>(hp + (int64_t)t * HW * FF_C);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)fmaf((float)gq[e], (float)hv[e], (float)bq[e]);
      *reinterpret_cast<bf16x4*>(zp + (int64_t)t * HW * FF_C) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// 4 waves x 16 pixels per round.  slab (floats): dW1g [32][64] | dW1b [32][64] | dW2g [12][32] | dW2b [12][32] | db1g [32] | db1b [32] | db2g [12] | db2b [12]
#define FFB_R 64
#define FFB_P64 (64 + 8)
#define FFB_P32 (32 + 8)

__device__ __forceinline__ bf16x8 ff_tr_frag(const bf16* tile, int pitch, int pix0, int ch0, int r16) {
  const bf16* a0 = tile + (pix0 + (r16 >> 2)) * pitch + ch0 + 4 * (r16 & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitch));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) void film_fused_bwd_kernel(const bf16* __restrict__ ZT, const bf16* __restrict__ Hh, const bf16* __restrict__ DZ,
                                                             const frag8* __restrict__ Wpk, const float* __restrict__ b1g,
                                                             const float* __restrict__ b1b, const float* __restrict__ b2g,
                                                             const float* __restrict__ b2b, bf16* __restrict__ DH, float* __restrict__ slab,
                                                             int64_t npix, int HW, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  frag8* w1 = reinterpret_cast<frag8*>(smem);
  frag8* w2g = w1 + FF_FR_W1;
  frag8* w2b = w2g + FF_FR_W2;
  frag8* w2gT = w2b + FF_FR_W2;
  frag8* w2bT = w2gT + FF_FR_W2T;
  float* tb = reinterpret_cast<float*>(w2bT + FF_FR_W2T);
  bf16* t_z = reinterpret_cast<bf16*>(tb + 96);               // [R][P64]  z_type
  bf16* t_hid = t_z + FFB_R * FFB_P64;                        // [R][P64]  relu(hidden): gamma units 0..31 | beta units 32..63
  bf16* t_dhid = t_hid + FFB_R * FFB_P64;                     // [R][P64]  d hidden (pre-activation)
  bf16* t_dgb = t_dhid + FFB_R * FFB_P64;                     // [R][P32]  d gamma at 0..11, d beta at 16..27
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4, r16 = px;
  const int prow = wave * 16 + px;
  copy_frags_lds<bf16>(w1, Wpk, FF_FR_BWD, tid, 256);
  ff_tables(tb, b1g, b1b, b2g, b2b, tid, 256);
  __syncthreads();
  const bf16 one = (bf16)1.f, zero = (bf16)0.f;
  const bf16x8 ones = (r16 == 0) ? bf16x8{one, one, one, one, one, one, one, one} : bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};
  const bf16x8 zeros = bf16x8{zero, zero, zero, zero, zero, zero, zero, zero};
  // ownership: wave w holds d W1 rows [16 (w & 1), +16) of net w >> 1 (all 64 columns), the d W2 tile (net w >> 1, hidden columns
  // [16 (w & 1), +16)) and, for even w, d b2 of that net
  f32x4 gW1[4], gb1, gW2, gb2;
#pragma unroll
  for (int i = 0; i < 4; ++i) gW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  gb1 = gW2 = gb2 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t nround = (npix + FFB_R - 1) / FFB_R;
  for (int64_t rd = blockIdx.x; rd < nround; rd += gridDim.x) {
    const int64_t p = rd * FFB_R + prow;
    const bool inb = p < npix;
    const int64_t pc = inb ? p : npix - 1;
    LQTile<bf16, 2> zt;
    lq_load<bf16, 2>(zt, ZT, pc, FF_CZ, kc, true);
    f32x4 ha[4];
    ff_hidden(ha, zt, w1, tb, lane, kc);
    const frag8 hg = ff_relu_frag(ha[0], ha[1]), hb = ff_relu_frag(ha[2], ha[3]);
    const f32x4 g = mfma16(w2g[lane], hg, *reinterpret_cast<const f32x4*>(tb + 64 + 4 * kc));
    float gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gq[e] = (float)(bf16)g[e];      // gamma as the forward stored and applied it
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    if (inb && kc < 3) {
      const int64_t b = p / HW, hw = p - b * HW;
      const int64_t row0 = ((b * T) * HW + hw) * FF_C + 4 * kc;
      for (int t = 0; t < T; ++t) {
        const int64_t o = row0 + (int64_t)t * HW * FF_C;
        const bf16x4 dz = *reinterpret_cast<const bf16x4*>(DZ + o);
        const bf16x4 hv = *reinterpret_cast<const bf16x4*>(Hh + o);
        bf16x4 dh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = (float)dz[e];
          dg[e] = fmaf(d, (float)hv[e], dg[e]);
          db[e] += d;
          dh[e] = (bf16)(gq[e] * d);
        }
        *reinterpret_cast<bf16x4*>(DH + o) = dh;
      }
    }
    const bf16x4 dgq = bf16x4{(bf16)dg[0], (bf16)dg[1], (bf16)dg[2], (bf16)dg[3]};
    const bf16x4 dbq = bf16x4{(bf16)db[0], (bf16)db[1], (bf16)db[2], (bf16)db[3]};
    const frag8 dgf = frag8{dgq[0], dgq[1], dgq[2], dgq[3], zero, zero, zero, zero};
    const frag8 dbf = frag8{dbq[0], dbq[1], dbq[2], dbq[3], zero, zero, zero, zero};
    f32x4 dha[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      dha[m] = mfma16(w2gT[m * 64 + lane], dgf, f32x4{0.f, 0.f, 0.f, 0.f});
      dha[2 + m] = mfma16(w2bT[m * 64 + lane], dbf, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    frag8 dhf[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 8; ++e) dhf[n][e] = (bf16)(ha[2 * n + (e >> 2)][e & 3] > 0.f ? dha[2 * n + (e >> 2)][e & 3] : 0.f);
    {
      bf16x8* zp = reinterpret_cast<bf16x8*>(t_z + prow * FFB_P64 + 16 * kc);
      zp[0] = inb ? zt.f[0] : zeros;
      zp[1] = inb ? zt.f[1] : zeros;
      *reinterpret_cast<bf16x8*>(t_hid + prow * FFB_P64 + 8 * kc) = hg;
      *reinterpret_cast<bf16x8*>(t_hid + prow * FFB_P64 + 32 + 8 * kc) = hb;
      *reinterpret_cast<bf16x8*>(t_dhid + prow * FFB_P64 + 8 * kc) = dhf[0];
      *reinterpret_cast<bf16x8*>(t_dhid + prow * FFB_P64 + 32 + 8 * kc) = dhf[1];
      *reinterpret_cast<bf16x4*>(t_dgb + prow * FFB_P32 + 4 * kc) = dgq;
      *reinterpret_cast<bf16x4*>(t_dgb + prow * FFB_P32 + 16 + 4 * kc) = dbq;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < FFB_R / 32; ++ks) {
      const int pix0 = ks * 32 + 8 * kc;
      const bf16x8 ad = ff_tr_frag(t_dhid, FFB_P64, pix0, 16 * wave, r16);
#pragma unroll
      for (int i = 0; i < 4; ++i) gW1[i] = mfma16(ad, ff_tr_frag(t_z, FFB_P64, pix0, 16 * i, r16), gW1[i]);
      gb1 = mfma16(ad, ones, gb1);
      const bf16x8 ag = ff_tr_frag(t_dgb, FFB_P32, pix0, 16 * (wave >> 1), r16);
      gW2 = mfma16(ag, ff_tr_frag(t_hid, FFB_P64, pix0, 16 * wave, r16), gW2);
      if ((wave & 1) == 0) gb2 = mfma16(ag, ones, gb2);
    }
    __syncthreads();
  }
  float* my = slab + (int64_t)blockIdx.x * FF_SLAB;
  const int net = wave >> 1, row0 = 16 * (wave & 1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) my[net * 2048 + (row0 + 4 * kc + r) * FF_CZ + 16 * i + r16] = gW1[i][r];
  if (r16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) my[4864 + 32 * net + row0 + 4 * kc + r] = gb1[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * kc + r;
    if (c < FF_C) {
      my[4096 + 384 * net + c * FF_HID + row0 + r16] = gW2[r];
      if (r16 == 0 && (wave & 1) == 0) my[4928 + 12 * net + c] = gb2[r];
    }
  }
}


extern "C" {

int frl_film_fused_supported(int cond_dim, int hidden, int target_dim, int dtype) {
  return (dtype == FRL_BF16 && cond_dim == FF_CZ && hidden == FF_HID && target_dim == FF_C) ? 1 : 0;
}

size_t frl_film_fused_workspace_bytes(void) { return (size_t)FF_FR_BWD * sizeof(frag8) + 256 + (size_t)768 * FF_SLAB * sizeof(float); }

// z_type [B][HW][64] (stop-gradient input), h [B][T][HW][12] bf16; w1g / w1b [32][64], w2g / w2b [12][32] f32 with their biases
// -> z [B][T][HW][12] = gamma * h + beta, gamma / beta [B][HW][12] bf16
int frl_film_fused_fwd(const void* z_type, const void* h, const float* w1g, const float* b1g, const float* w2g, const float* b2g, const float* w1b,
                       const float* b1b, const float* w2b, const float* b2b, void* z, void* gamma, void* beta, int B, int T, int HW, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  const int64_t npix = (int64_t)B * HW;
  if (npix <= 0 || T <= 0) return frl_fail(-2, "film_fused_fwd: empty input");
  if (ws_bytes < frl_film_fused_workspace_bytes()) return frl_fail(-4, "film_fused_fwd: workspace too small");
  const frag8* pk = ff_packed(w1g, w1b, w2g, w2b, 0, reinterpret_cast<frag8*>(ws), stream);
  int64_t g = ((npix + 15) / 16 + 3) / 4;
  if (g > 2048) g = 2048;
  const size_t lds = (size_t)FF_FR_FWD * sizeof(frag8) + 96 * sizeof(float);
  FRL_LAUNCH(film_fused_fwd_kernel, dim3((unsigned)g), dim3(256), lds, stream, (const bf16*)z_type, (const bf16*)h, pk, b1g, b1b, b2g, b2b, (bf16*)z,
             (bf16*)gamma, (bf16*)beta, npix, HW, T);
  return frl_check_launch("film_fused_fwd");
}

// d z [B][T][HW][12] -> d h (same shape) and the eight parameter gradients (float32, the parameters' shapes)
int frl_film_fused_bwd(const void* z_type, const void* h, const void* dz, const float* w1g, const float* b1g, const float* w2g, const float* b2g,
                       const float* w1b, const float* b1b, const float* w2b, const float* b2b, void* dh, float* dw1g, float* db1g, float* dw2g,
                       float* db2g, float* dw1b, float* db1b, float* dw2b, float* db2b, int B, int T, int HW, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
  const int64_t npix = (int64_t)B * HW;
  if (npix <= 0 || T <= 0) return frl_fail(-2, "film_fused_bwd: empty input");
  if (ws_bytes < frl_film_fused_workspace_bytes()) return frl_fail(-4, "film_fused_bwd: workspace too small");
  char* w = (char*)ws;
  const size_t pkb = ((size_t)FF_FR_BWD * sizeof(frag8) + 255) / 256 * 256;
  const frag8* pk = ff_packed(w1g, w1b, w2g, w2b, 1, reinterpret_cast<frag8*>(w), stream);
  float* slab = reinterpret_cast<float*>(w + pkb);
  int64_t g = (npix + FFB_R - 1) / FFB_R;
  if (g > 768) g = 768;                                         // 47 KB of LDS: three workgroups per CU hide each other's row loads
  const size_t lds = (size_t)FF_FR_BWD * sizeof(frag8) + 96 * sizeof(float) + (size_t)FFB_R * (3 * FFB_P64 + FFB_P32) * sizeof(bf16);
  FRL_LAUNCH(film_fused_bwd_kernel, dim3((unsigned)g), dim3(256), lds, stream, (const bf16*)z_type, (const bf16*)h, (const bf16*)dz, pk, b1g, b1b, b2g,
             b2b, (bf16*)dh, slab, npix, HW, T);
  launch_slab_reduce_deferrable<float, FilmEpi>((const float*)slab, (int)g, (int64_t)FF_SLAB, FilmEpi{dw1g, dw1b, dw2g, dw2b, db1g, db1b, db2g, db2b}, stream);
  return frl_check_launch("film_fused_bwd");
}

}  // extern "C"
