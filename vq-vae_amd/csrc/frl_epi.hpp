// Epilogues of the weight-gradient slab reductions (frl_reduce.hpp): epi(i, s) receives column i of the summed slab and scatters it into
// the gradient tensors in the reference's layouts.  They live in one header because a train step may DEFER these reductions
// (frl_defer_begin / frl_defer_flush, defer.hip): every deferred job carries its epilogue as plain bytes and ONE launch at the end of
// the backward pass runs them all, instead of one ~5 us launch behind every weight-gradient kernel.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// fused TCN block backward (tcn_hot.hip, tcn_hot_bwd3.hip, tcn_hot_bwd4.hip, tcn_fused.hip): slab = [3][64][64] conv taps | [64][64] gate |
// [64] d conv bias | [64] d gate bias | [64] d gamma | [64] d beta
struct ThEpi {
  float *dWc, *dWg, *dbc, *dbg, *dgam, *dbet;
  __device__ void operator()(int64_t i, float s) const {
    if (i < 3 * 4096) {
      const int k = (int)(i / 4096), co = (int)((i % 4096) / 64), ci = (int)(i % 64);
      dWc[(co * 64 + ci) * 3 + k] = s;
    } else if (i < 4 * 4096) {
      dWg[i - 3 * 4096] = s;
    } else {
      const int j = (int)(i - 4 * 4096);
      if (j < 64) dbc[j] = s; else if (j < 128) dbg[j - 64] = s; else if (j < 192) dgam[j - 128] = s; else dbet[j - 192] = s;
    }
  }
};

// 3x3 convolution weight gradient (conv3x3.hip, conv3x3_wgrad.hip): slab = [9 taps][OCT out channels][Cin] | [OCT] bias
struct C3Epi {
  int OCT, Cin9, oc_base, Cout; float* dW; float* dB;
  __device__ void operator()(int64_t i, float s) const {
    if (i < (int64_t)OCT * Cin9) {
      const int Cin = Cin9 / 9;
      const int tap = (int)(i / ((int64_t)OCT * Cin)), rem = (int)(i % ((int64_t)OCT * Cin));
      const int ocl = rem / Cin, ic = rem % Cin;
      if (oc_base + ocl < Cout) dW[(int64_t)(oc_base + ocl) * Cin9 + ic * 9 + tap] = s;
    } else if (dB != nullptr) {
      const int ocl = (int)(i - (int64_t)OCT * Cin9);
      if (oc_base + ocl < Cout) dB[oc_base + ocl] = s;
    }
  }
};

// fused decoder + L2 backward (dec_fused.hip): slab = dW2 [F][Hd] | dW1 [Hd][CZP] (latent width padded to CZP) | db2 [F] | db1 [Hd]
struct DecEpi {
  float *dW2, *dW1, *db2, *db1; int Cz, CZP, F, Hd;
  __device__ void operator()(int64_t i, float s) const {
    if (i < F * Hd) { dW2[i] = s; return; }
    i -= F * Hd;
    if (i < Hd * CZP) { const int hh = (int)(i / CZP), c = (int)(i % CZP); if (c < Cz) dW1[hh * Cz + c] = s; return; }
    i -= Hd * CZP;
    if (i < F) db2[i] = s; else db1[i - F] = s;
  }
};

// fused type encoder backward (enc_fused.hip): slab = dW2 [64][128] | dW1 [128][64] | d beta2 [64] | d gamma2 [64] | d beta1 [128] | d gamma1 [128]
struct EncEpi {
  float *dw2, *dw1, *db2, *dg2, *db1, *dg1;
  __device__ void operator()(int64_t i, float s) const {
    if (i < 64 * 128) { dw2[i] = s; return; }
    i -= 64 * 128;
    if (i < 128 * 64) { dw1[i] = s; return; }
    i -= 128 * 64;
    if (i < 64) db2[i] = s;
    else if (i < 128) dg2[i - 64] = s;
    else if (i < 256) db1[i - 128] = s;
    else dg1[i - 256] = s;
  }
};

// fused FiLM backward (film_fused.hip)
struct FilmEpi {
  float *dw1g, *dw1b, *dw2g, *dw2b, *db1g, *db1b, *db2g, *db2b;
  __device__ void operator()(int64_t i, float s) const {
    if (i < 2048) { dw1g[i] = s; return; }
    if (i < 4096) { dw1b[i - 2048] = s; return; }
    if (i < 4480) { dw2g[i - 4096] = s; return; }
    if (i < 4864) { dw2b[i - 4480] = s; return; }
    if (i < 4896) { db1g[i - 4864] = s; return; }
    if (i < 4928) { db1b[i - 4896] = s; return; }
    if (i < 4940) { db2g[i - 4928] = s; return; }
    db2b[i - 4940] = s;
  }
};

// fused mixing heads backward (smooth_fused.hip): slab = dW_b [NB][HID] | dW_a [NA][HID] | db_b [NB] | db_a [NA]
struct ShEpi {
  float *dwb, *dwa, *dbb, *dba; int NB, NA, HID;
  __device__ void operator()(int64_t i, float s) const {
    if (i < NB * HID) { dwb[i] = s; return; }
    i -= NB * HID;
    if (i < NA * HID) { dwa[i] = s; return; }
    i -= NA * HID;
    if (i < NB) dbb[i] = s; else dba[i - NB] = s;
  }
};

// 1x1 convolution weight gradient (pw_wgrad.hip): slab = [Cout][Cin] | [Cout] bias
struct WgradEpi {
  float* dW; int64_t dso, dsi; float* dB; int Cout, Cin, accumulate_bias;
  __device__ void operator()(int64_t i, float s) const {
    if (i < (int64_t)Cout * Cin) {
      const int oc = (int)(i / Cin), ic = (int)(i % Cin);
      dW[oc * dso + ic * dsi] = s;
    } else if (dB != nullptr) {
      const int oc = (int)(i - (int64_t)Cout * Cin);
      if (accumulate_bias) dB[oc] += s; else dB[oc] = s;
    }
  }
};

// codebook gradient of the vector quantizer (vq.hip): per-code sums S[k] = sum_{idx = k} z  ->  g_E = ce (count_k e_k - S_k); the per-code
// sums themselves (sums_out) feed the EMA update inside the same call, so the job is deferrable only without them; the epilogue READS
// counts, E and the upstream scale when it runs: a caller that defers keeps those three alive and unchanged until the flush
struct CodeEpi {
  const float* E; const int32_t* counts; const float* gscale; float ce_base; int d, bf; float* gE; float* sums_out;
  __device__ void operator()(int64_t i, float s) const {
    if (sums_out) sums_out[i] = s;
    if (gE) {
      const float ce = ce_base * (gscale ? gscale[1] : 1.f);
      const float ev = bf ? (float)(__bf16)E[i] : E[i];
      gE[i] = ce * ((float)counts[i / d] * ev - s);
    }
  }
};

// Kind tags of the deferrable epilogues (0 = not deferrable: the result is consumed inside the same C-ABI call)
template <class Epi> struct FrlEpiKind { static constexpr int id = 0; };
template <> struct FrlEpiKind<ThEpi> { static constexpr int id = 1; };
template <> struct FrlEpiKind<C3Epi> { static constexpr int id = 2; };
template <> struct FrlEpiKind<DecEpi> { static constexpr int id = 3; };
template <> struct FrlEpiKind<EncEpi> { static constexpr int id = 4; };
template <> struct FrlEpiKind<FilmEpi> { static constexpr int id = 5; };
template <> struct FrlEpiKind<ShEpi> { static constexpr int id = 6; };
template <> struct FrlEpiKind<WgradEpi> { static constexpr int id = 7; };
template <> struct FrlEpiKind<CodeEpi> { static constexpr int id = 8; };
