// Weight-image cache: every conv-like call of this library first rewrites its float32 master weights into a packed MFMA fragment image
// (a tiny prologue launch per call: ~30 per train step at BASELINE configs[1], ~0.25 ms of GPU time made of launch latencies).  The
// weights change once per optimizer step, so a caller that owns the step (the trainer) keeps the images in an arena of its own and has
// ALL of them rewritten by ONE launch right behind the optimizer (frl_pack_cache_refresh); inside the step the calls find their image
// and skip the prologue.  Outside an active cache nothing changes: the image is packed into the call's workspace as before.
//
// Protocol (include/frl_hip.h): h = frl_pack_cache_create(arena, bytes); frl_pack_cache_activate(h) ... calls ... frl_pack_cache_activate(0);
// frl_pack_cache_refresh(h, stream) after the weights changed; frl_pack_cache_destroy(h).  One cache per trainer; the active cache is a
// process-wide setting (the autograd engine's worker threads run the backward calls).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

enum { FRL_PACK_PW = 0, FRL_PACK_C3 = 1, FRL_PACK_PW_REP = 2, FRL_PACK_PW_K4 = 3 };

// One contiguous run of fragments of an image, produced from one float32 weight tensor.
//   FRL_PACK_PW: pack_weights_lds<T, NF>(dst, W, Cout, Cin, MB, so, si)        (frl_common.hpp)
//   FRL_PACK_C3: c3_pack_kernel<T, NF>(dst, W, so, si, tap_rev, Cin, Cout)      (conv3x3.hip)
//   FRL_PACK_PW_REP: as FRL_PACK_PW with the rows of a block replicated over the lane quarters, oc = 4 * mb + (r & 3)
//   FRL_PACK_PW_K4: as FRL_PACK_PW (NF = 1) with k-element e of lane quarter kc <-> input channel 4 kc + e for e < 4, zero beyond (film_fused.hip)
struct FrlPackJob {
  const float* W;
  char* dst;             // absolute address once registered; offset inside the image when handed to frl_pack_cached
  long long so, si;
  int kind, dtype, NF;
  int Cout, Cin, MB;
  int tap_rev, nfrag;    // fragments in this run (16 bytes each for bf16, 4 bytes for f32)
};

static inline FrlPackJob frl_pack_job_pw(const float* W, size_t dst_off, int dtype, int NF, int Cout, int Cin, int MB, long long so, long long si) {
  FrlPackJob j;
  j.W = W; j.dst = (char*)dst_off; j.so = so; j.si = si; j.kind = FRL_PACK_PW; j.dtype = dtype; j.NF = NF;
  j.Cout = Cout; j.Cin = Cin; j.MB = MB; j.tap_rev = 0; j.nfrag = MB * NF * 64;
  return j;
}
static inline FrlPackJob frl_pack_job_c3(const float* W, size_t dst_off, int dtype, int NF, int Cout, int Cin, long long so, long long si, int tap_rev,
                                         int nfrag) {
  FrlPackJob j;
  j.W = W; j.dst = (char*)dst_off; j.so = so; j.si = si; j.kind = FRL_PACK_C3; j.dtype = dtype; j.NF = NF;
  j.Cout = Cout; j.Cin = Cin; j.MB = (Cout + 15) >> 4; j.tap_rev = tap_rev; j.nfrag = nfrag;
  return j;
}

// Image of `bytes` bytes described by `jobs` (dst = offset inside the image).  Returns NULL when no cache is active (or its arena is
// full): the caller packs into its workspace as always.  Otherwise the image's address inside the arena; *hit tells whether it is already
// packed (skip the prologue) or freshly registered (the caller packs it once, there; later refreshes keep it current).
void* frl_pack_cached(const FrlPackJob* jobs, int njobs, size_t bytes, bool* hit);
