// Diagnostic harness (not part of the product library): runs the hot TCN backward kernel with s_memtime stamps at its phase
// boundaries and prints the median cycles per phase.  Build: tools/diag/build.sh ; run on the GPU box: tools/diag/tcn_bwd_stamps
// build variants (tools/diag/build.sh): default (timing only), -DB3_STAMPS (phase stamps of tcn_hot_bwd3_kernel)
#include "../../vq-vae_amd/csrc/tcn_hot.hip"
#include "../../vq-vae_amd/csrc/tcn_hot_bwd3.hip"
#include "../../vq-vae_amd/csrc/tcn_hot_bwd4.hip"
#include "../../vq-vae_amd/csrc/frl_host.hip"
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = 256, T = 5, HW = 1024, C = 64;
  const int dil = argc > 1 ? atoi(argv[1]) : 1;
  if (argc > 2) frl_tcn_hot_bwd_variant(atoi(argv[2]));            // 3: tcn_hot_bwd3_kernel, 4 (default): tcn_hot_bwd4_kernel
  const int64_t npix = (int64_t)B * HW, n = npix * T * C;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<bf16> hx(n), hdy(n);
  for (int64_t i = 0; i < n; ++i) { hx[i] = (bf16)nd(rng); hdy[i] = (bf16)nd(rng); }
  std::vector<float> wc(64 * 64 * 3), wg(64 * 64), v64(64);
  for (auto& w : wc) w = nd(rng) * 0.07f;
  for (auto& w : wg) w = nd(rng) * 0.12f;
  bf16 *x, *dy, *dx;
  float *dwc, *dwg, *b0, *gam, *bet, *b1, *g[6];
  void* ws;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&dx, n * 2));
  CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dwc, wc.size() * 4)); CK(hipMalloc(&dwg, wg.size() * 4));
  CK(hipMemcpy(dwc, wc.data(), wc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwg, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : v64) v = 0.1f * nd(rng);
  CK(hipMalloc(&b0, 256)); CK(hipMemcpy(b0, v64.data(), 256, hipMemcpyHostToDevice));
  CK(hipMalloc(&b1, 256)); CK(hipMemcpy(b1, v64.data(), 256, hipMemcpyHostToDevice));
  CK(hipMalloc(&bet, 256)); CK(hipMemcpy(bet, v64.data(), 256, hipMemcpyHostToDevice));
  for (auto& v : v64) v = 1.f;
  CK(hipMalloc(&gam, 256)); CK(hipMemcpy(gam, v64.data(), 256, hipMemcpyHostToDevice));
  const size_t gsz[6] = {64 * 64 * 3 * 4, 256, 256, 256, 64 * 64 * 4, 256};
  for (int i = 0; i < 6; ++i) CK(hipMalloc(&g[i], gsz[i]));
  const size_t wsb = frl_tcn_hot_bwd_workspace_bytes(npix);
  CK(hipMalloc(&ws, wsb));
  unsigned long long* dbg;
  const size_t ndbg = (size_t)256 * 4 * 16 * 8;
  CK(hipMalloc(&dbg, ndbg * 8)); CK(hipMemset(dbg, 0, ndbg * 8));
#ifdef B3_STAMPS
  CK(hipMemcpyToSymbol(HIP_SYMBOL(b3_dbg), &dbg, sizeof(dbg)));
#endif
#ifdef B4_STAMPS
  CK(hipMemcpyToSymbol(HIP_SYMBOL(b4_dbg), &dbg, sizeof(dbg)));
  {
    int knob[2] = {argc > 3 ? atoi(argv[3]) : 0, argc > 4 ? atoi(argv[4]) : 16};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(b4_knob), knob, sizeof(knob)));
    printf("knobs: priority mode %d, subgroup 0 share %d/32\n", knob[0], knob[1]);
  }
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, 0));
    int rc = frl_tcn_hot_bwd(x, nullptr, dy, dwc, b0, gam, bet, dwg, b1, dx, g[0], g[1], g[2], g[3], g[4], g[5], npix, HW, dil, 1e-5f, ws, wsb, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("iter %d rc=%d  %.1f us (pack + bwd + slab reduce)\n", it, rc, ms * 1e3f);
  }
#if !defined(B3_STAMPS) && !defined(B4_STAMPS)
  return 0;
#endif
  {
    std::vector<unsigned long long> h2((size_t)256 * 96 + 2048 + 512);
    CK(hipMemcpy(h2.data(), dbg, h2.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[12] = {"wait E' (x DMA landed)", "conv, stats, S1 publish n", "wait A", "S2 gate/sigmoid/dgpre", "wait B", "P2 gate wgrad",
                          "S3 gateT/GN bwd", "wait C", "DMA issue, dy loads, P3 publish", "wait D", "dx convT + store", "P4 conv wgrad (+prologue)"};
    double tot = 0;
    for (int ph = 0; ph < 12; ++ph) {
      double sum = 0;
      for (size_t b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) sum += (double)h2[b * 96 + w * 12 + ph];
      sum /= (256.0 * 8 * 16);
      tot += sum;
      printf("phase %2d %-36s %8.0f cycles / tile (mean over waves)\n", ph, nm[ph], sum);
    }
    printf("total %.0f cycles / tile\n", tot);
    {
      double mx = 0, mn = 1e30, av = 0;
      for (size_t b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) {
        double t = 0;
        for (int ph = 0; ph < 12; ++ph) t += (double)h2[b * 96 + w * 12 + ph];
        mx = std::max(mx, t); mn = std::min(mn, t); av += t / 2048.0;
      }
      printf("per wave, whole tile loop: min %.0f  mean %.0f  max %.0f cycles\n", mn, av, mx);
      double cy = 0, rt = 0;
      for (int b = 0; b < 256; ++b) { cy += (double)h2[(size_t)256 * 96 + 2048 + 2 * b]; rt += (double)h2[(size_t)256 * 96 + 2048 + 2 * b + 1]; }
      printf("workgroup lifetime: %.0f shader cycles = %.1f us of the 100 MHz clock -> %.3f GHz\n", cy / 256, rt / 256 / 100.0, cy / rt / 10.0);
    }
#ifdef B4_STAMPS
    for (int b = 0; b < 3; ++b) {
      printf("workgroup %d: SIMD of waves 0..7:", b);
      for (int w = 0; w < 8; ++w) printf(" %llu", h2[(size_t)256 * 96 + b * 8 + w]);
      printf("\n");
    }
#endif
  }
  return 0;
}
