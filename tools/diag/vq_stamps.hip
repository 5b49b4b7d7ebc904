// Diagnostic harness: phase stamps of the VQ assignment kernel (16-wave variant) on BASELINE configs[1] sizes.
#define VQ_STAMPS 1
#include "../../vq-vae_amd/csrc/vq.hip"
#include "../../vq-vae_amd/csrc/frl_host.hip"
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int64_t N = 262144; const int K = argc > 1 ? atoi(argv[1]) : 512, d = 64;
  const int G = argc > 2 ? atoi(argv[2]) : 512;          // workgroups of the kernel variant (512: resident, 256: streaming)
  if (argc > 3) frl_vq_stream_tiles(atoi(argv[3]));
  std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<bf16> hz(N * d); for (auto& v : hz) v = (bf16)nd(rng);
  std::vector<float> he((size_t)K * d); for (auto& v : he) v = nd(rng);
  bf16 *z, *zq; float *E, *stats; int32_t *idx, *counts; void* ws; unsigned long long* dbg;
  CK(hipMalloc(&z, N * d * 2)); CK(hipMalloc(&zq, N * d * 2)); CK(hipMalloc(&E, K * d * 4)); CK(hipMalloc(&stats, 64)); CK(hipMalloc(&idx, N * 4));
  CK(hipMalloc(&counts, K * 4));
  CK(hipMemcpy(z, hz.data(), N * d * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(E, he.data(), (size_t)K * d * 4, hipMemcpyHostToDevice));
  const size_t wsb = frl_vq_workspace_bytes(N, K, d); CK(hipMalloc(&ws, wsb));
  CK(hipMalloc(&dbg, 512 * 130 * 8)); CK(hipMemset(dbg, 0, 512 * 130 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(vq_dbg), &dbg, sizeof(dbg)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  void* prep; const size_t pb = frl_vq_prepared_bytes(K, d); CK(hipMalloc(&prep, pb));
  { int rc = frl_vq_prepare(E, N, K, d, FRL_BF16, prep, pb, 0); printf("prepare rc=%d\n", rc); }
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0, 0));
    int rc = frl_vq_assign_fwd_prepared(z, E, prep, N, K, d, idx, zq, stats, counts, FRL_BF16, ws, wsb, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("iter %d rc=%d %.1f us (whole frl_vq_assign_fwd_prepared)\n", it, rc, ms * 1e3f);
  }
  { float st[4]; CK(hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost)); printf("stats: sqerr %.6e perplexity %.4f re-evaluated %.0f\n", st[0], st[1], st[2]); }
  std::vector<unsigned long long> h(512 * 130); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  const char* nm[7] = {"wait for rows / z load + norms", "store drain (streaming)", "image / norms fill (once)", "main loop", "epilogue", "park + exact re-evaluation", "partials, atomics, ticket, statistics"};
  double tot = 0;
  for (int ph = 0; ph < 7; ++ph) { double s = 0; int nz = 0; for (int b = 0; b < G; ++b) for (int w = 0; w < 16; ++w) { const double v = (double)h[(size_t)b * 128 + w * 8 + ph]; s += v; nz += h[(size_t)b * 128 + w * 8 + 3] != 0; } s /= nz; tot += s; printf("phase %d %-28s %9.0f cycles per wave (whole kernel)\n", ph, nm[ph], s); }
  printf("total %.0f cycles per wave\n", tot);
  {  // workgroup lifetimes in s_memtime ticks and in 100 MHz wall-clock ticks
    double sm = 0, sw = 0; std::vector<double> life;
    for (int b = 0; b < G; ++b) { sm += (double)h[G * 128 + 2 * b]; sw += (double)h[G * 128 + 2 * b + 1]; life.push_back((double)h[G * 128 + 2 * b + 1] * 0.01); }
    std::sort(life.begin(), life.end());
    printf("workgroup lifetime: mean %.0f s_memtime ticks = %.2f us wall (=> %.1f MHz tick rate); p10 %.2f p50 %.2f p90 %.2f max %.2f us\n", sm / G, sw / G * 0.01, sm / sw * 100.0,
           life[G / 10], life[G / 2], life[G * 9 / 10], life[G - 1]);
    for (int ph = 0; ph < 7; ++ph) { double mx = 0; for (int b = 0; b < G; ++b) for (int w = 0; w < 16; ++w) mx = std::max(mx, (double)h[(size_t)b * 128 + w * 8 + ph]); printf("phase %d max over waves %9.0f\n", ph, mx); }
  }
  return 0;
}
