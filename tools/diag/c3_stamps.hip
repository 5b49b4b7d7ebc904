// Diagnostic harness: phase stamps of the 3x3 convolution forward kernel (bf16, B=256, 32x32).
#define C3_STAMPS 1
#include "../../vq-vae_amd/csrc/conv3x3.hip"
#include "../../vq-vae_amd/csrc/frl_host.hip"
#include <vector>
#include <random>
#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int B = 256, H = 32, W = 32, Cin = argc > 1 ? atoi(argv[1]) : 64, Cout = argc > 2 ? atoi(argv[2]) : 64;
  const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout;
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<bf16> hx(nx); for (auto& v : hx) v = (bf16)nd(rng);
  std::vector<float> hw((size_t)Cout * Cin * 9), hb(Cout, 0.1f); for (auto& v : hw) v = nd(rng) * 0.04f;
  bf16 *x, *y; float *w, *b; void* ws; unsigned long long* dbg;
  CK_(hipMalloc(&x, nx * 2)); CK_(hipMalloc(&y, ny * 2)); CK_(hipMalloc(&w, hw.size() * 4)); CK_(hipMalloc(&b, Cout * 4));
  CK_(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK_(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemcpy(b, hb.data(), Cout * 4, hipMemcpyHostToDevice));
  const size_t wsb = (size_t)9 * 256 * 256 * 4 + 4096; CK_(hipMalloc(&ws, wsb));
  CK_(hipMalloc(&dbg, 256 * 64 * 8)); CK_(hipMemset(dbg, 0, 256 * 64 * 8));
  CK_(hipMemcpyToSymbol(HIP_SYMBOL(c3_dbg), &dbg, sizeof(dbg)));
  hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
  for (int it = 0; it < 4; ++it) {
    CK_(hipEventRecord(e0, 0));
    int rc = frl_conv3x3_fwd(x, w, b, y, B, H, W, Cin, Cout, 1, FRL_BF16, ws, wsb, 0);
    CK_(hipEventRecord(e1, 0)); CK_(hipEventSynchronize(e1));
    float ms; CK_(hipEventElapsedTime(&ms, e0, e1)); printf("iter %d rc=%d %.1f us (pack + conv3x3 %d->%d)\n", it, rc, ms * 1e3f, Cin, Cout);
  }
  std::vector<unsigned long long> h(256 * 64); CK_(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  const char* nm[5] = {"epilogue/loop of prev stage", "wait barrier 1", "halo commit + weight copy", "wait barrier 2", "prefetch issue + 9 taps"};
  double tot = 0;
  for (int ph = 0; ph < 5; ++ph) { double s = 0; for (int bb = 0; bb < 256; ++bb) for (int wv = 0; wv < 8; ++wv) s += (double)h[(size_t)bb * 64 + wv * 8 + ph]; s /= 256.0 * 8; tot += s; printf("phase %d %-30s %9.0f cycles per wave (whole kernel)\n", ph, nm[ph], s); }
  printf("total %.0f cycles per wave\n", tot);
  return 0;
}
