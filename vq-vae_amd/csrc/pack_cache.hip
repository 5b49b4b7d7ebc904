// Weight-image cache (frl_pack.hpp): registry of packed MFMA fragment images + ONE kernel that rewrites all of them from the float32
// master weights.  The generic job kernel reproduces pack_weights_lds (frl_common.hpp) and c3_pack_kernel (conv3x3.hip) fragment for
// fragment with run-time (dtype, NF) -- tests compare cached and uncached runs bit for bit.
#include "frl_common.hpp"
#include "frl_host.hpp"
#include "frl_pack.hpp"
#include <map>
#include <mutex>
#include <string.h>
#include <vector>

#define PK_FRAGS_PER_BLOCK 512

// fragment i of a FRL_PACK_PW job: frag index (mb * NF + s) * 64 + lane;  row r of block mb <-> oc = qo*(r>>2) + 4*mb + (r&3);
// element e of frag s <-> ic = q*kc + s*FE + e  (frl_common.hpp: pack_weights_lds)
__device__ __forceinline__ void pk_frag_pw(const FrlPackJob& j, int i) {
  const int lane = i & 63, fs = i >> 6;
  const int s = fs % j.NF, mb = fs / j.NF;
  const int r = lane & 15, kc = lane >> 4;
  const int qo = 4 * j.MB;
  const int oc = j.kind == FRL_PACK_PW_REP ? 4 * mb + (r & 3) : qo * (r >> 2) + 4 * mb + (r & 3);
  if (j.dtype == FRL_BF16) {
    const int q = j.NF * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ic = j.kind == FRL_PACK_PW_K4 ? (e < 4 ? 4 * kc + e : j.Cin) : q * kc + 8 * s + e;
      v[e] = (oc < j.Cout && ic < j.Cin) ? (bf16)j.W[oc * j.so + ic * j.si] : (bf16)0.f;
    }
    reinterpret_cast<bf16x8*>(j.dst)[i] = v;
  } else {
    const int q = j.NF;
    const int ic = q * kc + s;
    reinterpret_cast<float*>(j.dst)[i] = (oc < j.Cout && ic < j.Cin) ? j.W[oc * j.so + ic * j.si] : 0.f;
  }
}

// fragment g of a FRL_PACK_C3 job (conv3x3.hip: c3_pack_kernel): block (oc0/4, ck/CK) holds [9 taps][4 m][NF][64] fragments
__device__ __forceinline__ void pk_frag_c3(const FrlPackJob& j, int g) {
  const int FE = j.dtype == FRL_BF16 ? 8 : 1;
  const int q = j.NF * FE, CK = 4 * q;
  const int MB = (j.Cout + 15) >> 4, qo = 4 * MB;
  const int nck = (j.Cin + CK - 1) / CK;
  const int per_block = 9 * 4 * j.NF * 64;
  const int blk = g / per_block, i = g % per_block;
  const int oc0 = (blk / nck) * 4, ck = (blk % nck) * CK;
  const int ln = i & 63, fs = i >> 6;
  const int s = fs % j.NF, m = (fs / j.NF) & 3, tap = fs / (j.NF * 4);
  const int r = ln & 15, kq = ln >> 4;
  const int oc = qo * (r >> 2) + 4 * (oc0 + m) + (r & 3);
  const int tsrc = j.tap_rev ? 8 - tap : tap;
  const bool mok = (oc0 + m) < MB;
  if (j.dtype == FRL_BF16) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ic = ck + q * kq + 8 * s + e;
      v[e] = (mok && oc < j.Cout && ic < j.Cin) ? (bf16)j.W[oc * j.so + ic * j.si + tsrc] : (bf16)0.f;
    }
    reinterpret_cast<bf16x8*>(j.dst)[g] = v;
  } else {
    const int ic = ck + q * kq + s;
    reinterpret_cast<float*>(j.dst)[g] = (mok && oc < j.Cout && ic < j.Cin) ? j.W[oc * j.so + ic * j.si + tsrc] : 0.f;
  }
}

// blocks[b] = {job index, first fragment of this block inside the job}
__global__ __launch_bounds__(256) void frl_pack_jobs_kernel(const FrlPackJob* __restrict__ jobs, const int2* __restrict__ blocks) {
  const int2 bk = blocks[blockIdx.x];
  const FrlPackJob j = jobs[bk.x];
  const int end = (bk.y + PK_FRAGS_PER_BLOCK) < j.nfrag ? (bk.y + PK_FRAGS_PER_BLOCK) : j.nfrag;
  for (int i = bk.y + threadIdx.x; i < end; i += 256) {
    if (j.kind == FRL_PACK_C3) pk_frag_c3(j, i); else pk_frag_pw(j, i);
  }
}

namespace {
struct Key {
  unsigned char bytes[8 * 72];       // the job list itself (W pointers, shapes, strides, offsets) is the key
  size_t n;
  bool operator<(const Key& o) const { return n != o.n ? n < o.n : memcmp(bytes, o.bytes, n) < 0; }
};
struct Cache {
  char* arena = nullptr;
  size_t bytes = 0, used = 0, table_bytes = 0;
  std::map<Key, char*> images;
  std::vector<FrlPackJob> jobs;      // absolute destinations
  std::vector<int2> blocks;
  bool dirty = false;
};
std::mutex g_mu;
std::map<int, Cache> g_caches;
int g_next = 1;
int g_active = 0;                    // process-wide: the backward calls run on autograd worker threads
constexpr size_t PK_MAX_JOBS = 2048, PK_MAX_BLOCKS = 16384;
}  // namespace

void* frl_pack_cached(const FrlPackJob* jobs, int njobs, size_t bytes, bool* hit) {
  *hit = false;
  if (g_active == 0 || njobs <= 0 || njobs > 8) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_caches.find(g_active);
  if (it == g_caches.end()) return nullptr;
  Cache& c = it->second;
  Key k;
  memset(&k, 0, sizeof(k));
  k.n = (size_t)njobs * sizeof(FrlPackJob);
  memcpy(k.bytes, jobs, k.n);
  auto im = c.images.find(k);
  if (im != c.images.end()) { *hit = true; return im->second; }
  const size_t need = (bytes + 255) / 256 * 256;
  int nblk = 0;
  for (int i = 0; i < njobs; ++i) nblk += (jobs[i].nfrag + PK_FRAGS_PER_BLOCK - 1) / PK_FRAGS_PER_BLOCK;
  if (c.used + need > c.bytes || c.jobs.size() + njobs > PK_MAX_JOBS || c.blocks.size() + nblk > PK_MAX_BLOCKS) return nullptr;   // full
  char* img = c.arena + c.used;
  c.used += need;
  for (int i = 0; i < njobs; ++i) {
    FrlPackJob j = jobs[i];
    j.dst = img + (size_t)j.dst;
    const int ji = (int)c.jobs.size();
    c.jobs.push_back(j);
    for (int f = 0; f < j.nfrag; f += PK_FRAGS_PER_BLOCK) c.blocks.push_back(int2{ji, f});
  }
  c.images[k] = img;
  c.dirty = true;
  return img;
}

extern "C" {

// bytes of the arena that are reserved for the job tables; an arena must be larger than this
size_t frl_pack_cache_table_bytes(void) { return PK_MAX_JOBS * sizeof(FrlPackJob) + PK_MAX_BLOCKS * sizeof(int2); }

// arena: caller-owned device memory that outlives the cache (256-byte aligned).  Returns a handle > 0, or a negative code.
int frl_pack_cache_create(void* arena, size_t bytes) {
  const size_t tb = (frl_pack_cache_table_bytes() + 255) / 256 * 256;
  if (arena == nullptr || bytes <= tb + 4096) return frl_fail(-4, "pack_cache_create: arena too small");
  std::lock_guard<std::mutex> lk(g_mu);
  Cache c;
  c.arena = (char*)arena;
  c.bytes = bytes;
  c.table_bytes = tb;
  c.used = tb;
  const int h = g_next++;
  g_caches[h] = c;
  return h;
}

int frl_pack_cache_destroy(int handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_active == handle) g_active = 0;
  g_caches.erase(handle);
  return 0;
}

// handle: the cache that conv-like calls consult from now on; 0: none (every call packs into its own workspace).  Returns the previous one.
int frl_pack_cache_activate(int handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int was = g_active;
  g_active = (handle != 0 && g_caches.count(handle)) ? handle : 0;
  return was;
}

// Number of images registered so far (test / diagnostics).
int frl_pack_cache_images(int handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_caches.find(handle);
  return it == g_caches.end() ? -1 : (int)it->second.images.size();
}

// Rewrites every registered image from its float32 master weights with ONE launch.  When images were registered since the last call
// the job tables are uploaded first (a host-to-device copy: not capturable -- a step that is to be captured in a graph must have run
// eagerly once, so that all its images exist).
int frl_pack_cache_refresh(int handle, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_caches.find(handle);
  if (it == g_caches.end()) return frl_fail(-2, "pack_cache_refresh: unknown handle");
  Cache& c = it->second;
  if (c.jobs.empty()) return 0;
  FrlPackJob* jd = (FrlPackJob*)c.arena;
  int2* bd = (int2*)(c.arena + PK_MAX_JOBS * sizeof(FrlPackJob));
  if (c.dirty) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return frl_fail(-5, "pack_cache_refresh: images were registered during a stream capture (run the step eagerly first)");
    FRL_HIP(hipMemcpyAsync(jd, c.jobs.data(), c.jobs.size() * sizeof(FrlPackJob), hipMemcpyHostToDevice, stream));
    FRL_HIP(hipMemcpyAsync(bd, c.blocks.data(), c.blocks.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
    FRL_HIP(hipStreamSynchronize(stream));       // the host vectors may grow again right after this call
    c.dirty = false;
  }
  FRL_LAUNCH(frl_pack_jobs_kernel, dim3((unsigned)c.blocks.size()), dim3(256), 0, stream, (const FrlPackJob*)jd, (const int2*)bd);
  return frl_check_launch("pack_cache_refresh");
}

}  // extern "C"
