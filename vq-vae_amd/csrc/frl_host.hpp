// Host-side helpers shared by the C-ABI entry points (error string, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

int frl_fail(int code, const char* msg);           // records thread-local message, returns code
int frl_check_launch(const char* what);            // hipGetLastError() -> 0 or negative code

// Kernel launch with per-launch error capture.  hipGetLastError() is sticky per thread and may hold a stale error from
// runtime initialisation done elsewhere in the process (e.g. by PyTorch), so the state is cleared before every launch
// and the first failure of a call is parked in g_frl_pending until frl_check_launch() reports it.
extern thread_local hipError_t g_frl_pending;
// Optional per-kernel timing (frl_kernel_timing_enable): a HIP event pair recorded on the launch stream immediately around every
// kernel the library launches, keyed by the kernel expression -- the live counterpart of a rocprofv3 kernel trace (bench.py).
extern int g_frl_timing;
void frl_timing_begin(const char* kernel, hipStream_t st);
void frl_timing_end(hipStream_t st);
#define FRL_ARG1_(a, ...) a
#define FRL_ARG5_(a, b, c, d, e, ...) e
#define FRL_STR2_(...) #__VA_ARGS__
#define FRL_STR_(...) FRL_STR2_(__VA_ARGS__)
#define FRL_LAUNCH(...)                                                     \
  do {                                                                      \
    (void)hipGetLastError();                                                \
    if (g_frl_timing) frl_timing_begin(FRL_STR_(FRL_ARG1_(__VA_ARGS__)), FRL_ARG5_(__VA_ARGS__)); \
    hipLaunchKernelGGL(__VA_ARGS__);                                        \
    if (g_frl_timing) frl_timing_end(FRL_ARG5_(__VA_ARGS__));               \
    hipError_t _le = hipGetLastError();                                     \
    if (_le != hipSuccess && g_frl_pending == hipSuccess) g_frl_pending = _le; \
  } while (0)

// same with an explicit timing name (for launches through a function-pointer variable)
#define FRL_LAUNCH_AS(name, ...)                                            \
  do {                                                                      \
    (void)hipGetLastError();                                                \
    if (g_frl_timing) frl_timing_begin(name, FRL_ARG5_(__VA_ARGS__));       \
    hipLaunchKernelGGL(__VA_ARGS__);                                        \
    if (g_frl_timing) frl_timing_end(FRL_ARG5_(__VA_ARGS__));               \
    hipError_t _le = hipGetLastError();                                     \
    if (_le != hipSuccess && g_frl_pending == hipSuccess) g_frl_pending = _le; \
  } while (0)

#define FRL_HIP(expr)                                            \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) return frl_fail(-100 - (int)_e, hipGetErrorString(_e)); \
  } while (0)

// internal cross-file dispatchers
int frl_pw_dispatch(const void* x, const void* xmask, int mask_act, const float* w, int64_t so, int64_t si,
                    const float* bias, void* y, int64_t P, int Cin, int Cout, int act, int dtype, void* ws, size_t ws_bytes,
                    hipStream_t st, const void* yadd = nullptr);
