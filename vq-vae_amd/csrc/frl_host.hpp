// Host-side helpers shared by the C-ABI entry points (error string, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

int frl_fail(int code, const char* msg);           // records thread-local message, returns code
int frl_check_launch(const char* what);            // hipGetLastError() -> 0 or negative code

#define FRL_HIP(expr)                                            \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) return frl_fail(-100 - (int)_e, hipGetErrorString(_e)); \
  } while (0)

// internal cross-file dispatchers
int frl_pw_dispatch(const void* x, const void* xmask, int mask_act, const float* w, int64_t so, int64_t si,
                    const float* bias, void* y, int64_t P, int Cin, int Cout, int act, int dtype, hipStream_t st);
