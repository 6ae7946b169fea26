// Host stand-in for <hip/hip_runtime.h>: lets g++ compile athenak_amd/csrc/akmi_numerics.hpp (pure per-cell / per-face
// arithmetic) for the CPU, so that tests/test_numerics_host.py can compare it with the oracle bit for bit without a GPU.
// Test infrastructure only; nothing of the product includes it.
#pragma once
#include <math.h>
#include <string.h>
#include <cmath>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif
static inline long long __double_as_longlong(double x) { long long r; memcpy(&r, &x, 8); return r; }
static inline double __longlong_as_double(long long x) { double r; memcpy(&r, &x, 8); return r; }
static inline bool __any(bool x) { return x; }                  // a "wave" of one lane
#define __builtin_amdgcn_rsq(x) (1.0/std::sqrt(x))              // the short forms are exercised on the GPU (akmi_selftest_fp64)
#define __builtin_amdgcn_rcp(x) (1.0/(x))
#define __log2f(x) log2f(x)                                       // (glibc declares a private symbol of that name)
