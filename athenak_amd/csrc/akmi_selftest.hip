// akmi_selftest.hip -- bit-equality self-test of the short fp64 forms in akmi_numerics.hpp.
//
// The stage kernels replace three compiler expansions by shorter instruction sequences that must return
// the SAME BITS (the parity bar of the path is bit equality with the CPU build of the reference):
//   mode 0   sqrt_x(x)            against  sqrt(x)        (18 -> 10 VALU instructions + 2 for the guard)
//   mode 1   rcp_x(x)             against  1.0/x          (11 ->  7 + 2)
//   mode 2   ldexp(x, shift(dx))  against  x/dx           for dx a power of two (11 -> 1)
// akmi_selftest_fp64 evaluates both sides for `n` operands per mode on the device and returns the
// number of operands whose results differ in any bit (NaN results count as equal when both are NaN).
// Operands: three waves out of four hold only operands inside the window of the short form (random
// mantissa, exponent uniform over the window) -- these waves run the short form; the fourth holds
// arbitrary 64-bit patterns (subnormals, infinities, NaNs, negative numbers, zeros) and window
// boundaries +/- 1 ulp, which must divert the whole wave to the compiler's expansion.  The first 4096
// operands of every mode are a fixed table of edge values.
#include <hip/hip_runtime.h>
#include "akmi_common.hpp"

namespace akmi {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {     // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30))*0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27))*0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }

// edge table: exponents around every boundary the short forms and the compiler's scalings know about,
// crossed with mantissas 0, 1, all ones, all ones - 1, 1000..0, and both signs
__device__ double edge_operand(unsigned idx) {
  const int exps[] = {0, 1, 2, 52, 53, 54, 255, 256, 257, 322, 323, 324, 511, 767, 768, 1021, 1022, 1023, 1024, 1025,
                      1278, 1279, 1535, 1722, 1723, 1724, 1790, 1791, 2044, 2045, 2046, 2047};
  const unsigned long long mans[] = {0ull, 1ull, 0xfffffffffffffull, 0xffffffffffffeull, 0x8000000000000ull,
                                     0x5555555555555ull, 0xaaaaaaaaaaaaaull, 0x0000000100000ull};
  const unsigned ne = sizeof(exps)/sizeof(exps[0]), nm = sizeof(mans)/sizeof(mans[0]);
  const unsigned e = idx % ne, m = (idx/ne) % nm, sg = (idx/(ne*nm)) & 1u;
  return from_bits(((unsigned long long)sg << 63) | ((unsigned long long)exps[e] << 52) | mans[m]);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_selftest_fp64(long long n, unsigned long long seed,
                                                        unsigned long long *mismatch, unsigned long long *shortform) {
  const long long stride = (long long)gridDim.x*blockDim.x;
  unsigned long long bad = 0, fast = 0;
  for (long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x; t < n + stride; t += stride) {
    // whole waves iterate together (t < n + stride keeps the last partial wave converged): the guards vote
    const bool live = t < n;
    const unsigned long long wave = (unsigned long long)(t >> 6);
    const unsigned long long r = mix64(seed ^ ((unsigned long long)t*0x2545f4914f6cdd1dull));
    const unsigned long long hw = mix64(seed + 0x1234567ull + wave);
    double x;
    if (t < 4096) {
      x = edge_operand((unsigned)t);
    } else if ((hw & 3ull) != 0) {        // in-window wave: exponent in [323, 1723), any mantissa, positive
      const unsigned long long e = 323ull + (r >> 53) % 1400ull;
      x = from_bits((e << 52) | (r & 0xfffffffffffffull));
      if (MODE != 0 && (r & (1ull << 52))) x = -x;
    } else {                              // anything
      x = from_bits(r);
      if ((r & 0xf00ull) == 0) x = from_bits(((unsigned long long)((r >> 20) & 1 ? 323 : 1723) << 52) - ((r >> 12) & 3));
    }
    double a, b;
    if constexpr (MODE == 0) {
      a = sqrt_x(x);
      b = sqrt(x);
      fast += !__any(!((hi_word(x) - (323u << 20)) < (1400u << 20)));
    } else if constexpr (MODE == 1) {
      a = rcp_x(x);
      b = 1.0/x;
      fast += !__any(!in_core_range(x));
    } else {
      // wave-uniform power of two between 2^-80 and 2^80, as a block's cell size is
      const int k = (int)((hw >> 8) % 161ull) - 80;
      const double dx = from_bits((unsigned long long)(1023 + k) << 52);
      a = ldexp(x, 1023 - (int)(((unsigned long long)__double_as_longlong(dx) >> 52) & 0x7ff));
      b = x/dx;
      fast += 1;
    }
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const bool same = (ba == bb) || (a != a && b != b);
    if (live && !same) ++bad;
  }
  for (int o = 32; o > 0; o >>= 1) {
    bad += __shfl_xor(bad, o, 64);
    fast += __shfl_xor(fast, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (bad) atomicAdd(mismatch, bad);
    atomicAdd(shortform, fast >> 6);      // waves x iterations that ran the short form (every lane counted it)
  }
}

}  // namespace akmi

extern "C" {
int akmi_selftest_fp64(int mode, long long n, unsigned long long seed, long long *mismatch, long long *shortform_waves,
                       void *stream);
}

// mode 0/1/2 as above.  mismatch / shortform_waves: host pointers (shortform_waves may be null): the number of
// operands whose two results differ, and the number of wave-iterations in which the short form (not the
// fallback) ran -- the test asserts that this is the large majority, i.e. that the equality is not vacuous.
extern "C" int akmi_selftest_fp64(int mode, long long n, unsigned long long seed, long long *mismatch,
                                  long long *shortform_waves, void *stream) {
  using namespace akmi;
  if (mode < 0 || mode > 2 || n <= 0 || !mismatch) { set_error("akmi_selftest_fp64: bad arguments"); return AKMI_FAIL; }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long *d = nullptr;
  if (hipMalloc(&d, 2*sizeof(unsigned long long)) != hipSuccess) { set_error("akmi_selftest_fp64: hipMalloc"); return AKMI_FAIL; }
  hipMemsetAsync(d, 0, 2*sizeof(unsigned long long), st);
  const int blocks = 256*8;
  if (mode == 0) k_selftest_fp64<0><<<blocks, 256, 0, st>>>(n, seed, d, d + 1);
  else if (mode == 1) k_selftest_fp64<1><<<blocks, 256, 0, st>>>(n, seed, d, d + 1);
  else k_selftest_fp64<2><<<blocks, 256, 0, st>>>(n, seed, d, d + 1);
  hipError_t e = hipGetLastError();
  unsigned long long h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d);
  if (e != hipSuccess) { set_error("akmi_selftest_fp64: %s", hipGetErrorString(e)); return AKMI_FAIL; }
  *mismatch = (long long)h[0];
  if (shortform_waves) *shortform_waves = (long long)h[1];
  return AKMI_COMPLETE;
}
