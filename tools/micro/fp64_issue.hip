// Microbenchmark (GPU box): issue rate of dependent / independent fp64 VALU chains at 1, 2, 3, 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_issue.hip -o /tmp/fp64_issue && /tmp/fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ void __launch_bounds__(64) k(double *out, int n, double a, double b) {
  double x[CH];
  for (int c = 0; c < CH; ++c) x[c] = out[threadIdx.x + c];
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
        else if (OP == 1) x[c] = x[c]*a;
        else if (OP == 2) x[c] = x[c] + b;
        else if (OP == 3) x[c] = __builtin_amdgcn_rcp(x[c]);
        else if (OP == 4) x[c] = b/x[c];
        else if (OP == 5) x[c] = sqrt(x[c]);
        else if (OP == 6) x[c] = ldexp(x[c], 1) - x[c];
        else if (OP == 7) x[c] = (x[c] > a) ? b : x[c] + a;
      }
    }
  }
  double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
  if (s == 12345.678) out[0] = s;
}
template <int CH, int OP>
void run(const char *name, double *d, int waves_per_simd, int ops_per_elem) {
  const int n = 2000;
  int blocks = 256*4*waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH, OP><<<blocks, 64>>>(d, 10, 1.0000001, 1e-9);
  hipEventRecord(e0);
  k<CH, OP><<<blocks, 64>>>(d, n, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = (double)n*16*CH*waves_per_simd;     // source-level ops per SIMD
  printf("%-10s chains %d waves/SIMD %d: %.3f ms  -> %.2f ns per op per SIMD (%.1f cycles at 2.4 GHz)\n", name, CH, waves_per_simd,
         ms, ms*1e6/insts_per_simd, ms*1e6/insts_per_simd*2.4);
}
int main() {
  double *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  for (int w = 1; w <= 4; ++w) {
    run<1, 0>("fma", d, w, 1); run<2, 0>("fma", d, w, 1); run<4, 0>("fma", d, w, 1);
    run<1, 1>("mul", d, w, 1); run<4, 1>("mul", d, w, 1);
    run<1, 2>("add", d, w, 1); run<4, 2>("add", d, w, 1);
    run<1, 3>("rcp", d, w, 1); run<4, 3>("rcp", d, w, 1);
    run<1, 4>("div", d, w, 1); run<4, 4>("div", d, w, 1);
    run<1, 5>("sqrt", d, w, 1); run<4, 5>("sqrt", d, w, 1);
    run<1, 6>("ldexp-sub", d, w, 1); run<4, 6>("ldexp-sub", d, w, 1);
    run<1, 7>("cmp-sel-add", d, w, 1); run<4, 7>("cmp-sel-add", d, w, 1);
  }
  return 0;
}
