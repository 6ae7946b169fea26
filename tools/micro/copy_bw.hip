// Microbenchmark (GPU box): what a streaming fp64 kernel can reach on MI355X HBM3E -- 8- and 16-byte accesses per lane, plain vs
// non-temporal loads / stores, read-only and write-only streams, grid sizes.  512 MiB per array (beyond the 256 MB MALL).
// hipcc --offload-arch=gfx950 -O3 copy_bw.hip -o copy_bw && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0 plain 8B, 1 nt store, 2 nt load+store, 3 16B plain, 4 16B nt both
__global__ void __launch_bounds__(256) k_copy(const double *__restrict__ a, double *__restrict__ b, long n) {
  long i = (long)blockIdx.x*256 + threadIdx.x;
  const long st = (long)gridDim.x*256;
  if (MODE <= 2) {
    for (; i < n; i += st) {
      double v = (MODE == 2) ? __builtin_nontemporal_load(a + i) : a[i];
      if (MODE >= 1) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
  } else {
    const double2 *a2 = (const double2 *)a; double2 *b2 = (double2 *)b;
    for (; i < n/2; i += st) {
      double2 v;
      if (MODE == 4) { v.x = __builtin_nontemporal_load(&a2[i].x); v.y = __builtin_nontemporal_load(&a2[i].y); } else v = a2[i];
      if (MODE == 4) { __builtin_nontemporal_store(v.x, &b2[i].x); __builtin_nontemporal_store(v.y, &b2[i].y); } else b2[i] = v;
    }
  }
}
__global__ void __launch_bounds__(256) k_read(const double *__restrict__ a, double *__restrict__ out, long n) {
  long i = (long)blockIdx.x*256 + threadIdx.x; const long st = (long)gridDim.x*256; double s = 0;
  for (; i < n; i += st) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(double *__restrict__ b, long n, double v) {
  long i = (long)blockIdx.x*256 + threadIdx.x; const long st = (long)gridDim.x*256;
  for (; i < n; i += st) b[i] = v;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms/5;
}
int main() {
  const long n = 64l*1024*1024;
  double *a, *b; hipMalloc(&a, n*8); hipMalloc(&b, n*8); hipMemset(a, 0, n*8); hipMemset(b, 0, n*8);
  const char *names[5] = {"8B plain", "8B nt-store", "8B nt-load+store", "16B plain", "16B nt both"};
  for (int grid : {256*4, 256*8, 256*16, 256*64, (int)(n/256)}) {
    float ms[5];
    ms[0] = timeit([&] { k_copy<0><<<grid, 256>>>(a, b, n); });
    ms[1] = timeit([&] { k_copy<1><<<grid, 256>>>(a, b, n); });
    ms[2] = timeit([&] { k_copy<2><<<grid, 256>>>(a, b, n); });
    ms[3] = timeit([&] { k_copy<3><<<grid > (int)(n/512) ? (int)(n/512) : grid, 256>>>(a, b, n); });
    ms[4] = timeit([&] { k_copy<4><<<grid > (int)(n/512) ? (int)(n/512) : grid, 256>>>(a, b, n); });
    for (int m = 0; m < 5; ++m) printf("copy  grid %8d  %-18s %.3f ms  %.2f TB/s (read + write)\n", grid, names[m], ms[m], 2.0*n*8/ms[m]/1e9);
    float r = timeit([&] { k_read<<<grid, 256>>>(a, b, n); }), w = timeit([&] { k_write<<<grid, 256>>>(b, n, 1.0); });
    printf("read  grid %8d  %.3f ms  %.2f TB/s | write %.3f ms  %.2f TB/s\n", grid, r, n*8/r/1e9, w, n*8/w/1e9);
  }
  return 0;
}
