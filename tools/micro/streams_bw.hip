// Microbenchmark (GPU box): does the NUMBER of concurrent streams of a kernel matter on MI355X HBM3E?  A thread reads S
// arrays at one index and writes W arrays (the shape of k_corner_ct: 18 in, 3 out; of the x3 march: 24 in, 8 out), against
// the same bytes as ONE interleaved array read with 8- or 16-byte loads (array of structures), 32 MiB per stream.
// hipcc --offload-arch=gfx950 -O3 streams_bw.hip -o streams_bw && ./streams_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Ptrs { const double *in[24]; double *out[8]; };
template <int S, int W>
__global__ void __launch_bounds__(256) k_soa(Ptrs p, long n) {
  const long i = (long)blockIdx.x*256 + threadIdx.x;
  if (i >= n) return;
  double v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) v[s] = p.in[s][i];
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) acc += v[s];
#pragma unroll
  for (int w = 0; w < W; ++w) p.out[w][i] = acc + w;
}
// the same bytes, inputs interleaved per cell: in[i*S + s], outputs interleaved: out[i*W + w]
template <int S, int W>
__global__ void __launch_bounds__(256) k_aos(const double *__restrict__ in, double *__restrict__ out, long n) {
  const long i = (long)blockIdx.x*256 + threadIdx.x;
  if (i >= n) return;
  const double2 *q = reinterpret_cast<const double2 *>(in + i*S);
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < S/2; ++s) { const double2 t = q[s]; acc += t.x + t.y; }
#pragma unroll
  for (int w = 0; w < W; ++w) out[i*W + w] = acc + w;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms/5;
}
template <int S, int W> void run(long n, double *pool) {
  Ptrs p;
  for (int s = 0; s < S; ++s) p.in[s] = pool + (long)s*n;
  for (int w = 0; w < W; ++w) p.out[w] = pool + (long)(24 + w)*n;
  const int grid = (int)((n + 255)/256);
  const float a = timeit([&] { k_soa<S, W><<<grid, 256>>>(p, n); });
  const float b = timeit([&] { k_aos<S, W><<<grid, 256>>>(pool, pool + 24*n, n); });
  const double gb = (double)(S + W)*n*8/1e9;
  printf("%2d in + %d out  separate arrays %.3f ms %.2f TB/s | one interleaved array each way %.3f ms %.2f TB/s\n", S, W, a, gb/a, b, gb/b);
}
int main() {
  const long n = 16l*1024*1024;          // 128 MiB per stream
  double *pool; hipMalloc(&pool, 32*n*8); hipMemset(pool, 0, 32*n*8);
  run<2, 2>(n, pool); run<4, 2>(n, pool); run<8, 2>(n, pool); run<12, 4>(n, pool); run<18, 3>(n, pool); run<24, 8>(n, pool);
  return 0;
}
