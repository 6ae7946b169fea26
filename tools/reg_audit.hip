// tools/reg_audit.hip -- where the registers of a PLM + HLLD face go: the pieces of akmi_numerics.hpp compiled on their
// own (operands from memory, results to memory, nothing else alive), with the flags of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-schedule-relaxed-occupancy=true \
//         -I athenak_amd/csrc -c tools/reg_audit.hip --save-temps      (tools/reg_audit.sh; VGPR counts from the .s)
#include <hip/hip_runtime.h>
#include "akmi_numerics.hpp"
using namespace akmi;

#define LD(n) const double n = in[(q++)*N + t]
template <bool EO, bool FM, int WAVES>
__global__ void __launch_bounds__(256, WAVES) ra_hlld(const double *__restrict__ in, double *__restrict__ out, long N, double gamma) {
  const long t = (long)blockIdx.x*256 + threadIdx.x;
  int q = 0;
  LD(dl); LD(ul); LD(vl); LD(wl); LD(el); LD(byl); LD(bzl); LD(dr); LD(ur); LD(vr); LD(wr); LD(er); LD(byr); LD(bzr); LD(bn);
  const Cons1D f = hlld<EO, FM>(gamma, dl, ul, vl, wl, el, byl, bzl, dr, ur, vr, wr, er, byr, bzr, bn);
  out[t] = f.d; out[N + t] = f.mx; out[2*N + t] = f.my; out[3*N + t] = f.mz; out[4*N + t] = f.e; out[5*N + t] = f.by;
  out[6*N + t] = f.bz;
}
// speeds and selection only (step 1 + 2 of hlld()): what is alive before a side is chosen
__global__ void __launch_bounds__(256) ra_fast_speeds(const double *__restrict__ in, double *__restrict__ out, long N, double gamma) {
  const long t = (long)blockIdx.x*256 + threadIdx.x;
  int q = 0;
  LD(dl); LD(pl); LD(byl); LD(bzl); LD(dr); LD(pr); LD(byr); LD(bzr); LD(bn);
  out[t] = fast_speed<true>(gamma, dl, pl, bn, byl, bzl) + fast_speed<true>(gamma, dr, pr, bn, byr, bzr);
}
// PLM of the seven variables of one cell (three cells in, two face values per variable out)
__global__ void __launch_bounds__(256) ra_plm7(const double *__restrict__ in, double *__restrict__ out, long N) {
  const long t = (long)blockIdx.x*256 + threadIdx.x;
  for (int v = 0; v < 7; ++v) {
    double up, down;
    plm(in[(3*v)*N + t], in[(3*v + 1)*N + t], in[(3*v + 2)*N + t], up, down);
    out[(2*v)*N + t] = up; out[(2*v + 1)*N + t] = down;
  }
}
// one face as the marches see it: left state given (7 values kept from the previous cell), PLM of the cell on the right
// from its three-cell stencil, HLLD, the seven flux components out
template <bool EO, bool FM, int WAVES>
__global__ void __launch_bounds__(256, WAVES) ra_face(const double *__restrict__ in, double *__restrict__ out, long N, double gamma) {
  const long t = (long)blockIdx.x*256 + threadIdx.x;
  double L[7], R[7], up[7];
  for (int v = 0; v < 7; ++v) L[v] = in[v*N + t];
  for (int v = 0; v < 7; ++v) plm(in[(7 + 3*v)*N + t], in[(8 + 3*v)*N + t], in[(9 + 3*v)*N + t], up[v], R[v]);
  const double bn = in[28*N + t];
  const Cons1D f = hlld<EO, FM>(gamma, L[0], L[1], L[2], L[3], L[4], L[5], L[6], R[0], R[1], R[2], R[3], R[4], R[5], R[6], bn);
  out[t] = f.d; out[N + t] = f.mx; out[2*N + t] = f.my; out[3*N + t] = f.mz; out[4*N + t] = f.e; out[5*N + t] = f.by;
  out[6*N + t] = f.bz;
  for (int v = 0; v < 7; ++v) out[(7 + v)*N + t] = up[v];        // the left state of the next face stays alive
}
// WAVES = waves per SIMD the allocation has to allow: 2 -> 256 VGPRs, 3 -> 168, 4 -> 128, 5 -> 96, 6 -> 80
#define INST(W) \
  template __global__ void ra_hlld<false, false, W>(const double *, double *, long, double); \
  template __global__ void ra_hlld<true, true, W>(const double *, double *, long, double);   \
  template __global__ void ra_face<false, false, W>(const double *, double *, long, double); \
  template __global__ void ra_face<true, true, W>(const double *, double *, long, double);
INST(2) INST(3) INST(4) INST(5) INST(6)
