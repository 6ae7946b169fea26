// Test infrastructure: athenak_amd/csrc/akmi_numerics.hpp compiled for the CPU (tests/host_shim/hip/hip_runtime.h stands
// in for the HIP runtime header) behind a C ABI, so that tests/test_numerics_host.py can compare the product's per-cell /
// per-face arithmetic with the oracle's single-state functions (oracle/akref.h: akref_plm ... akref_hlld) bit for bit on
// a machine without a GPU.  A "wave" is one lane here: the wave-uniform early-outs of hlld<EO> become per-face branches.
#include <hip/hip_runtime.h>
#include "akmi_numerics.hpp"

using namespace akmi;

extern "C" {

// kind: 1 plm, 2 ppm4, 3 ppmx, 4 wenoz, 5 teno; st = (b2, b1, here, a1, a2) (plm uses the inner three)
void hn_recon(int kind, const double *st, double *up, double *down) {
  if (kind == 1) plm(st[1], st[2], st[3], *up, *down);
  else if (kind == 2) ppm4(st[0], st[1], st[2], st[3], st[4], *up, *down);
  else if (kind == 3) ppmx(st[0], st[1], st[2], st[3], st[4], *up, *down);
  else if (kind == 4) wenoz(st[0], st[1], st[2], st[3], st[4], *up, *down);
  else teno(st[0], st[1], st[2], st[3], st[4], *up, *down);
}
// kind: AKMI_RS_* 0 llf, 1 hlle, 2 hllc, 4 roe
void hn_riemann_hyd(int kind, double gamma, const double *l, const double *r, double *f) {
  if (kind == 0) riemann_hyd<0>(gamma, l[0], l[1], l[2], l[3], l[4], r[0], r[1], r[2], r[3], r[4], f[0], f[1], f[2], f[3], f[4]);
  else if (kind == 1) riemann_hyd<1>(gamma, l[0], l[1], l[2], l[3], l[4], r[0], r[1], r[2], r[3], r[4], f[0], f[1], f[2], f[3], f[4]);
  else if (kind == 4) riemann_hyd<4>(gamma, l[0], l[1], l[2], l[3], l[4], r[0], r[1], r[2], r[3], r[4], f[0], f[1], f[2], f[3], f[4]);
  else riemann_hyd<2>(gamma, l[0], l[1], l[2], l[3], l[4], r[0], r[1], r[2], r[3], r[4], f[0], f[1], f[2], f[3], f[4]);
}
// kind: 0 llf, 1 hlle, 3 hlld, 13 hlld with the early-outs; f = (d, mx, my, mz, E, F(by), F(bz))
void hn_riemann_mhd(int kind, double gamma, const double *l, const double *r, double bn, double *f) {
  Cons1D c;
#define A_ gamma, l[0], l[1], l[2], l[3], l[4], l[5], l[6], r[0], r[1], r[2], r[3], r[4], r[5], r[6], bn
  if (kind == 0) c = riemann_mhd<0>(A_);
  else if (kind == 1) c = riemann_mhd<1>(A_);
  else if (kind == 13) c = riemann_mhd<3, true, false>(A_);
  else c = riemann_mhd<3, false, false>(A_);
#undef A_
  f[0] = c.d; f[1] = c.mx; f[2] = c.my; f[3] = c.mz; f[4] = c.e; f[5] = c.by; f[6] = c.bz;
}
// the same over n faces (l, r: n x 7, bn: n, f: n x 7), so that a test can run a million faces quickly
void hn_riemann_mhd_n(int kind, double gamma, long n, const double *l, const double *r, const double *bn, double *f) {
  for (long i = 0; i < n; ++i) hn_riemann_mhd(kind, gamma, l + 7*i, r + 7*i, bn[i], f + 7*i);
}
void hn_riemann_hyd_n(int kind, double gamma, long n, const double *l, const double *r, double *f) {
  for (long i = 0; i < n; ++i) hn_riemann_hyd(kind, gamma, l + 5*i, r + 5*i, f + 5*i);
}
void hn_recon_n(int kind, long n, const double *st, double *up, double *down) {
  for (long i = 0; i < n; ++i) hn_recon(kind, st + 5*i, up + i, down + i);
}

}  // extern "C"
