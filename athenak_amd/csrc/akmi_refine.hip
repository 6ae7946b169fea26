// akmi_refine.hip -- the operators between a MeshBlock and its coarse buffer that SMR/AMR boundary
// exchange is made of: restriction of cell- and face-centred data, slope-limited
// prolongation of cell-centred data, and the divergence-preserving prolongation of the face field
// (shared faces first, then the Toth & Roe interior).  The mesh tree, the level-aware neighbour
// tables and the flux/EMF correction that call them in the reference are not built yet; the
// operators are exposed through the C ABI over caller-given index boxes, which is how
// src/bvals/prolongation.cpp drives them (iprol boxes of the receive buffers).
#include "akmi_common.hpp"

namespace akmi {

struct CGeo { int cN1, cN2, cN3, cis, cie, cjs, cje, cks, cke; };

static CGeo make_cgeo(const Geo &g) {        // src/mesh/mesh.cpp:286-330
  CGeo c;
  const int cnx1 = g.nx1/2, cnx2 = g.multi_d ? g.nx2/2 : 1, cnx3 = g.three_d ? g.nx3/2 : 1;
  c.cN1 = cnx1 + 2*g.ng; c.cN2 = g.multi_d ? cnx2 + 2*g.ng : 1; c.cN3 = g.three_d ? cnx3 + 2*g.ng : 1;
  c.cis = g.ng; c.cie = c.cis + cnx1 - 1;
  c.cjs = g.multi_d ? g.ng : 0; c.cje = g.multi_d ? c.cjs + cnx2 - 1 : 0;
  c.cks = g.three_d ? g.ng : 0; c.cke = g.three_d ? c.cks + cnx3 - 1 : 0;
  return c;
}

struct Box { int il, iu, jl, ju, kl, ku; };

// thread -> (m, v, k, j, i) of a box; blockIdx.z = (m*nv + v)*nk + (k - kl)
__device__ __forceinline__ bool box_index(const Box &bx, int nv, int &m, int &v, int &k, int &j, int &i) {
  i = bx.il + blockIdx.x*64 + threadIdx.x;
  j = bx.jl + blockIdx.y*4 + threadIdx.y;
  const int nk = bx.ku - bx.kl + 1;
  int z = blockIdx.z;
  k = bx.kl + z%nk; z /= nk;
  v = z%nv; m = z/nv;
  return i <= bx.iu && j <= bx.ju;
}
static dim3 box_grid(const Box &bx, int nv, int nmb) {
  return dim3(cdiv(bx.iu - bx.il + 1, 64), cdiv(bx.ju - bx.jl + 1, 4), (bx.ku - bx.kl + 1)*nv*nmb);
}

__device__ __forceinline__ double sgn1(double x) { return (x < 0.0) ? -1.0 : 1.0; }   // SIGN, athena.hpp:52
__device__ __forceinline__ double mm8(double dl, double dr) {
  return 0.125*(sgn1(dl) + sgn1(dr))*fmin(fabs(dl), fabs(dr));
}

// MeshRefinement::RestrictCC, src/mesh/mesh_refinement.cpp:1223-1277
__global__ void __launch_bounds__(256)
k_restrict_cc(Geo g, CGeo c, int nvar, const unsigned char *__restrict__ mask, const double *__restrict__ u,
              double *__restrict__ cu) {
  const Box bx{c.cis, c.cie, c.cjs, c.cje, c.cks, c.cke};
  int m, n, k, j, i;
  if (!box_index(bx, nvar, m, n, k, j, i)) return;
  if (mask && !mask[m]) return;
  const int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
  auto U = [&](int kk, int jj, int ii) { return u[ix5(nvar, g.N3, g.N2, g.N1, m, n, kk, jj, ii)]; };
  double r;
  if (!g.multi_d) r = 0.5*(U(k, j, fi) + U(k, j, fi + 1));
  else if (!g.three_d) r = 0.25*(U(k, fj, fi) + U(k, fj, fi + 1) + U(k, fj + 1, fi) + U(k, fj + 1, fi + 1));
  else r = 0.125*(U(fk, fj, fi) + U(fk, fj, fi + 1) + U(fk, fj + 1, fi) + U(fk, fj + 1, fi + 1)
                + U(fk + 1, fj, fi) + U(fk + 1, fj, fi + 1) + U(fk + 1, fj + 1, fi) + U(fk + 1, fj + 1, fi + 1));
  cu[ix5(nvar, c.cN3, c.cN2, c.cN1, m, n, k, j, i)] = r;
}

struct Faces { double *b1, *b2, *b3; };
struct CFaces { const double *b1, *b2, *b3; };

// MeshRefinement::RestrictFC, src/mesh/mesh_refinement.cpp:1283-1382
__global__ void __launch_bounds__(256)
k_restrict_fc(Geo g, CGeo c, const unsigned char *__restrict__ mask, CFaces f, Faces cf) {
  const Box bx{c.cis, c.cie, c.cjs, c.cje, c.cks, c.cke};
  int m, v, k, j, i;
  if (!box_index(bx, 1, m, v, k, j, i)) return;
  if (mask && !mask[m]) return;
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
  auto B1 = [&](int kk, int jj, int ii) { return f.b1[ix4(N3, N2, N1 + 1, m, kk, jj, ii)]; };
  auto B2 = [&](int kk, int jj, int ii) { return f.b2[ix4(N3, N2 + 1, N1, m, kk, jj, ii)]; };
  auto B3 = [&](int kk, int jj, int ii) { return f.b3[ix4(N3 + 1, N2, N1, m, kk, jj, ii)]; };
  auto C1 = [&](int kk, int jj, int ii) -> double & { return cf.b1[ix4(c.cN3, c.cN2, c.cN1 + 1, m, kk, jj, ii)]; };
  auto C2 = [&](int kk, int jj, int ii) -> double & { return cf.b2[ix4(c.cN3, c.cN2 + 1, c.cN1, m, kk, jj, ii)]; };
  auto C3 = [&](int kk, int jj, int ii) -> double & { return cf.b3[ix4(c.cN3 + 1, c.cN2, c.cN1, m, kk, jj, ii)]; };
  if (!g.multi_d) {
    C1(k, j, i) = B1(k, j, fi);
    if (i == c.cie) C1(k, j, i + 1) = B1(k, j, fi + 2);
    const double b2c = 0.5*(B2(k, j, fi) + B2(k, j, fi + 1));
    C2(k, j, i) = b2c; C2(k, j + 1, i) = b2c;
    const double b3c = 0.5*(B3(k, j, fi) + B3(k, j, fi + 1));
    C3(k, j, i) = b3c; C3(k + 1, j, i) = b3c;
  } else if (!g.three_d) {
    C1(k, j, i) = 0.5*(B1(k, fj, fi) + B1(k, fj + 1, fi));
    if (i == c.cie) C1(k, j, i + 1) = 0.5*(B1(k, fj, fi + 2) + B1(k, fj + 1, fi + 2));
    C2(k, j, i) = 0.5*(B2(k, fj, fi) + B2(k, fj, fi + 1));
    if (j == c.cje) C2(k, j + 1, i) = 0.5*(B2(k, fj + 2, fi) + B2(k, fj + 2, fi + 1));
    const double b3c = 0.25*(B3(k, fj, fi) + B3(k, fj, fi + 1) + B3(k, fj + 1, fi) + B3(k, fj + 1, fi + 1));
    C3(k, j, i) = b3c; C3(k + 1, j, i) = b3c;
  } else {
    C1(k, j, i) = 0.25*(B1(fk, fj, fi) + B1(fk, fj + 1, fi) + B1(fk + 1, fj, fi) + B1(fk + 1, fj + 1, fi));
    if (i == c.cie)
      C1(k, j, i + 1) = 0.25*(B1(fk, fj, fi + 2) + B1(fk, fj + 1, fi + 2) + B1(fk + 1, fj, fi + 2) + B1(fk + 1, fj + 1, fi + 2));
    C2(k, j, i) = 0.25*(B2(fk, fj, fi) + B2(fk, fj, fi + 1) + B2(fk + 1, fj, fi) + B2(fk + 1, fj, fi + 1));
    if (j == c.cje)
      C2(k, j + 1, i) = 0.25*(B2(fk, fj + 2, fi) + B2(fk, fj + 2, fi + 1) + B2(fk + 1, fj + 2, fi) + B2(fk + 1, fj + 2, fi + 1));
    C3(k, j, i) = 0.25*(B3(fk, fj, fi) + B3(fk, fj, fi + 1) + B3(fk, fj + 1, fi) + B3(fk, fj + 1, fi + 1));
    if (k == c.cke)
      C3(k + 1, j, i) = 0.25*(B3(fk + 2, fj, fi) + B3(fk + 2, fj, fi + 1) + B3(fk + 2, fj + 1, fi) + B3(fk + 2, fj + 1, fi + 1));
  }
}

// ProlongCC, src/mesh/prolongation.hpp:19-63
__global__ void __launch_bounds__(256)
k_prolong_cc(Geo g, CGeo c, Box bx, int nvar, const double *__restrict__ cu, double *__restrict__ u) {
  int m, v, k, j, i;
  if (!box_index(bx, nvar, m, v, k, j, i)) return;
  const int fi = (i - c.cis)*2 + g.is, fj = (j - c.cjs)*2 + g.js, fk = (k - c.cks)*2 + g.ks;
  auto CA = [&](int kk, int jj, int ii) { return cu[ix5(nvar, c.cN3, c.cN2, c.cN1, m, v, kk, jj, ii)]; };
  auto A = [&](int kk, int jj, int ii) -> double & { return u[ix5(nvar, g.N3, g.N2, g.N1, m, v, kk, jj, ii)]; };
  const double q = CA(k, j, i);
  const double dvar1 = mm8(q - CA(k, j, i - 1), CA(k, j, i + 1) - q);
  double dvar2 = 0.0, dvar3 = 0.0;
  if (g.multi_d) dvar2 = mm8(q - CA(k, j - 1, i), CA(k, j + 1, i) - q);
  if (g.three_d) dvar3 = mm8(q - CA(k - 1, j, i), CA(k + 1, j, i) - q);
  A(fk, fj, fi) = q - dvar1 - dvar2 - dvar3;
  A(fk, fj, fi + 1) = q + dvar1 - dvar2 - dvar3;
  if (g.multi_d) {
    A(fk, fj + 1, fi) = q - dvar1 + dvar2 - dvar3;
    A(fk, fj + 1, fi + 1) = q + dvar1 + dvar2 - dvar3;
  }
  if (g.three_d) {
    A(fk + 1, fj, fi) = q - dvar1 - dvar2 + dvar3;
    A(fk + 1, fj, fi + 1) = q + dvar1 - dvar2 + dvar3;
    A(fk + 1, fj + 1, fi) = q - dvar1 + dvar2 + dvar3;
    A(fk + 1, fj + 1, fi + 1) = q + dvar1 + dvar2 + dvar3;
  }
}

// ProlongFCSharedX1Face / X2Face / X3Face, src/mesh/prolongation.hpp:69-160
template <int COMP>
__global__ void __launch_bounds__(256)
k_prolong_fc_shared(Geo g, CGeo c, Box bx, const double *__restrict__ cb, double *__restrict__ b) {
  int m, v, k, j, i;
  if (!box_index(bx, 1, m, v, k, j, i)) return;
  const int fi = (i - c.cis)*2 + g.is;
  const int fj = g.multi_d ? (j - c.cjs)*2 + g.js : j;
  const int fk = g.three_d ? (k - c.cks)*2 + g.ks : k;
  constexpr int d1 = COMP == 0, d2 = COMP == 1, d3 = COMP == 2;
  auto CB = [&](int kk, int jj, int ii) { return cb[ix4(c.cN3 + d3, c.cN2 + d2, c.cN1 + d1, m, kk, jj, ii)]; };
  auto B = [&](int kk, int jj, int ii) -> double & { return b[ix4(g.N3 + d3, g.N2 + d2, g.N1 + d1, m, kk, jj, ii)]; };
  const double q = CB(k, j, i);
  if constexpr (COMP == 0) {
    double dvar2 = 0.0, dvar3 = 0.0;
    if (g.multi_d) dvar2 = mm8(q - CB(k, j - 1, i), CB(k, j + 1, i) - q);
    if (g.three_d) dvar3 = mm8(q - CB(k - 1, j, i), CB(k + 1, j, i) - q);
    B(fk, fj, fi) = q - dvar2 - dvar3;
    if (g.multi_d) B(fk, fj + 1, fi) = q + dvar2 - dvar3;
    if (g.three_d) {
      B(fk + 1, fj, fi) = q - dvar2 + dvar3;
      B(fk + 1, fj + 1, fi) = q + dvar2 + dvar3;
    }
  } else if constexpr (COMP == 1) {
    const double dvar1 = mm8(q - CB(k, j, i - 1), CB(k, j, i + 1) - q);
    double dvar3 = 0.0;
    if (g.three_d) dvar3 = mm8(q - CB(k - 1, j, i), CB(k + 1, j, i) - q);
    B(fk, fj, fi) = q - dvar1 - dvar3;
    B(fk, fj, fi + 1) = q + dvar1 - dvar3;
    if (g.three_d) {
      B(fk + 1, fj, fi) = q - dvar1 + dvar3;
      B(fk + 1, fj, fi + 1) = q + dvar1 + dvar3;
    }
  } else {
    const double dvar1 = mm8(q - CB(k, j, i - 1), CB(k, j, i + 1) - q);
    double dvar2 = 0.0;
    if (g.multi_d) dvar2 = mm8(q - CB(k, j - 1, i), CB(k, j + 1, i) - q);
    B(fk, fj, fi) = q - dvar1 - dvar2;
    B(fk, fj, fi + 1) = q + dvar1 - dvar2;
    if (g.multi_d) {
      B(fk, fj + 1, fi) = q - dvar1 + dvar2;
      B(fk, fj + 1, fi + 1) = q + dvar1 + dvar2;
    }
  }
}

// ProlongFCInternal, src/mesh/prolongation.hpp:166-230; 1-D: src/bvals/prolongation.cpp:765-770
__global__ void __launch_bounds__(256)
k_prolong_fc_internal(Geo g, CGeo c, Box bx, Faces f) {
  int m, v, k, j, i;
  if (!box_index(bx, 1, m, v, k, j, i)) return;
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int fi = (i - c.cis)*2 + g.is, fj = (j - c.cjs)*2 + g.js, fk = (k - c.cks)*2 + g.ks;
  auto B1 = [&](int kk, int jj, int ii) -> double & { return f.b1[ix4(N3, N2, N1 + 1, m, kk, jj, ii)]; };
  auto B2 = [&](int kk, int jj, int ii) -> double & { return f.b2[ix4(N3, N2 + 1, N1, m, kk, jj, ii)]; };
  auto B3 = [&](int kk, int jj, int ii) -> double & { return f.b3[ix4(N3 + 1, N2, N1, m, kk, jj, ii)]; };
  if (!g.multi_d) {
    B1(fk, fj, fi + 1) = 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2));
  } else if (g.three_d) {
    double Uxx = 0.0, Vyy = 0.0, Wzz = 0.0, Uxyz = 0.0, Vxyz = 0.0, Wxyz = 0.0;
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int jsgn = 2*jj - 1;
      const int fjj = fj + jj, fjp = fj + 2*jj;
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        const int isgn = 2*ii - 1;
        const int fii = fi + ii, fip = fi + 2*ii;
        Uxx += isgn*(jsgn*(B2(fk, fjp, fii) + B2(fk + 1, fjp, fii)) + (B3(fk + 2, fjj, fii) - B3(fk, fjj, fii)));
        Vyy += jsgn*((B3(fk + 2, fjj, fii) - B3(fk, fjj, fii)) + isgn*(B1(fk, fjj, fip) + B1(fk + 1, fjj, fip)));
        Wzz += isgn*(B1(fk + 1, fjj, fip) - B1(fk, fjj, fip)) + jsgn*(B2(fk + 1, fjp, fii) - B2(fk, fjp, fii));
        Uxyz += isgn*jsgn*(B1(fk + 1, fjj, fip) - B1(fk, fjj, fip));
        Vxyz += isgn*jsgn*(B2(fk + 1, fjp, fii) - B2(fk, fjp, fii));
        Wxyz += isgn*jsgn*(B3(fk + 2, fjj, fii) - B3(fk, fjj, fii));
      }
    }
    Uxx *= 0.125; Vyy *= 0.125; Wzz *= 0.125;
    Uxyz *= 0.0625; Vxyz *= 0.0625; Wxyz *= 0.0625;
    B1(fk, fj, fi + 1) = 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2)) + Uxx - Vxyz - Wxyz;
    B1(fk, fj + 1, fi + 1) = 0.5*(B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2)) + Uxx - Vxyz + Wxyz;
    B1(fk + 1, fj, fi + 1) = 0.5*(B1(fk + 1, fj, fi) + B1(fk + 1, fj, fi + 2)) + Uxx + Vxyz - Wxyz;
    B1(fk + 1, fj + 1, fi + 1) = 0.5*(B1(fk + 1, fj + 1, fi) + B1(fk + 1, fj + 1, fi + 2)) + Uxx + Vxyz + Wxyz;
    B2(fk, fj + 1, fi) = 0.5*(B2(fk, fj, fi) + B2(fk, fj + 2, fi)) + Vyy - Uxyz - Wxyz;
    B2(fk, fj + 1, fi + 1) = 0.5*(B2(fk, fj, fi + 1) + B2(fk, fj + 2, fi + 1)) + Vyy - Uxyz + Wxyz;
    B2(fk + 1, fj + 1, fi) = 0.5*(B2(fk + 1, fj, fi) + B2(fk + 1, fj + 2, fi)) + Vyy + Uxyz - Wxyz;
    B2(fk + 1, fj + 1, fi + 1) = 0.5*(B2(fk + 1, fj, fi + 1) + B2(fk + 1, fj + 2, fi + 1)) + Vyy + Uxyz + Wxyz;
    B3(fk + 1, fj, fi) = 0.5*(B3(fk + 2, fj, fi) + B3(fk, fj, fi)) + Wzz - Uxyz - Vxyz;
    B3(fk + 1, fj, fi + 1) = 0.5*(B3(fk + 2, fj, fi + 1) + B3(fk, fj, fi + 1)) + Wzz - Uxyz + Vxyz;
    B3(fk + 1, fj + 1, fi) = 0.5*(B3(fk + 2, fj + 1, fi) + B3(fk, fj + 1, fi)) + Wzz + Uxyz - Vxyz;
    B3(fk + 1, fj + 1, fi + 1) = 0.5*(B3(fk + 2, fj + 1, fi + 1) + B3(fk, fj + 1, fi + 1)) + Wzz + Uxyz + Vxyz;
  } else {
    const double tmp1 = 0.25*(B2(fk, fj + 2, fi + 1) - B2(fk, fj, fi + 1) - B2(fk, fj + 2, fi) + B2(fk, fj, fi));
    const double tmp2 = 0.25*(B1(fk, fj, fi) - B1(fk, fj, fi + 2) - B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2));
    B1(fk, fj, fi + 1) = 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2)) + tmp1;
    B1(fk, fj + 1, fi + 1) = 0.5*(B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2)) + tmp1;
    B2(fk, fj + 1, fi) = 0.5*(B2(fk, fj, fi) + B2(fk, fj + 2, fi)) + tmp2;
    B2(fk, fj + 1, fi + 1) = 0.5*(B2(fk, fj, fi + 1) + B2(fk, fj + 2, fi + 1)) + tmp2;
  }
}

// box of coarse indices: the two fine cells (faces) of every coarse index must lie inside the fine
// array, and the slope stencil (one coarse cell/face each side, `halo`) inside the coarse array
static int check_box(const Geo &g, const CGeo &c, const int *box, int halo, int d1, int d2, int d3,
                     const char *who) {
  if (g.nx1 % 2 || (g.multi_d && g.nx2 % 2) || (g.three_d && g.nx3 % 2)) {
    set_error("%s: MeshBlock sizes must be even for refinement", who); return AKMI_FAIL;
  }
  const int lo[3] = {box[0], box[2], box[4]}, hi[3] = {box[1], box[3], box[5]};
  const int cs[3] = {c.cis, c.cjs, c.cks}, fs[3] = {g.is, g.js, g.ks};
  const int cN[3] = {c.cN1 + d1, c.cN2 + d2, c.cN3 + d3}, fN[3] = {g.N1 + d1, g.N2 + d2, g.N3 + d3};
  const bool on[3] = {true, (bool)g.multi_d, (bool)g.three_d};
  const int own[3] = {d1, d2, d3};
  for (int q = 0; q < 3; ++q) {
    if (lo[q] > hi[q]) { set_error("%s: empty index box", who); return AKMI_FAIL; }
    if (!on[q]) {
      if (lo[q] != 0 || hi[q] != 0) { set_error("%s: index box in a collapsed direction", who); return AKMI_FAIL; }
      continue;
    }
    const int flo = (lo[q] - cs[q])*2 + fs[q];
    const int fhi = (hi[q] - cs[q])*2 + fs[q] + (own[q] ? 0 : 1);     // a face maps to ONE fine face
    const int h = own[q] ? 0 : halo;                                  // no slope along a face's own axis
    if (flo < 0 || fhi > fN[q] - 1 || lo[q] - h < 0 || hi[q] + h > cN[q] - 1) {
      set_error("%s: index box [%d,%d] of direction %d reaches outside the arrays", who, lo[q], hi[q], q + 1);
      return AKMI_FAIL;
    }
  }
  return AKMI_COMPLETE;
}

}  // namespace akmi

using namespace akmi;

// Restricted face fluxes for a coarser neighbour, buffer order of PackAndSendFluxCC
// (src/bvals/flux_correct_cc.cpp:78-148); box = coarse index box, one face thick along DIR
template <int DIR>
__global__ void k_restrict_flux_cc(Geo g, CGeo c, Box bx, int nvar, const double *__restrict__ flx,
                                   double *__restrict__ out) {
  int m, v, k, j, i;
  if (!box_index(bx, nvar, m, v, k, j, i)) return;
  const int ni = bx.iu - bx.il + 1, nj = bx.ju - bx.jl + 1, nk = bx.ku - bx.kl + 1;
  const int f3 = g.N3 + (DIR == 2), f2 = g.N2 + (DIR == 1), f1 = g.N1 + (DIR == 0);
  const int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
#define FX(k, j, i) flx[ix5(nvar, f3, f2, f1, m, v, k, j, i)]
  double r;
  size_t o;
  if constexpr (DIR == 0) {
    if (!g.multi_d) r = FX(0, 0, fi);
    else if (!g.three_d) r = 0.5*(FX(0, fj, fi) + FX(0, fj + 1, fi));
    else r = 0.25*(FX(fk, fj, fi) + FX(fk, fj + 1, fi) + FX(fk + 1, fj, fi) + FX(fk + 1, fj + 1, fi));
    o = (size_t)(j - bx.jl) + (size_t)nj*((k - bx.kl) + (size_t)nk*v);
  } else if constexpr (DIR == 1) {
    if (!g.three_d) r = 0.5*(FX(0, fj, fi) + FX(0, fj, fi + 1));
    else r = 0.25*(FX(fk, fj, fi) + FX(fk, fj, fi + 1) + FX(fk + 1, fj, fi) + FX(fk + 1, fj, fi + 1));
    o = (size_t)(i - bx.il) + (size_t)ni*((k - bx.kl) + (size_t)nk*v);
  } else {
    r = 0.25*(FX(fk, fj, fi) + FX(fk, fj, fi + 1) + FX(fk, fj + 1, fi) + FX(fk, fj + 1, fi + 1));
    o = (size_t)(i - bx.il) + (size_t)ni*((j - bx.jl) + (size_t)nj*v);
  }
#undef FX
  out[(size_t)m*nvar*ni*nj*nk + o] = r;
}

// Restricted edge EMFs for a coarser neighbour (PackAndSendFluxFC, src/bvals/flux_correct_fc.cpp:84-360):
// the two fine edges of a coarse edge averaged along the edge's own direction
template <int COMP>
__global__ void k_restrict_emf(Geo g, CGeo c, Box bx, const double *__restrict__ e,
                               double *__restrict__ out) {
  int m, v, k, j, i;
  if (!box_index(bx, 1, m, v, k, j, i)) return;
  const int ni = bx.iu - bx.il + 1, nj = bx.ju - bx.jl + 1, nk = bx.ku - bx.kl + 1;
  const int e3 = g.N3 + (COMP != 2), e2 = g.N2 + (COMP != 1), e1 = g.N1 + (COMP != 0);
  const int fi = 2*i - c.cis, fj = g.multi_d ? 2*j - c.cjs : 0, fk = g.three_d ? 2*k - c.cks : 0;
#define EE(k, j, i) e[ix4(e3, e2, e1, m, k, j, i)]
  double r;
  if constexpr (COMP == 0) r = g.multi_d ? 0.5*(EE(fk, fj, fi) + EE(fk, fj, fi + 1)) : EE(fk, fj, fi);
  else if constexpr (COMP == 1) r = g.multi_d ? 0.5*(EE(fk, fj, fi) + EE(fk, fj + 1, fi)) : EE(fk, fj, fi);
  else r = g.three_d ? 0.5*(EE(fk, fj, fi) + EE(fk + 1, fj, fi)) : EE(fk, fj, fi);
#undef EE
  out[(size_t)m*ni*nj*nk + (size_t)(i - bx.il) + (size_t)ni*((j - bx.jl) + (size_t)nj*(k - bx.kl))] = r;
}

// coarse box of a flux operator: inside the coarse index space incl. its upper faces/edges
static int check_flux_box(const CGeo &c, const int *box, const char *who) {
  if (box[0] > box[1] || box[2] > box[3] || box[4] > box[5] || box[0] < c.cis || box[1] > c.cie + 1 ||
      box[2] < c.cjs || box[3] > c.cje + 1 || box[4] < c.cks || box[5] > c.cke + 1) {
    set_error("%s: box [%d,%d]x[%d,%d]x[%d,%d] outside the coarse faces [%d,%d]x[%d,%d]x[%d,%d]", who, box[0],
              box[1], box[2], box[3], box[4], box[5], c.cis, c.cie + 1, c.cjs, c.cje + 1, c.cks, c.cke + 1);
    return AKMI_FAIL;
  }
  return AKMI_COMPLETE;
}

// Primitive -> conserved over a box of fine cells: SingleP2C_* (src/eos/ideal_c2p_hyd.hpp:76-83,
// ideal_c2p_mhd.hpp:75-84) as PrimToConsFineBndry applies them (src/bvals/prolong_prims.cpp:190-300)
__global__ void k_prim2cons(Geo g, Box bx, int is_ideal, const double *__restrict__ w,
                            const double *__restrict__ bcc, double *__restrict__ u) {
  int m, v, k, j, i;
  if (!box_index(bx, 1, m, v, k, j, i)) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  const double d = w[c], vx = w[c + cs], vy = w[c + 2*cs], vz = w[c + 3*cs];
  u[c] = d; u[c + cs] = d*vx; u[c + 2*cs] = d*vy; u[c + 3*cs] = d*vz;
  if (is_ideal) {
    if (bcc) {
      const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
      const double bx_ = bcc[b], by = bcc[b + cs], bz = bcc[b + 2*cs];
      u[c + 4*cs] = w[c + 4*cs] + 0.5*(d*(vx*vx + vy*vy + vz*vz) + (bx_*bx_ + by*by + bz*bz));
    } else {
      u[c + 4*cs] = w[c + 4*cs] + 0.5*d*(vx*vx + vy*vy + vz*vz);
    }
  }
  for (int n = is_ideal ? 5 : 4; n < g.nvar; ++n) u[c + n*cs] = d*w[c + n*cs];
}

extern "C" {

int akmi_restrict_cc_masked(const akmi_pack *p, int nvar, const unsigned char *mask, const double *u, double *cu,
                            void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  const Box bx{c.cis, c.cie, c.cjs, c.cje, c.cks, c.cke};
  k_restrict_cc<<<box_grid(bx, nvar, g.nmb), dim3(64, 4), 0, (hipStream_t)stream>>>(g, c, nvar, mask, u, cu);
  AKMI_CHECK_LAUNCH("restrict_cc");
  return AKMI_COMPLETE;
}
int akmi_restrict_cc(const akmi_pack *p, int nvar, const double *u, double *cu, void *stream) {
  return akmi_restrict_cc_masked(p, nvar, nullptr, u, cu, stream);
}

int akmi_restrict_fc_masked(const akmi_pack *p, const unsigned char *mask, const double *bx1f, const double *bx2f,
                            const double *bx3f, double *cbx1f, double *cbx2f, double *cbx3f, void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  const Box bx{c.cis, c.cie, c.cjs, c.cje, c.cks, c.cke};
  k_restrict_fc<<<box_grid(bx, 1, g.nmb), dim3(64, 4), 0, (hipStream_t)stream>>>(
      g, c, mask, CFaces{bx1f, bx2f, bx3f}, Faces{cbx1f, cbx2f, cbx3f});
  AKMI_CHECK_LAUNCH("restrict_fc");
  return AKMI_COMPLETE;
}
int akmi_restrict_fc(const akmi_pack *p, const double *bx1f, const double *bx2f, const double *bx3f,
                     double *cbx1f, double *cbx2f, double *cbx3f, void *stream) {
  return akmi_restrict_fc_masked(p, nullptr, bx1f, bx2f, bx3f, cbx1f, cbx2f, cbx3f, stream);
}

int akmi_restrict_flux_cc(const akmi_pack *p, int nvar, int dir, const int *box, const double *flx,
                          double *out, void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  if (dir < 0 || dir > 2 || (dir == 1 && !g.multi_d) || (dir == 2 && !g.three_d)) {
    set_error("restrict_flux_cc: dir = %d", dir); return AKMI_FAIL;
  }
  if (check_flux_box(c, box, "restrict_flux_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  if (box[2*dir] != box[2*dir + 1]) { set_error("restrict_flux_cc: the box must be one face thick along dir"); return AKMI_FAIL; }
  // transverse extents are cells, not faces
  const int hi[3] = {c.cie, c.cje, c.cke};
  for (int d = 0; d < 3; ++d)
    if (d != dir && box[2*d + 1] > hi[d]) { set_error("restrict_flux_cc: transverse range beyond the coarse cells"); return AKMI_FAIL; }
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  const dim3 grid = box_grid(bx, nvar, g.nmb), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
  if (dir == 0) k_restrict_flux_cc<0><<<grid, block, 0, st>>>(g, c, bx, nvar, flx, out);
  else if (dir == 1) k_restrict_flux_cc<1><<<grid, block, 0, st>>>(g, c, bx, nvar, flx, out);
  else k_restrict_flux_cc<2><<<grid, block, 0, st>>>(g, c, bx, nvar, flx, out);
  AKMI_CHECK_LAUNCH("restrict_flux_cc");
  return AKMI_COMPLETE;
}

int akmi_restrict_emf(const akmi_pack *p, int comp, const int *box, const double *e, double *out,
                      void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  if (comp < 0 || comp > 2) { set_error("restrict_emf: comp = %d", comp); return AKMI_FAIL; }
  if (check_flux_box(c, box, "restrict_emf") != AKMI_COMPLETE) return AKMI_FAIL;
  const int hi[3] = {c.cie, c.cje, c.cke};
  if (box[2*comp + 1] > hi[comp]) { set_error("restrict_emf: range along the edge beyond the coarse cells"); return AKMI_FAIL; }
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  const dim3 grid = box_grid(bx, 1, g.nmb), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
  if (comp == 0) k_restrict_emf<0><<<grid, block, 0, st>>>(g, c, bx, e, out);
  else if (comp == 1) k_restrict_emf<1><<<grid, block, 0, st>>>(g, c, bx, e, out);
  else k_restrict_emf<2><<<grid, block, 0, st>>>(g, c, bx, e, out);
  AKMI_CHECK_LAUNCH("restrict_emf");
  return AKMI_COMPLETE;
}

int akmi_prim2cons(const akmi_pack *p, const int *box, const double *w, const double *bcc, double *u,
                   void *stream) {
  Geo g = make_geo(p);
  if (box[0] > box[1] || box[2] > box[3] || box[4] > box[5] || box[0] < 0 || box[1] >= g.N1 ||
      box[2] < 0 || box[3] >= g.N2 || box[4] < 0 || box[5] >= g.N3) {
    set_error("prim2cons: box [%d,%d]x[%d,%d]x[%d,%d] outside the block's cells", box[0], box[1], box[2],
              box[3], box[4], box[5]);
    return AKMI_FAIL;
  }
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  k_prim2cons<<<box_grid(bx, 1, g.nmb), dim3(64, 4), 0, (hipStream_t)stream>>>(g, bx, p->is_ideal, w, bcc, u);
  AKMI_CHECK_LAUNCH("prim2cons");
  return AKMI_COMPLETE;
}

int akmi_prolong_cc(const akmi_pack *p, int nvar, const int *box, const double *cu, double *u,
                    void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  if (check_box(g, c, box, 1, 0, 0, 0, "prolong_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  k_prolong_cc<<<box_grid(bx, nvar, g.nmb), dim3(64, 4), 0, (hipStream_t)stream>>>(g, c, bx, nvar, cu, u);
  AKMI_CHECK_LAUNCH("prolong_cc");
  return AKMI_COMPLETE;
}

int akmi_prolong_fc_shared(const akmi_pack *p, int comp, const int *box, const double *cb, double *b,
                           void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  if (comp < 0 || comp > 2) { set_error("prolong_fc_shared: comp = %d", comp); return AKMI_FAIL; }
  if (check_box(g, c, box, 1, comp == 0, comp == 1, comp == 2, "prolong_fc_shared") != AKMI_COMPLETE)
    return AKMI_FAIL;
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  const dim3 grid = box_grid(bx, 1, g.nmb), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
  if (comp == 0) k_prolong_fc_shared<0><<<grid, block, 0, st>>>(g, c, bx, cb, b);
  else if (comp == 1) k_prolong_fc_shared<1><<<grid, block, 0, st>>>(g, c, bx, cb, b);
  else k_prolong_fc_shared<2><<<grid, block, 0, st>>>(g, c, bx, cb, b);
  AKMI_CHECK_LAUNCH("prolong_fc_shared");
  return AKMI_COMPLETE;
}

int akmi_prolong_fc_internal(const akmi_pack *p, const int *box, double *bx1f, double *bx2f,
                             double *bx3f, void *stream) {
  Geo g = make_geo(p); CGeo c = make_cgeo(g);
  if (check_box(g, c, box, 0, 0, 0, 0, "prolong_fc_internal") != AKMI_COMPLETE) return AKMI_FAIL;
  const Box bx{box[0], box[1], box[2], box[3], box[4], box[5]};
  k_prolong_fc_internal<<<box_grid(bx, 1, g.nmb), dim3(64, 4), 0, (hipStream_t)stream>>>(
      g, c, bx, Faces{bx1f, bx2f, bx3f});
  AKMI_CHECK_LAUNCH("prolong_fc_internal");
  return AKMI_COMPLETE;
}

}  // extern "C"
