// akmi_common.hpp -- geometry of a MeshBlockPack on the device + launch helpers.
#ifndef AKMI_COMMON_HPP_
#define AKMI_COMMON_HPP_
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/akmi.h"
#include "akmi_numerics.hpp"

namespace akmi {

// RegionIndcs (src/mesh/mesh.hpp:35-41, src/mesh/mesh.cpp:285-330) + array extents
struct Geo {
  int nmb, nvar, nx1, nx2, nx3, ng;
  int N1, N2, N3;
  int is, ie, js, je, ks, ke;
  int multi_d, three_d;
  const double *dx;  // device [nmb][3]
};

inline Geo make_geo(const akmi_pack *p) {
  Geo g;
  g.nmb = p->nmb; g.nvar = p->nvar; g.nx1 = p->nx1; g.nx2 = p->nx2; g.nx3 = p->nx3;
  g.ng = p->ng;
  g.multi_d = p->nx2 > 1; g.three_d = p->nx3 > 1;
  g.N1 = p->nx1 + 2*p->ng;
  g.N2 = g.multi_d ? p->nx2 + 2*p->ng : 1;
  g.N3 = g.three_d ? p->nx3 + 2*p->ng : 1;
  g.is = p->ng; g.ie = g.is + p->nx1 - 1;
  g.js = g.multi_d ? p->ng : 0; g.je = g.multi_d ? g.js + p->nx2 - 1 : 0;
  g.ks = g.three_d ? p->ng : 0; g.ke = g.three_d ? g.ks + p->nx3 - 1 : 0;
  g.dx = p->dx;
  return g;
}

inline Eos make_eos(const akmi_pack *p) {
  Eos e;
  e.gamma = p->gamma; e.dfloor = p->dfloor; e.pfloor = p->pfloor; e.tfloor = p->tfloor;
  e.sfloor = p->sfloor; e.sigma_max = p->sigma_max;
  return e;
}

// LayoutRight offsets (src/athena.hpp:111,127-128)
__host__ __device__ __forceinline__ size_t ix5(int nv, int n3, int n2, int n1, int m, int n,
                                               int k, int j, int i) {
  return ((((size_t)m*nv + n)*n3 + k)*n2 + j)*n1 + i;
}
__host__ __device__ __forceinline__ size_t ix4(int n3, int n2, int n1, int m, int k, int j,
                                               int i) {
  return (((size_t)m*n3 + k)*n2 + j)*n1 + i;
}

void set_error(const char *fmt, ...);

#define AKMI_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      akmi::set_error("%s: %s", name, hipGetErrorString(e_));                \
      return AKMI_FAIL;                                                      \
    }                                                                        \
  } while (0)

inline int cdiv(int a, int b) { return (a + b - 1)/b; }

}  // namespace akmi
#endif
