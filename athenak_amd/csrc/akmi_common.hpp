// akmi_common.hpp -- geometry of a MeshBlockPack on the device + launch helpers.
#ifndef AKMI_COMMON_HPP_
#define AKMI_COMMON_HPP_
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>
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
  e.iso_cs = p->iso_cs; e.is_ideal = p->is_ideal;
  return e;
}

inline FaceEos make_face_eos(const akmi_pack *p) {
  FaceEos e;
  e.gamma = p->gamma; e.dfloor = p->dfloor;
  e.efloor = p->is_ideal ? p->pfloor/(p->gamma - 1.0) : 0.0;
  e.iso_cs = p->iso_cs;
  return e;
}

void set_error(const char *fmt, ...);

// reconstruction x Riemann solver of a flux launch; the kernels are templated on both, the
// (grid-uniform) run-time choice is resolved here on the host, as ReconDispatch and the
// CalculateFluxes<rsolver> templates do in the reference (recon.hpp:134-185)
struct Scheme {
  int recon, rsolver;
  FaceEos eos;
  bool iso = false;       // EOS_Data::is_ideal == false (task-granular kernels only)
};
template <int V> using IC = std::integral_constant<int, V>;

template <class F>
inline int dispatch_recon(int recon, F &&f) {
#ifdef AKMI_DEV_FAST      // developer builds (tools/build_variant.sh -DAKMI_DEV_FAST): PLM only, compiles in seconds
  if (recon == AKMI_RECON_PLM) return f(IC<1>{});
  if (AKMI_DEV_FAST + 0 == 2 && recon == AKMI_RECON_PPM4) return f(IC<2>{});     // -DAKMI_DEV_FAST=2: + PPM4
#else
  switch (recon) {
    case AKMI_RECON_DC:    return f(IC<0>{});
    case AKMI_RECON_PLM:   return f(IC<1>{});
    case AKMI_RECON_PPM4:  return f(IC<2>{});
    case AKMI_RECON_PPMX:  return f(IC<3>{});
    case AKMI_RECON_WENOZ: return f(IC<4>{});
    case AKMI_RECON_TENO:  return f(IC<5>{});
  }
#endif
  set_error("reconstruct = %d not implemented", recon);
  return AKMI_FAIL;
}

// ADV: also accept rsolver = advect (kinematic runs; instantiated for the task-granular flux
// kernels only)
template <bool MHD, bool ADV = false, class F>
inline int dispatch_rsolver(int rs, F &&f) {
#ifndef AKMI_DEV_FAST
  if (rs == AKMI_RS_LLF) return f(IC<0>{});
  if (rs == AKMI_RS_HLLE) return f(IC<1>{});
  if constexpr (ADV) {
    if (rs == AKMI_RS_ADVECT) return f(IC<5>{});
  }
#endif
  if constexpr (MHD) {
    if (rs == AKMI_RS_HLLD) return f(IC<3>{});
    set_error("<mhd> rsolver = %d not implemented (llf, hlle, hlld)", rs);
  } else {
    if (rs == AKMI_RS_HLLC) return f(IC<2>{});
    if (rs == AKMI_RS_ROE) return f(IC<4>{});
    set_error("<hydro> rsolver = %d not implemented (llf, hlle, hllc, roe)", rs);
  }
  return AKMI_FAIL;
}

// f(IC<RECON>, IC<RS>) for the run-time pair
template <bool MHD, bool ADV = false, class F>
inline int dispatch_scheme(const Scheme &sc, F &&f) {
  return dispatch_recon(sc.recon, [&](auto R) {
    return dispatch_rsolver<MHD, ADV>(sc.rsolver, [&](auto S) { return f(R, S); });
  });
}

// the fused stage kernels: the isothermal solvers are RS + 10 (rs_iso<RS>(), akmi_numerics.hpp)
template <bool MHD, class F>
inline int dispatch_scheme_eos(const Scheme &sc, F &&f) {
  if (!sc.iso) return dispatch_scheme<MHD>(sc, f);
#ifdef AKMI_DEV_FAST
  set_error("developer build: ideal gas only");
  return AKMI_FAIL;
#endif
  return dispatch_recon(sc.recon, [&](auto R) {
    const int rs = sc.rsolver;
    if (rs == AKMI_RS_LLF) return f(R, IC<10>{});
    if (rs == AKMI_RS_HLLE) return f(R, IC<11>{});
    if constexpr (MHD) {
      if (rs == AKMI_RS_HLLD) return f(R, IC<13>{});
      set_error("<mhd> rsolver = %d not implemented for the isothermal EOS (llf, hlle, hlld)", rs);
    } else {
      if (rs == AKMI_RS_ROE) return f(R, IC<14>{});
      set_error("<hydro> rsolver = %d not implemented for the isothermal EOS (llf, hlle, roe)", rs);
    }
    return (int)AKMI_FAIL;
  });
}

inline int check_scheme(const akmi_pack *p, int recon, const char *who) {
  if (p->nvar < (p->is_ideal ? 5 : 4)) {
    set_error("%s: nvar = %d is smaller than the fluid variable set of the EOS (5 ideal gas, 4 "
              "isothermal)", who, p->nvar);
    return AKMI_FAIL;
  }
  if (recon >= AKMI_RECON_PPM4 && recon <= AKMI_RECON_TENO && p->ng < 3) {
    // hydro.cpp:173-179: the +/-2 stencil requires at least 3 ghost zones
    set_error("%s: ppm4/ppmx/wenoz/teno need nghost>=3", who);
    return AKMI_FAIL;
  }
  return AKMI_COMPLETE;
}

// wave-uniform base (scalar registers) + 32-bit BYTE offset of the lane: global_load v, v_off, s[base].
// One variable of one MeshBlock has to stay below 4 GB (checked by the launchers that use it).
__device__ __forceinline__ double ldu(const double *__restrict__ base, unsigned ob) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + ob);
}
__device__ __forceinline__ void stu(double *__restrict__ base, unsigned ob, double v) {
  *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + ob) = v;
}

// Lane mapping of the cell / face kernels that have one thread per element: the threads of a launch run over the box
// [kl, kl+nk) x [jl, jl+nj) x columns of MeshBlock m, i fastest, in one of these forms (chosen per kernel and pack at launch,
// measured on 120 blocks of 16^3 and 960 blocks of 32^3 with four ghost cells: profiles/r06_lane_mapping.txt):
//   0                       rows of n1 elements (whole rows of the array: contiguous addresses across a wave), one group of
//                           workgroups per plane: blockIdx.z = m*nk + plane
//   FLAT_K                  the planes flattened too, blockIdx.z = m: a plane of a small MeshBlock is 1.1 - 2.3 workgroups, and
//                           every plane rounds that up
//   FLAT_K | FLAT_COLS      ... and only the columns [il, iu] of a row: the 16 active cells of a row of 24
// `in` is false for the threads outside the box.
enum { FLAT_K = 1, FLAT_COLS = 2 };
struct Cell3 { int i, j, k, m; bool in; };
__device__ __forceinline__ Cell3 flat_cells(int mode, int n1, int il, int iu, int jl, int nj, int kl, int nk) {
  const unsigned p = (blockIdx.x*blockDim.y + threadIdx.y)*blockDim.x + threadIdx.x;
  const unsigned c0 = (mode & FLAT_COLS) ? (unsigned)il : 0u, ni = (mode & FLAT_COLS) ? (unsigned)(iu - il + 1) : (unsigned)n1;
  const unsigned plane = ni*(unsigned)nj;
  unsigned kk, pr, m;
  if (mode & FLAT_K) { kk = p/plane; pr = p - kk*plane; m = blockIdx.z; }
  else { m = blockIdx.z/(unsigned)nk; kk = blockIdx.z - m*(unsigned)nk; pr = p; }
  const unsigned jj = pr/ni;
  const int i = (int)(c0 + pr - jj*ni);
  return Cell3{i, jl + (int)jj, kl + (int)kk, (int)m, (int)kk < nk && (int)jj < nj && i >= il && i <= iu};
}
inline dim3 flat_cells_grid(int mode, int n1, int il, int iu, long nj, long nk, int nmb, int threads = 256) {
  const long ni = (mode & FLAT_COLS) ? iu - il + 1 : n1;
  if (mode & FLAT_K) return dim3((unsigned)((ni*nj*nk + threads - 1)/threads), 1, (unsigned)nmb);
  return dim3((unsigned)((ni*nj + threads - 1)/threads), 1, (unsigned)(nmb*nk));
}

// gfx950 runs wave64 only; the kernels that count on it (ballot words, lane shifts, four waves per 256 threads) say so
constexpr int AKMI_WAVE = 64;
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE)
static_assert(__AMDGCN_WAVEFRONT_SIZE == AKMI_WAVE, "libakmi is written for 64-lane wavefronts (gfx950)");
#endif

// The value of the neighbouring lane of the wave (lane - 1 / lane + 1) as a DPP move of the two halves: two VALU
// instructions per double, no trip through the LDS crossbar (ds_bpermute_b32 x 2 + address arithmetic + lgkmcnt wait is
// what __shfl_up / __shfl_down(x, 1, 64) compile to).  Lane 0 / lane 63 keep their own value, as with __shfl_up / _down.
// Every lane of the wave has to be active at the call.
__device__ __forceinline__ double lane_below(double x) {          // == __shfl_up(x, 1, 64)
  const long long v = __double_as_longlong(x);
  int lo = (int)v, hi = (int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);     // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_above(double x) {          // == __shfl_down(x, 1, 64)
  const long long v = __double_as_longlong(x);
  int lo = (int)v, hi = (int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);     // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// a wave-uniform double the vector unit computed (e.g. 1/dx), moved to scalar registers: frees two VGPRs per value
__device__ __forceinline__ double to_sgpr(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// true when x is a normal power of two, i.e. 1/x is exact (mantissa bits all zero)
__host__ __device__ inline bool is_pow2(double x) {
  unsigned long long b;
  memcpy(&b, &x, sizeof b);
  const unsigned e = (unsigned)((b >> 52) & 0x7ff);
  return (b & 0xfffffffffffffull) == 0 && e > 1 && e < 2046 && (b >> 63) == 0;
}

// for a power of two dx: x/dx == ldexp(x, pow2_shift(dx)) bit for bit (scaling by 2^n is exact up to the one
// correct rounding into the subnormal range that the division performs as well); the shift is formed on the
// scalar unit from the bits of the (wave-uniform) cell size: no reciprocal to keep in vector registers
__device__ __forceinline__ int pow2_shift(double dx) {
  return 1023 - (int)(((unsigned long long)__double_as_longlong(dx) >> 52) & 0x7ff);
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

// akmi_stage.hip: the flux kernels of the task-granular entry points by the sweeps of the fused stage
// (bcc0 == nullptr: hydro).  AKMI_COMPLETE / AKMI_FAIL, or -1 when the pack is outside what the sweeps
// address (the caller then uses its own kernels).
int sweeps_store_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0, const double *bcc0,
                        const double *bx1f, const double *bx2f, const double *bx3f, double *flx1, double *flx2,
                        double *flx3, int face_shaped, double *e3x1, double *e2x1, double *e1x2, double *e3x2,
                        double *e2x3, double *e1x3, void *stream);

}  // namespace akmi
#endif
