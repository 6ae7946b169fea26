// akmi_tasks.hip -- one HIP kernel (family) per reference task body, behind the C ABI of
// include/akmi.h.  These are the task-granular entry points (drop-in for the bodies of
// Hydro::Fluxes, RKUpdate, ConToPrim, NewTimeStep, MHD::Fluxes, CornerE, CT ...); the fused
// per-stage fast path lives in akmi_stage.hip and must reproduce these bit for bit.
//
// Thread mapping everywhere: threadIdx.x runs along i (the contiguous index of the
// (m,n,k,j,i) LayoutRight arrays) so every global access of a wave is a 512-B coalesced
// row segment; blockDim = (64,4): one wave64 per j-row, 4 rows per workgroup.
#include <cstdarg>
#include "akmi_common.hpp"

namespace akmi {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int BX = 64, BY = 4;

// (lane mapping of the cell kernels below: flat_cells, akmi_common.hpp)

// ---------------------------------------------------------------------------------------
// Hydro fluxes: reconstruct in registers + Riemann solver RS.  hydro_fluxes.cpp:77-229.
template <int DIR, int RECON, int RS, bool ISO>
__global__ void __launch_bounds__(BX*BY)
k_hydro_flux(Geo g, FaceEos eos, const double *__restrict__ w0, double *__restrict__ flx,
             int f3, int f2, int f1, int il, int iu, int jl, int ju, int kl, int nk) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  constexpr int ivx = 1 + DIR, ivy = 1 + (DIR + 1)%3, ivz = 1 + (DIR + 2)%3;
  const long s = (DIR == 0) ? 1 : (DIR == 1 ? (long)g.N1 : (long)g.N1*g.N2);
  const size_t cs = (size_t)g.N3*g.N2*g.N1;   // variable stride
  const double *q = w0 + ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  double ld, lx, ly, lz, le, rd, rx, ry, rz, re;
  face_states<RECON, 1>(q + 0*cs, s, eos, ld, rd);
  face_states<RECON, 0>(q + ivx*cs, s, eos, lx, rx);
  face_states<RECON, 0>(q + ivy*cs, s, eos, ly, ry);
  face_states<RECON, 0>(q + ivz*cs, s, eos, lz, rz);
  double fd, fx, fy, fz, fe = 0.0;
  if constexpr (ISO) {                      // no energy variable: g.nvar == 4
    le = re = 0.0;
    riemann_hyd_iso<RS>(eos.iso_cs, ld, lx, ly, lz, rd, rx, ry, rz, fd, fx, fy, fz);
  } else {
    face_states<RECON, 2>(q + 4*cs, s, eos, le, re);
    riemann_hyd<RS>(eos.gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, fd, fx, fy, fz, fe);
  }
  const size_t fs = (size_t)f3*f2*f1;
  double *f = flx + ix5(g.nvar, f3, f2, f1, m, 0, k, j, i);
  f[0] = fd; f[ivx*fs] = fx; f[ivy*fs] = fy; f[ivz*fs] = fz;
  if constexpr (!ISO) f[4*fs] = fe;
  // passive scalars: reconstructed like any primitive, upwinded by the sign of the mass flux
  // (hydro_fluxes.cpp:135-147)
  constexpr int NF = ISO ? 4 : 5;
  for (int n = NF; n < g.nvar; ++n) {
    double sl, sr;
    face_states<RECON, 0>(q + n*cs, s, eos, sl, sr);
    f[n*fs] = fd*((fd >= 0.0) ? sl : sr);
  }
}

template <int DIR>
static int launch_hydro_flux(const Geo &g, const Scheme &sc, const double *w0,
                             double *flx, int fsh, hipStream_t st, int ext = 0) {
  int il = g.is, iu = g.ie, jl = g.js, ju = g.je, kl = g.ks, ku = g.ke;
  if (ext) {                      // <hydro>/fofc: hydro_fluxes.cpp:92-101
    il = g.is - 1; iu = g.ie + 1;
    if (g.multi_d) { jl = g.js - 1; ju = g.je + 1; }
    if (g.three_d) { kl = g.ks - 1; ku = g.ke + 1; }
  }
  int f3 = g.N3, f2 = g.N2, f1 = g.N1;
  if (DIR == 0) { il = g.is - ext; iu = g.ie + 1 + ext; f1 += fsh; }
  if (DIR == 1) { jl = g.js - ext; ju = g.je + 1 + ext; f2 += fsh; }
  if (DIR == 2) { kl = g.ks - ext; ku = g.ke + 1 + ext; f3 += fsh; }
  int nk = ku - kl + 1;
  dim3 grid(cdiv(iu - il + 1, BX), cdiv(ju - jl + 1, BY), nk*g.nmb), block(BX, BY);
  int rc = dispatch_scheme<false, true>(sc, [&](auto R, auto S) {
    if (sc.iso)
      k_hydro_flux<DIR, decltype(R)::value, decltype(S)::value, true><<<grid, block, 0, st>>>(
          g, sc.eos, w0, flx, f3, f2, f1, il, iu, jl, ju, kl, nk);
    else
      k_hydro_flux<DIR, decltype(R)::value, decltype(S)::value, false><<<grid, block, 0, st>>>(
          g, sc.eos, w0, flx, f3, f2, f1, il, iu, jl, ju, kl, nk);
    return AKMI_COMPLETE;
  });
  if (rc != AKMI_COMPLETE) return rc;
  AKMI_CHECK_LAUNCH("hydro_flux");
  return AKMI_COMPLETE;
}

// ---------------------------------------------------------------------------------------
// RKUpdate: hydro_update.cpp:50-81
__global__ void __launch_bounds__(BX*BY)
k_rk_update(Geo g, double gam0, double gam1, double beta_dt, double *__restrict__ u0,
            const double *__restrict__ u1, const double *__restrict__ flx1,
            const double *__restrict__ flx2, const double *__restrict__ flx3, int fsh, int fm) {
  const Cell3 q = flat_cells(fm, g.N1, g.is, g.ie, g.js, g.nx2, g.ks, g.nx3);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  if (!q.in) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  // cell sizes that are powers of two (every level of a refined mesh on a 2^n root grid): x/dx as one v_ldexp_f64,
  // bit for bit the quotient (akmi_common.hpp pow2_shift; tests/test_gpu_fastmath.py)
  const bool p2 = is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
  for (int n = 0; n < g.nvar; ++n) {
    const double d1 = flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i + 1)] -
                      flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i)];
    double divf = p2 ? ldexp(d1, n1) : d1/dx1;
    if (g.multi_d) {
      const double d2 = flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j + 1, i)] -
                        flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j, i)];
      divf += p2 ? ldexp(d2, n2) : d2/dx2;
    }
    if (g.three_d) {
      const double d3 = flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k + 1, j, i)] -
                        flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k, j, i)];
      divf += p2 ? ldexp(d3, n3) : d3/dx3;
    }
    size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, n, k, j, i);
    u0[c] = gam0*u0[c] + gam1*u1[c] - beta_dt*divf;
  }
}

// First stage OUT OF PLACE (task-granular path): CopyCons (u1 := u0, hydro_tasks.cpp:130-152) followed by RKUpdate
// leaves u1 = old state, u0 = gam0*old + gam1*old - beta_dt*divF in the active cells and the old state in the
// ghost zones.  This kernel writes exactly that u0 into `dst` (every cell: active cells updated with the same
// expression on the same operands -- u1 == u0 after CopyCons --, ghost cells copied) and leaves `src` alone; the caller
// swaps the two registers.  One pass over the arrays instead of copy + update.
__global__ void __launch_bounds__(BX*BY)
k_rk_update_oop(Geo g, double gam0, double gam1, double beta_dt, const double *__restrict__ src,
                double *__restrict__ dst, const double *__restrict__ flx1, const double *__restrict__ flx2,
                const double *__restrict__ flx3, int fsh) {
  const Cell3 q = flat_cells(FLAT_K, g.N1, 0, g.N1 - 1, 0, g.N2, 0, g.N3);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  if (!q.in) return;
  const bool act = i >= g.is && i <= g.ie && j >= g.js && j <= g.je && k >= g.ks && k <= g.ke;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const bool p2 = is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
  for (int n = 0; n < g.nvar; ++n) {
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, n, k, j, i);
    const double old = src[c];
    if (!act) { dst[c] = old; continue; }
    const double d1 = flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i + 1)] -
                      flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i)];
    double divf = p2 ? ldexp(d1, n1) : d1/dx1;
    if (g.multi_d) {
      const double d2 = flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j + 1, i)] -
                        flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j, i)];
      divf += p2 ? ldexp(d2, n2) : d2/dx2;
    }
    if (g.three_d) {
      const double d3 = flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k + 1, j, i)] -
                        flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k, j, i)];
      divf += p2 ? ldexp(d3, n3) : d3/dx3;
    }
    dst[c] = gam0*old + gam1*old - beta_dt*divf;
  }
}

// ---------------------------------------------------------------------------------------
// ConsToPrim: ideal_hyd.cpp:45-103, ideal_mhd.cpp:47-122.  Floor counters: one wave-level
// ballot + one atomicAdd per wave that actually hit a floor (never on the hot path).
template <bool MHD>
__global__ void __launch_bounds__(BX*BY)
k_c2p(Geo g, Eos eos, double *__restrict__ u0, const double *__restrict__ bx1f,
      const double *__restrict__ bx2f, const double *__restrict__ bx3f,
      double *__restrict__ w0, double *__restrict__ bcc0, int il, int iu, int jl, int ju,
      int kl, int nk, int *__restrict__ counters) {
  const Cell3 q = flat_cells(0, g.N1, il, iu, jl, ju - jl + 1, kl, nk);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  if (!q.in) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  double ubx = 0.0, uby = 0.0, ubz = 0.0;
  if constexpr (MHD) {
    ubx = 0.5*(bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i)] +
               bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i + 1)]);
    uby = 0.5*(bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i)] +
               bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j + 1, i)]);
    ubz = 0.5*(bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i)] +
               bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k + 1, j, i)]);
    const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
    bcc0[b] = ubx; bcc0[b + cs] = uby; bcc0[b + 2*cs] = ubz;
  }
  if (!eos.is_ideal) {
    // SingleC2P_IsothermalHyd / _IsothermalMHD (isothermal_hyd.cpp:30-45, isothermal_mhd.cpp:32-47,
    // density floor fmax(dfloor, b^2/sigma_max) :104-106): no energy variable (g.nvar == 4)
    double ud = u0[c];
    double dfloor_ = eos.dfloor;
    if constexpr (MHD) dfloor_ = fmax(eos.dfloor, (sqr(ubx) + sqr(uby) + sqr(ubz))/eos.sigma_max);
    if (ud < dfloor_) { ud = dfloor_; u0[c] = ud; atomicAdd(&counters[0], 1); }
    const double di = 1.0/ud;
    w0[c] = ud; w0[c + cs] = di*u0[c + cs]; w0[c + 2*cs] = di*u0[c + 2*cs];
    w0[c + 3*cs] = di*u0[c + 3*cs];
    for (int n = 4; n < g.nvar; ++n) w0[c + n*cs] = u0[c + n*cs]/ud;   // isothermal_*.cpp: no floor
    return;
  }
  double ud = u0[c], umx = u0[c + cs], umy = u0[c + 2*cs], umz = u0[c + 3*cs], ue = u0[c + 4*cs];
  double wd, wvx, wvy, wvz, we;
  bool dfl = false, efl = false, tfl = false;
  if constexpr (MHD) {
    c2p_mhd(eos, ud, umx, umy, umz, ue, ubx, uby, ubz, wd, wvx, wvy, wvz, we, dfl, efl, tfl);
  } else {
    c2p_hyd(eos, ud, umx, umy, umz, ue, wd, wvx, wvy, wvz, we, dfl, efl, tfl);
  }
  if (dfl) { u0[c] = ud; atomicAdd(&counters[0], 1); }
  if (efl) { u0[c + 4*cs] = ue; atomicAdd(&counters[1], 1); }
  if (tfl) { u0[c + 4*cs] = ue; atomicAdd(&counters[2], 1); }
  w0[c] = wd; w0[c + cs] = wvx; w0[c + 2*cs] = wvy; w0[c + 3*cs] = wvz; w0[c + 4*cs] = we;
  for (int n = 5; n < g.nvar; ++n) {        // scalars with their floor, ideal_hyd.cpp:94-101
    double us = u0[c + n*cs];
    if (us < 0.0) { us = 0.0; u0[c + n*cs] = 0.0; }
    w0[c + n*cs] = us/ud;
  }
}

// ---------------------------------------------------------------------------------------
// NewTimeStep: hydro_newdt.cpp:73-119, mhd_newdt.cpp:76-150.  Three minima; positive
// doubles order like their bit patterns, so the cross-workgroup reduction is one
// 64-bit atomicMin per workgroup per direction after a wave-level DPP/shuffle min.
__device__ __forceinline__ double wave_min(double v) {
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ void block_min3_atomic(double a, double b, double c,
                                                  double *__restrict__ dt3) {
  __shared__ double sm[3][BY];
  a = wave_min(a); b = wave_min(b); c = wave_min(c);
  const int lane = threadIdx.x & 63, w = threadIdx.y;
  if (lane == 0) { sm[0][w] = a; sm[1][w] = b; sm[2][w] = c; }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < 3) {
    double v = sm[threadIdx.x][0];
    for (int q = 1; q < BY; ++q) v = fmin(v, sm[threadIdx.x][q]);
    if (v < __hip_atomic_load(&dt3[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(reinterpret_cast<unsigned long long *>(&dt3[threadIdx.x]),
                (unsigned long long)__double_as_longlong(v));
  }
}

template <bool MHD>
__device__ __forceinline__ void cell_dt(const Geo &g, const Eos &eos, const double *__restrict__ w0,
                                        const double *__restrict__ bcc0, int m, int k, int j,
                                        int i, double &d1, double &d2, double &d3) {
  const double gamma = eos.gamma;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  double wd = w0[c], vx = w0[c + cs], vy = w0[c + 2*cs], vz = w0[c + 3*cs];
  double mv1, mv2, mv3;
  if (!eos.is_ideal) {                     // hydro_newdt.cpp:109-111, mhd_newdt.cpp:137-144
    if constexpr (MHD) {
      const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
      double bx = bcc0[b], by = bcc0[b + cs], bz = bcc0[b + 2*cs];
      mv1 = fabs(vx) + fast_speed_iso(eos.iso_cs, wd, bx, by, bz);
      mv2 = fabs(vy) + fast_speed_iso(eos.iso_cs, wd, by, bz, bx);
      mv3 = fabs(vz) + fast_speed_iso(eos.iso_cs, wd, bz, bx, by);
    } else {
      mv1 = fabs(vx) + eos.iso_cs; mv2 = fabs(vy) + eos.iso_cs; mv3 = fabs(vz) + eos.iso_cs;
    }
    d1 = g.dx[3*m]/mv1; d2 = g.dx[3*m + 1]/mv2; d3 = g.dx[3*m + 2]/mv3;
    return;
  }
  double pr = (gamma - 1.0)*w0[c + 4*cs];
  if constexpr (MHD) {
    const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
    double bx = bcc0[b], by = bcc0[b + cs], bz = bcc0[b + 2*cs];
    mv1 = fabs(vx) + fast_speed(gamma, wd, pr, bx, by, bz);
    mv2 = fabs(vy) + fast_speed(gamma, wd, pr, by, bz, bx);
    mv3 = fabs(vz) + fast_speed(gamma, wd, pr, bz, bx, by);
  } else {
    double cs_ = sqrt(gamma*pr/wd);
    mv1 = fabs(vx) + cs_; mv2 = fabs(vy) + cs_; mv3 = fabs(vz) + cs_;
  }
  d1 = g.dx[3*m]/mv1; d2 = g.dx[3*m + 1]/mv2; d3 = g.dx[3*m + 2]/mv3;
}

template <bool MHD>
__global__ void __launch_bounds__(BX*BY)
k_newdt(Geo g, Eos eos, const double *__restrict__ w0, const double *__restrict__ bcc0,
        double *__restrict__ dt3) {
  const Cell3 q = flat_cells(0, g.N1, g.is, g.ie, g.js, g.nx2, g.ks, g.nx3);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  double d1 = (double)FLT_MAX, d2 = (double)FLT_MAX, d3 = (double)FLT_MAX;
  if (q.in) cell_dt<MHD>(g, eos, w0, bcc0, m, k, j, i, d1, d2, d3);
  block_min3_atomic(d1, d2, d3, dt3);
}

__global__ void k_init_dt(double *dt3) {
  if (threadIdx.x < 3) dt3[threadIdx.x] = (double)FLT_MAX;
}

// ---------------------------------------------------------------------------------------
// History sums (src/outputs/history.cpp:78-160, 272-374): volume-weighted sums over the active
// cells of the conserved variables, the three kinetic energies and (MHD) the three magnetic
// energies.  One cell per thread, wave reduction by shuffles, workgroup reduction through LDS,
// one fp64 atomicAdd per workgroup and quantity (the reference's Kokkos reduction is not
// order-deterministic either; values agree to round-off).
AKMI_DEV double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool MHD>
__global__ void __launch_bounds__(BX*BY)
k_history(Geo g, int ideal, const double *__restrict__ u0, const double *__restrict__ bx1f,
          const double *__restrict__ bx2f, const double *__restrict__ bx3f,
          double *__restrict__ out) {
  constexpr int NH = MHD ? 11 : 8;
  __shared__ double sm[NH][BY];
  const int i = g.is + blockIdx.x*BX + threadIdx.x;
  const int j = g.js + blockIdx.y*BY + threadIdx.y;
  const int nk = g.ke - g.ks + 1;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  double h[NH];
#pragma unroll
  for (int n = 0; n < NH; ++n) h[n] = 0.0;
  if (i <= g.ie && j <= g.je) {
    const double vol = g.dx[3*m]*g.dx[3*m + 1]*g.dx[3*m + 2];
    const size_t cs = (size_t)g.N3*g.N2*g.N1;
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
    const double d = u0[c], m1 = u0[c + cs], m2 = u0[c + 2*cs], m3 = u0[c + 3*cs];
    // history.cpp: the kinetic energies start at slot nhydro|nmhd (4 without an energy variable)
    const int o = ideal ? 5 : 4;
    h[0] = vol*d; h[1] = vol*m1; h[2] = vol*m2; h[3] = vol*m3;
    if (ideal) h[4] = vol*u0[c + 4*cs];
    h[o] = vol*0.5*sqr(m1)/d;
    h[o + 1] = vol*0.5*sqr(m2)/d;
    h[o + 2] = vol*0.5*sqr(m3)/d;
    if constexpr (MHD) {
      h[o + 3] = vol*0.25*(sqr(bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i + 1)]) +
                       sqr(bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i)]));
      h[o + 4] = vol*0.25*(sqr(bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j + 1, i)]) +
                       sqr(bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i)]));
      h[o + 5] = vol*0.25*(sqr(bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k + 1, j, i)]) +
                        sqr(bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i)]));
    }
  }
#pragma unroll
  for (int n = 0; n < NH; ++n) {
    const double v = wave_sum(h[n]);
    if (threadIdx.x == 0) sm[n][threadIdx.y] = v;
  }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < NH) {
    double v = sm[threadIdx.x][0];
    for (int q = 1; q < BY; ++q) v += sm[threadIdx.x][q];
    atomicAdd(&out[threadIdx.x], v);
  }
}

// ---------------------------------------------------------------------------------------
// MHD fluxes: reconstruct w0 and bcc0 in registers + Riemann solver RS.  mhd_fluxes.cpp:84-266.
template <int DIR, int RECON, int RS, bool ISO>
__global__ void __launch_bounds__(BX*BY)
k_mhd_flux(Geo g, FaceEos eos, const double *__restrict__ w0, const double *__restrict__ bcc0,
           const double *__restrict__ bxf, double *__restrict__ flx, double *__restrict__ ey,
           double *__restrict__ ez, int f3, int f2, int f1, int il, int iu, int jl, int ju,
           int kl, int nk) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  constexpr int ivx = 1 + DIR, ivy = 1 + (DIR + 1)%3, ivz = 1 + (DIR + 2)%3;
  constexpr int iby = (DIR + 1)%3, ibz = (DIR + 2)%3;
  const long s = (DIR == 0) ? 1 : (DIR == 1 ? (long)g.N1 : (long)g.N1*g.N2);
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const double *q = w0 + ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  const double *b = bcc0 + ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
  double ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz;
  face_states<RECON, 1>(q + 0*cs, s, eos, ld, rd);
  face_states<RECON, 0>(q + ivx*cs, s, eos, lx, rx);
  face_states<RECON, 0>(q + ivy*cs, s, eos, ly, ry);
  face_states<RECON, 0>(q + ivz*cs, s, eos, lz, rz);
  face_states<RECON, 0>(b + iby*cs, s, eos, lby, rby);
  face_states<RECON, 0>(b + ibz*cs, s, eos, lbz, rbz);
  const double bxi = bxf[ix4(f3, f2, f1, m, k, j, i)];
  Cons1D fl;
  if constexpr (ISO) {
    le = re = 0.0;
    fl = riemann_mhd_iso<RS>(eos, ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  } else {
    face_states<RECON, 2>(q + 4*cs, s, eos, le, re);
    fl = riemann_mhd<RS, true>(eos.gamma, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
  }
  const size_t fs = (size_t)f3*f2*f1;
  double *f = flx + ix5(g.nvar, f3, f2, f1, m, 0, k, j, i);
  f[0] = fl.d; f[ivx*fs] = fl.mx; f[ivy*fs] = fl.my; f[ivz*fs] = fl.mz;
  if constexpr (!ISO && RS != 5) f[4*fs] = fl.e;      // advect leaves the energy flux untouched
  // passive scalars, active transverse range only (mhd_fluxes.cpp:153-166)
  constexpr int NF = ISO ? 4 : 5;
  if (g.nvar > NF && i >= g.is && i <= g.ie + (DIR == 0) && j >= g.js && j <= g.je + (DIR == 1) &&
      k >= g.ks && k <= g.ke + (DIR == 2)) {
    for (int n = NF; n < g.nvar; ++n) {
      double sl, sr;
      face_states<RECON, 0>(q + n*cs, s, eos, sl, sr);
      f[n*fs] = fl.d*((fl.d >= 0.0) ? sl : sr);
    }
  }
  const size_t ec = ix4(g.N3, g.N2, g.N1, m, k, j, i);
  ey[ec] = -fl.by;
  ez[ec] = fl.bz;
}

template <int DIR>
static int launch_mhd_flux(const Geo &g, const Scheme &sc, const double *w0,
                           const double *bcc0, const double *bxf, double *flx, double *ey,
                           double *ez, hipStream_t st, int ext = 0) {
  // ext = 1: <mhd>/fofc, face-normal range one face wider on both sides (mhd_fluxes.cpp:100-105)
  int il, iu, jl, ju, kl, ku;
  int f3 = g.N3, f2 = g.N2, f1 = g.N1;
  if (DIR == 0) {
    il = g.is - ext; iu = g.ie + 1 + ext; jl = g.js; ju = g.je; kl = g.ks; ku = g.ke;
    if (g.multi_d) { jl = g.js - 1; ju = g.je + 1; }
    if (g.three_d) { kl = g.ks - 1; ku = g.ke + 1; }
    f1 += 1;
  } else if (DIR == 1) {
    il = g.is - 1; iu = g.ie + 1; jl = g.js - ext; ju = g.je + 1 + ext; kl = g.ks; ku = g.ke;
    if (g.three_d) { kl = g.ks - 1; ku = g.ke + 1; }
    f2 += 1;
  } else {
    il = g.is - 1; iu = g.ie + 1; jl = g.js - 1; ju = g.je + 1; kl = g.ks - ext; ku = g.ke + 1 + ext;
    f3 += 1;
  }
  int nk = ku - kl + 1;
  dim3 grid(cdiv(iu - il + 1, BX), cdiv(ju - jl + 1, BY), nk*g.nmb), block(BX, BY);
  int rc = dispatch_scheme<true, true>(sc, [&](auto R, auto S) {
    if (sc.iso)
      k_mhd_flux<DIR, decltype(R)::value, decltype(S)::value, true><<<grid, block, 0, st>>>(
          g, sc.eos, w0, bcc0, bxf, flx, ey, ez, f3, f2, f1, il, iu, jl, ju, kl, nk);
    else
      k_mhd_flux<DIR, decltype(R)::value, decltype(S)::value, false><<<grid, block, 0, st>>>(
          g, sc.eos, w0, bcc0, bxf, flx, ey, ez, f3, f2, f1, il, iu, jl, ju, kl, nk);
    return AKMI_COMPLETE;
  });
  if (rc != AKMI_COMPLETE) return rc;
  AKMI_CHECK_LAUNCH("mhd_flux");
  return AKMI_COMPLETE;
}

// ---------------------------------------------------------------------------------------
// CornerE, mhd_corner_e.cpp:26-417.  Cell-centred E = -(v x B) (:309-317) is recomputed
// in registers from w0/bcc0 instead of being staged through e1_cc/e2_cc/e3_cc arrays.
struct EccAccess {
  const double *w0, *bcc0;
  int nvar, N3, N2, N1;
  size_t cs;
  __device__ __forceinline__ size_t cw(int m, int k, int j, int i) const {
    return ix5(nvar, N3, N2, N1, m, 0, k, j, i);
  }
  __device__ __forceinline__ size_t cb(int m, int k, int j, int i) const {
    return ix5(3, N3, N2, N1, m, 0, k, j, i);
  }
  __device__ __forceinline__ double e1(int m, int k, int j, int i) const {
    size_t a = cw(m, k, j, i), b = cb(m, k, j, i);
    return w0[a + 3*cs]*bcc0[b + cs] - w0[a + 2*cs]*bcc0[b + 2*cs];
  }
  __device__ __forceinline__ double e2(int m, int k, int j, int i) const {
    size_t a = cw(m, k, j, i), b = cb(m, k, j, i);
    return w0[a + cs]*bcc0[b + 2*cs] - w0[a + 3*cs]*bcc0[b];
  }
  __device__ __forceinline__ double e3(int m, int k, int j, int i) const {
    size_t a = cw(m, k, j, i), b = cb(m, k, j, i);
    return w0[a + 2*cs]*bcc0[b] - w0[a + cs]*bcc0[b + cs];
  }
};

__global__ void k_corner_e_1d(Geo g, const double *__restrict__ e3x1,
                              const double *__restrict__ e2x1, double *__restrict__ e2,
                              double *__restrict__ e3) {
  const int i = g.is + blockIdx.x*blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (i > g.ie + 1) return;
  const int ks = g.ks, ke = g.ke, js = g.js, je = g.je;
  double a2 = e2x1[ix4(g.N3, g.N2, g.N1, m, ks, js, i)];
  double a3 = e3x1[ix4(g.N3, g.N2, g.N1, m, ks, js, i)];
  e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, ks, js, i)] = a2;
  e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, ke + 1, js, i)] = a2;
  e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, ks, js, i)] = a3;
  e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, ks, je + 1, i)] = a3;
}

#define CCE(a, m, k, j, i) a[ix4(g.N3, g.N2, g.N1, m, k, j, i)]
#define F1D(m, k, j, i) flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, 0, k, j, i)]
#define F2D(m, k, j, i) flx2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, 0, k, j, i)]
#define F3D(m, k, j, i) flx3[ix5(g.nvar, g.N3 + 1, g.N2, g.N1, m, 0, k, j, i)]

__device__ __forceinline__ double corner_e3(const Geo &g, const EccAccess &cc,
    const double *__restrict__ e3x1, const double *__restrict__ e3x2,
    const double *__restrict__ flx1, const double *__restrict__ flx2, int m, int k, int j,
    int i) {
  double e3_l2, e3_r2, e3_l1, e3_r1;
  if (F1D(m, k, j - 1, i) >= 0.0) e3_l2 = CCE(e3x2, m, k, j, i - 1) - cc.e3(m, k, j - 1, i - 1);
  else                            e3_l2 = CCE(e3x2, m, k, j, i) - cc.e3(m, k, j - 1, i);
  if (F1D(m, k, j, i) >= 0.0)     e3_r2 = CCE(e3x2, m, k, j, i - 1) - cc.e3(m, k, j, i - 1);
  else                            e3_r2 = CCE(e3x2, m, k, j, i) - cc.e3(m, k, j, i);
  if (F2D(m, k, j, i - 1) >= 0.0) e3_l1 = CCE(e3x1, m, k, j - 1, i) - cc.e3(m, k, j - 1, i - 1);
  else                            e3_l1 = CCE(e3x1, m, k, j, i) - cc.e3(m, k, j, i - 1);
  if (F2D(m, k, j, i) >= 0.0)     e3_r1 = CCE(e3x1, m, k, j - 1, i) - cc.e3(m, k, j - 1, i);
  else                            e3_r1 = CCE(e3x1, m, k, j, i) - cc.e3(m, k, j, i);
  return 0.25*(e3_l1 + e3_r1 + e3_l2 + e3_r2 + CCE(e3x2, m, k, j, i - 1) + CCE(e3x2, m, k, j, i) +
               CCE(e3x1, m, k, j - 1, i) + CCE(e3x1, m, k, j, i));
}

__global__ void __launch_bounds__(BX*BY)
k_corner_e_2d(Geo g, EccAccess cc, const double *__restrict__ e3x1,
              const double *__restrict__ e2x1, const double *__restrict__ e1x2,
              const double *__restrict__ e3x2, const double *__restrict__ flx1,
              const double *__restrict__ flx2, double *__restrict__ e1, double *__restrict__ e2,
              double *__restrict__ e3) {
  const int i = g.is + blockIdx.x*BX + threadIdx.x;
  const int j = g.js + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z;
  if (i > g.ie + 1 || j > g.je + 1) return;
  const int ks = g.ks, ke = g.ke;
  e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, ks, j, i)] = CCE(e2x1, m, ks, j, i);
  e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, ke + 1, j, i)] = CCE(e2x1, m, ks, j, i);
  e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, ks, j, i)] = CCE(e1x2, m, ks, j, i);
  e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, ke + 1, j, i)] = CCE(e1x2, m, ks, j, i);
  e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, ks, j, i)] =
      corner_e3(g, cc, e3x1, e3x2, flx1, flx2, m, ks, j, i);
}


// 3-D CornerE as a march along k (the form the fused stage uses, without its LDS tile and CT): a lane owns
// the corner (j,i) of a flattened row, computes the cell-centred E = -(v x B) of ITS cell (k,j,i) and of
// the cell below in j once per plane, takes the i-1 neighbours from the lane below (__shfl_up; waves
// overlap by one lane, lane 0 only provides) and keeps the k-1 operands of the corner formulas in
// registers from step to step.  25 loads per corner instead of ~70, upwinding by select instead of
// branches, scalar-base addressing.  Same operands, same operations -> same bits as k_corner_e_3d.
constexpr int CX = 64, CY = 4;
__device__ __forceinline__ double upw_sel(bool pos, double fa, double ca, double fb, double cb) {
  const double f = pos ? fa : fb, c = pos ? ca : cb;        // all four operands are loaded; select, no branch
  return f - c;
}
__global__ void __launch_bounds__(CX*CY)
k_corner_e_3d_march(Geo g, const double *__restrict__ w0, const double *__restrict__ bcc0,
                    const double *__restrict__ e3x1, const double *__restrict__ e2x1,
                    const double *__restrict__ e1x2, const double *__restrict__ e3x2,
                    const double *__restrict__ e2x3, const double *__restrict__ e1x3,
                    const double *__restrict__ flx1, const double *__restrict__ flx2,
                    const double *__restrict__ flx3, double *__restrict__ e1, double *__restrict__ e2,
                    double *__restrict__ e3, int ckl, int nchunk) {
  const long p = ((long)blockIdx.x*CY + threadIdx.y)*(CX - 1) + (long)threadIdx.x - 1;
  const long pc = p < 0 ? 0 : p;
  const int jj = (int)(pc/g.N1);
  const int i = (int)(pc - (long)jj*g.N1);
  const int j = g.js + jj;
  const int m = blockIdx.z/nchunk;
  const int k0 = g.ks + (blockIdx.z - m*nchunk)*ckl;
  const int k1 = (k0 + ckl - 1 < g.ke + 1) ? k0 + ckl - 1 : g.ke + 1;
  // the lane reads cells / faces (k, j and j-1, i): i in [is-1, ie+1] (the column is-1 only provides)
  const bool rd = p >= 0 && j <= g.je + 1 && i >= g.is - 1 && i <= g.ie + 1;
  const bool wr = rd && threadIdx.x != 0 && i >= g.is;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t fs1 = (size_t)g.N3*g.N2*(g.N1 + 1), fs2 = (size_t)g.N3*(g.N2 + 1)*g.N1,
               fs3 = (size_t)(g.N3 + 1)*g.N2*g.N1;
  const double *wm = w0 + (size_t)m*g.nvar*cs, *bm = bcc0 + (size_t)m*3*cs;
  const double *f1m = flx1 + (size_t)m*g.nvar*fs1, *f2m = flx2 + (size_t)m*g.nvar*fs2,
               *f3m = flx3 + (size_t)m*g.nvar*fs3;
  const unsigned N1 = (unsigned)g.N1, N2 = (unsigned)g.N2;
  const unsigned row = (unsigned)j*N1 + (unsigned)i;                      // (j,i) in a cell-shaped plane
  // byte offsets of (k0-1, j, i) in the array shapes involved; they advance by one plane per step
  unsigned oc = (((unsigned)(k0 - 1)*N2)*N1 + row)*8u;                                  // (N3, N2, N1)
  unsigned o1 = ((unsigned)(k0 - 1)*N2*(N1 + 1) + (unsigned)j*(N1 + 1) + (unsigned)i)*8u;   // (N3, N2, N1+1)
  unsigned o2 = ((unsigned)(k0 - 1)*(N2 + 1)*N1 + row)*8u;                              // (N3, N2+1, N1)
  unsigned o3 = oc;                                                                   // (N3+1, N2, N1): same rows
  const unsigned pc8 = N2*N1*8u, p18 = N2*(N1 + 1)*8u, p28 = (N2 + 1)*N1*8u;
  // edge arrays: e1 (N3+1, N2+1, N1), e2 (N3+1, N2, N1+1), e3 (N3, N2+1, N1+1)
  unsigned q1 = ((unsigned)k0*(N2 + 1)*N1 + row)*8u;
  unsigned q2 = ((unsigned)k0*N2*(N1 + 1) + (unsigned)j*(N1 + 1) + (unsigned)i)*8u;
  unsigned q3 = ((unsigned)k0*(N2 + 1)*(N1 + 1) + (unsigned)j*(N1 + 1) + (unsigned)i)*8u;
  const unsigned r18 = (N2 + 1)*N1*8u, r28 = N2*(N1 + 1)*8u, r38 = (N2 + 1)*(N1 + 1)*8u;
  double *e1m = e1 + (size_t)m*(g.N3 + 1)*(g.N2 + 1)*g.N1, *e2m = e2 + (size_t)m*(g.N3 + 1)*g.N2*(g.N1 + 1),
         *e3m = e3 + (size_t)m*g.N3*(g.N2 + 1)*(g.N1 + 1);
  const double *x31 = e3x1 + (size_t)m*cs, *x21 = e2x1 + (size_t)m*cs, *x12 = e1x2 + (size_t)m*cs,
               *x32 = e3x2 + (size_t)m*cs, *x23 = e2x3 + (size_t)m*cs, *x13 = e1x3 + (size_t)m*cs;
  // cell-centred E of cell (k,j,i) -> c1, c2, c3 and of cell (k,j-1,i) -> c1, c3 (mhd_corner_e.cpp:309-317)
  auto ecc = [&](unsigned o, double &c1, double &c2, double &c3) {
    const double vx = ldu(wm + cs, o), vy = ldu(wm + 2*cs, o), vz = ldu(wm + 3*cs, o);
    const double bx = ldu(bm, o), by = ldu(bm + cs, o), bz = ldu(bm + 2*cs, o);
    c1 = vz*by - vy*bz; c2 = vx*bz - vz*bx; c3 = vy*bx - vx*by;
  };
  // plane k0-1: the operands the first step needs from below
  double f1_km = 0.0, f2_km = 0.0, x2_km = 0.0, x1_km = 0.0, c1_mm = 0.0, c1_m0 = 0.0, c2_m0 = 0.0;
  if (rd) {
    double d2, d3;
    ecc(oc, c1_m0, c2_m0, d3);
    ecc(oc - N1*8u, c1_mm, d2, d3);
    f1_km = ldu(f1m, o1); f2_km = ldu(f2m, o2);
    x2_km = ldu(x12, oc); x1_km = ldu(x21, oc);
  }
  double c2_mm = __shfl_up(c2_m0, 1, 64);
  for (int k = k0; k <= k1; ++k) {
    oc += pc8; o1 += p18; o2 += p28; o3 += pc8;
    double c1_00 = 0.0, c2_00 = 0.0, c3_00 = 0.0, c1_0m = 0.0, c3_m0 = 0.0;
    double f1_k = 0.0, f1_jm = 0.0, f2_k = 0.0, f3_k = 0.0, f3_jm = 0.0;
    double x2_k = 0.0, x1_k = 0.0, x13_jm = 0.0, x13_j = 0.0, x23_i = 0.0, x32_i = 0.0, x31_jm = 0.0, x31_j = 0.0;
    if (rd) {
      double d2;
      ecc(oc, c1_00, c2_00, c3_00);
      ecc(oc - N1*8u, c1_0m, d2, c3_m0);
      f1_k = ldu(f1m, o1); f1_jm = ldu(f1m - (g.N1 + 1), o1);
      f2_k = ldu(f2m, o2);
      f3_k = ldu(f3m, o3); f3_jm = ldu(f3m - g.N1, o3);
      x2_k = ldu(x12, oc); x1_k = ldu(x21, oc);
      x13_jm = ldu(x13 - g.N1, oc); x13_j = ldu(x13, oc);
      x23_i = ldu(x23, oc); x32_i = ldu(x32, oc);
      x31_jm = ldu(x31 - g.N1, oc); x31_j = ldu(x31, oc);
    }
    const double c2_0m = __shfl_up(c2_00, 1, 64), c3_0m = __shfl_up(c3_00, 1, 64), c3_mm = __shfl_up(c3_m0, 1, 64);
    const double f2_im = __shfl_up(f2_k, 1, 64), f3_im = __shfl_up(f3_k, 1, 64);
    const double x23_im = __shfl_up(x23_i, 1, 64), x32_im = __shfl_up(x32_i, 1, 64);
    if (wr) {
      {  // E1 (mhd_corner_e.cpp:340-363)
        const double e1_l3 = upw_sel(f2_km >= 0.0, x13_jm, c1_mm, x13_j, c1_m0);
        const double e1_r3 = upw_sel(f2_k >= 0.0, x13_jm, c1_0m, x13_j, c1_00);
        const double e1_l2 = upw_sel(f3_jm >= 0.0, x2_km, c1_mm, x2_k, c1_0m);
        const double e1_r2 = upw_sel(f3_k >= 0.0, x2_km, c1_m0, x2_k, c1_00);
        stu(e1m, q1, 0.25*(e1_l3 + e1_r3 + e1_l2 + e1_r2 + x2_km + x2_k + x13_jm + x13_j));
      }
      {  // E2 (:365-388)
        const double e2_l3 = upw_sel(f1_km >= 0.0, x23_im, c2_mm, x23_i, c2_m0);
        const double e2_r3 = upw_sel(f1_k >= 0.0, x23_im, c2_0m, x23_i, c2_00);
        const double e2_l1 = upw_sel(f3_im >= 0.0, x1_km, c2_mm, x1_k, c2_0m);
        const double e2_r1 = upw_sel(f3_k >= 0.0, x1_km, c2_m0, x1_k, c2_00);
        stu(e2m, q2, 0.25*(e2_l3 + e2_r3 + e2_l1 + e2_r1 + x23_im + x23_i + x1_km + x1_k));
      }
      {  // E3 (:390-413)
        const double e3_l2 = upw_sel(f1_jm >= 0.0, x32_im, c3_mm, x32_i, c3_m0);
        const double e3_r2 = upw_sel(f1_k >= 0.0, x32_im, c3_0m, x32_i, c3_00);
        const double e3_l1 = upw_sel(f2_im >= 0.0, x31_jm, c3_mm, x31_j, c3_0m);
        const double e3_r1 = upw_sel(f2_k >= 0.0, x31_jm, c3_m0, x31_j, c3_00);
        stu(e3m, q3, 0.25*(e3_l1 + e3_r1 + e3_l2 + e3_r2 + x32_im + x32_i + x31_jm + x31_j));
      }
    }
    q1 += r18; q2 += r28; q3 += r38;
    f1_km = f1_k; f2_km = f2_k; x2_km = x2_k; x1_km = x1_k;
    c1_mm = c1_0m; c1_m0 = c1_00; c2_mm = c2_0m; c2_m0 = c2_00;
  }
}

__global__ void __launch_bounds__(BX*BY)
k_corner_e_3d(Geo g, EccAccess cc, const double *__restrict__ e3x1,
              const double *__restrict__ e2x1, const double *__restrict__ e1x2,
              const double *__restrict__ e3x2, const double *__restrict__ e2x3,
              const double *__restrict__ e1x3, const double *__restrict__ flx1,
              const double *__restrict__ flx2, const double *__restrict__ flx3,
              double *__restrict__ e1, double *__restrict__ e2, double *__restrict__ e3) {
  const int i = g.is + blockIdx.x*BX + threadIdx.x;
  const int j = g.js + blockIdx.y*BY + threadIdx.y;
  const int nk = g.ke - g.ks + 2;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + 1 || j > g.je + 1) return;
  // E1 (mhd_corner_e.cpp:340-363)
  double e1_l3, e1_r3, e1_l2, e1_r2;
  if (F2D(m, k - 1, j, i) >= 0.0) e1_l3 = CCE(e1x3, m, k, j - 1, i) - cc.e1(m, k - 1, j - 1, i);
  else                            e1_l3 = CCE(e1x3, m, k, j, i) - cc.e1(m, k - 1, j, i);
  if (F2D(m, k, j, i) >= 0.0)     e1_r3 = CCE(e1x3, m, k, j - 1, i) - cc.e1(m, k, j - 1, i);
  else                            e1_r3 = CCE(e1x3, m, k, j, i) - cc.e1(m, k, j, i);
  if (F3D(m, k, j - 1, i) >= 0.0) e1_l2 = CCE(e1x2, m, k - 1, j, i) - cc.e1(m, k - 1, j - 1, i);
  else                            e1_l2 = CCE(e1x2, m, k, j, i) - cc.e1(m, k, j - 1, i);
  if (F3D(m, k, j, i) >= 0.0)     e1_r2 = CCE(e1x2, m, k - 1, j, i) - cc.e1(m, k - 1, j, i);
  else                            e1_r2 = CCE(e1x2, m, k, j, i) - cc.e1(m, k, j, i);
  e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, k, j, i)] =
      0.25*(e1_l3 + e1_r3 + e1_l2 + e1_r2 + CCE(e1x2, m, k - 1, j, i) + CCE(e1x2, m, k, j, i) +
            CCE(e1x3, m, k, j - 1, i) + CCE(e1x3, m, k, j, i));
  // E2 (:365-388)
  double e2_l3, e2_r3, e2_l1, e2_r1;
  if (F1D(m, k - 1, j, i) >= 0.0) e2_l3 = CCE(e2x3, m, k, j, i - 1) - cc.e2(m, k - 1, j, i - 1);
  else                            e2_l3 = CCE(e2x3, m, k, j, i) - cc.e2(m, k - 1, j, i);
  if (F1D(m, k, j, i) >= 0.0)     e2_r3 = CCE(e2x3, m, k, j, i - 1) - cc.e2(m, k, j, i - 1);
  else                            e2_r3 = CCE(e2x3, m, k, j, i) - cc.e2(m, k, j, i);
  if (F3D(m, k, j, i - 1) >= 0.0) e2_l1 = CCE(e2x1, m, k - 1, j, i) - cc.e2(m, k - 1, j, i - 1);
  else                            e2_l1 = CCE(e2x1, m, k, j, i) - cc.e2(m, k, j, i - 1);
  if (F3D(m, k, j, i) >= 0.0)     e2_r1 = CCE(e2x1, m, k - 1, j, i) - cc.e2(m, k - 1, j, i);
  else                            e2_r1 = CCE(e2x1, m, k, j, i) - cc.e2(m, k, j, i);
  e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, k, j, i)] =
      0.25*(e2_l3 + e2_r3 + e2_l1 + e2_r1 + CCE(e2x3, m, k, j, i - 1) + CCE(e2x3, m, k, j, i) +
            CCE(e2x1, m, k - 1, j, i) + CCE(e2x1, m, k, j, i));
  // E3 (:390-413)
  e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, k, j, i)] =
      corner_e3(g, cc, e3x1, e3x2, flx1, flx2, m, k, j, i);
}

// ---------------------------------------------------------------------------------------
// CT, mhd_ct.cpp:23-80: all three face components in one launch.
__global__ void __launch_bounds__(BX*BY)
k_ct(Geo g, double gam0, double gam1, double beta_dt, const double *__restrict__ e1,
     const double *__restrict__ e2, const double *__restrict__ e3, double *__restrict__ b0x1f,
     double *__restrict__ b0x2f, double *__restrict__ b0x3f, const double *__restrict__ b1x1f,
     const double *__restrict__ b1x2f, const double *__restrict__ b1x3f) {
  const Cell3 q = flat_cells(FLAT_K, g.N1 + 1, g.is, g.ie + 1, g.js, g.je - g.js + 2, g.ks, g.ke - g.ks + 2);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  if (!q.in) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
#define E1(k, j, i) e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, k, j, i)]
#define E2(k, j, i) e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, k, j, i)]
#define E3(k, j, i) e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, k, j, i)]
  const bool p2 = is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);      // x/dx as v_ldexp_f64, see k_rk_update
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
#define DIVX(x, q) (p2 ? ldexp((x), n##q) : (x)/dx##q)
  if (g.multi_d && j <= g.je && k <= g.ke) {
    size_t c = ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i);
    double b = gam0*b0x1f[c] + gam1*b1x1f[c];
    b -= DIVX(beta_dt*(E3(k, j + 1, i) - E3(k, j, i)), 2);
    if (g.three_d) b += DIVX(beta_dt*(E2(k + 1, j, i) - E2(k, j, i)), 3);
    b0x1f[c] = b;
  }
  if (i <= g.ie && k <= g.ke) {
    size_t c = ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i);
    double b = gam0*b0x2f[c] + gam1*b1x2f[c];
    b += DIVX(beta_dt*(E3(k, j, i + 1) - E3(k, j, i)), 1);
    if (g.three_d) b -= DIVX(beta_dt*(E1(k + 1, j, i) - E1(k, j, i)), 3);
    b0x2f[c] = b;
  }
  if (i <= g.ie && j <= g.je) {
    size_t c = ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i);
    double b = gam0*b0x3f[c] + gam1*b1x3f[c];
    b -= DIVX(beta_dt*(E2(k, j, i + 1) - E2(k, j, i)), 1);
    if (g.multi_d) b += DIVX(beta_dt*(E1(k, j + 1, i) - E1(k, j, i)), 2);
    b0x3f[c] = b;
  }
#undef DIVX
#undef E1
#undef E2
#undef E3
}

// CT of the first stage out of place (see k_rk_update_oop): every face of the three arrays is written to the second
// register -- the faces CT updates (mhd_ct.cpp:45-77 ranges) with gam0*b + gam1*b -/+ ..., all others copied.
__global__ void __launch_bounds__(BX*BY)
k_ct_oop(Geo g, double gam0, double gam1, double beta_dt, const double *__restrict__ e1,
         const double *__restrict__ e2, const double *__restrict__ e3, const double *__restrict__ sx1f,
         const double *__restrict__ sx2f, const double *__restrict__ sx3f, double *__restrict__ dx1f,
         double *__restrict__ dx2f, double *__restrict__ dx3f) {
  const Cell3 q = flat_cells(FLAT_K, g.N1 + 1, 0, g.N1, 0, g.N2 + 1, 0, g.N3 + 1);      // every face index of the three arrays
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  if (!q.in) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
#define E1(k, j, i) e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, k, j, i)]
#define E2(k, j, i) e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, k, j, i)]
#define E3(k, j, i) e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, k, j, i)]
  const bool p2 = is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
#define DIVX(x, q) (p2 ? ldexp((x), n##q) : (x)/dx##q)
  // the ranges of k_ct: i in [is, ie+1], j in [js, je+1], k in [ks, ke+1], then per component
  const bool in = i >= g.is && i <= g.ie + 1 && j >= g.js && j <= g.je + 1 && k >= g.ks && k <= g.ke + 1;
  if (k < g.N3 && j < g.N2) {                                        // x1f (N3, N2, N1+1)
    const size_t c = ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i);
    double b = sx1f[c];
    if (in && g.multi_d && j <= g.je && k <= g.ke) {
      b = gam0*b + gam1*b;
      b -= DIVX(beta_dt*(E3(k, j + 1, i) - E3(k, j, i)), 2);
      if (g.three_d) b += DIVX(beta_dt*(E2(k + 1, j, i) - E2(k, j, i)), 3);
    }
    dx1f[c] = b;
  }
  if (k < g.N3 && i < g.N1) {                                        // x2f (N3, N2+1, N1)
    const size_t c = ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i);
    double b = sx2f[c];
    if (in && i <= g.ie && k <= g.ke) {
      b = gam0*b + gam1*b;
      b += DIVX(beta_dt*(E3(k, j, i + 1) - E3(k, j, i)), 1);
      if (g.three_d) b -= DIVX(beta_dt*(E1(k + 1, j, i) - E1(k, j, i)), 3);
    }
    dx2f[c] = b;
  }
  if (j < g.N2 && i < g.N1) {                                        // x3f (N3+1, N2, N1)
    const size_t c = ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i);
    double b = sx3f[c];
    if (in && i <= g.ie && j <= g.je) {
      b = gam0*b + gam1*b;
      b -= DIVX(beta_dt*(E2(k, j, i + 1) - E2(k, j, i)), 1);
      if (g.multi_d) b += DIVX(beta_dt*(E1(k, j + 1, i) - E1(k, j, i)), 2);
    }
    dx3f[c] = b;
  }
#undef DIVX
#undef E1
#undef E2
#undef E3
}

// Hydro::FOFC part 1 (hydro_fofc.cpp:46-85): trial update + floor test of one cell
__global__ void __launch_bounds__(BX*BY)
k_fofc_flag_hyd(Geo g, Eos eos, double gam0, double gam1, double beta_dt,
                const double *__restrict__ u0, const double *__restrict__ u1,
                const double *__restrict__ flx1, const double *__restrict__ flx2,
                const double *__restrict__ flx3, int fsh, int il, int iu, int jl, int ju, int kl,
                int nk, unsigned char *__restrict__ fofc, int *__restrict__ nfofc) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  const double dtodx1 = beta_dt/g.dx[3*m], dtodx2 = beta_dt/g.dx[3*m + 1];
  const double dtodx3 = beta_dt/g.dx[3*m + 2];
  double ut[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int n = 0; n < g.nvar; ++n) {
    double divf = dtodx1*(flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i + 1)] -
                          flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + fsh, m, n, k, j, i)]);
    if (g.multi_d)
      divf += dtodx2*(flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j + 1, i)] -
                      flx2[ix5(g.nvar, g.N3, g.N2 + fsh, g.N1, m, n, k, j, i)]);
    if (g.three_d)
      divf += dtodx3*(flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k + 1, j, i)] -
                      flx3[ix5(g.nvar, g.N3 + fsh, g.N2, g.N1, m, n, k, j, i)]);
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, n, k, j, i);
    ut[n] = gam0*u0[c] + gam1*u1[c] - divf;
  }
  bool fl;
  if (!eos.is_ideal) {
    fl = ut[0] < eos.dfloor;                 // isothermal_hyd.cpp: density floor only
  } else {
    double wd, wvx, wvy, wvz, we;
    bool dfl = false, efl = false, tfl = false;
    c2p_hyd(eos, ut[0], ut[1], ut[2], ut[3], ut[4], wd, wvx, wvy, wvz, we, dfl, efl, tfl);
    fl = dfl || efl || tfl;
  }
  if (fl) {
    fofc[ix4(g.N3, g.N2, g.N1, m, k, j, i)] = 1;
    atomicAdd(nfofc, 1);
  }
}

// Hydro::FOFC part 2 (hydro_fofc.cpp:100-366): a flagged cell rewrites the fluxes on its own faces
// with the first-order LLF flux of the two adjacent cell states.  Two flagged neighbours store
// the same value on their shared face.
__global__ void __launch_bounds__(BX*BY)
k_fofc_fix_hyd(Geo g, Eos eos, const double *__restrict__ w0, double *__restrict__ flx1,
               double *__restrict__ flx2, double *__restrict__ flx3, int fsh, int il, int iu,
               int jl, int ju, int kl, int nk, const unsigned char *__restrict__ fofc) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  if (!fofc[ix4(g.N3, g.N2, g.N1, m, k, j, i)]) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const int ndir = g.three_d ? 3 : (g.multi_d ? 2 : 1);
  for (int dir = 0; dir < ndir; ++dir) {
    const int ivx = 1 + dir, ivy = 1 + (dir + 1)%3, ivz = 1 + (dir + 2)%3;
    const int d1 = dir == 0, d2 = dir == 1, d3 = dir == 2;
    double *flx = dir == 0 ? flx1 : (dir == 1 ? flx2 : flx3);
    const int f1 = g.N1 + (d1 ? fsh : 0), f2 = g.N2 + (d2 ? fsh : 0), f3 = g.N3 + (d3 ? fsh : 0);
    const size_t fs = (size_t)f3*f2*f1;
    for (int side = 0; side < 2; ++side) {
      const int kf = k + side*d3, jf = j + side*d2, ic = i + side*d1;
      const double *ql = w0 + ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, kf - d3, jf - d2, ic - d1);
      const double *qr = w0 + ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, kf, jf, ic);
      double fd, fx, fy, fz, fe = 0.0;
      if (eos.is_ideal)
        llf_hyd(eos.gamma, ql[0], ql[ivx*cs], ql[ivy*cs], ql[ivz*cs], ql[4*cs], qr[0], qr[ivx*cs],
                qr[ivy*cs], qr[ivz*cs], qr[4*cs], fd, fx, fy, fz, fe);
      else
        llf_hyd_iso(eos.iso_cs, ql[0], ql[ivx*cs], ql[ivy*cs], ql[ivz*cs], qr[0], qr[ivx*cs],
                    qr[ivy*cs], qr[ivz*cs], fd, fx, fy, fz);
      double *f = flx + ix5(g.nvar, f3, f2, f1, m, 0, kf, jf, ic);
      f[0] = fd; f[ivx*fs] = fx; f[ivy*fs] = fy; f[ivz*fs] = fz;
      if (eos.is_ideal) f[4*fs] = fe;
    }
  }
}

// MHD::FOFC (mhd_fofc.cpp:30-493), ideal gas
struct FofcMhd {
  const double *w0, *bcc0, *b0[3], *b1[3], *u0, *u1;
  double *flx[3], *ey[3], *ez[3];      // per direction: flux, (e3x1,e1x2,e2x3), (e2x1,e3x2,e1x3)
};

__global__ void __launch_bounds__(BX*BY)
k_fofc_flag_mhd(Geo g, Eos eos, double gam0, double gam1, double beta_dt, FofcMhd a, int il, int iu,
                int jl, int ju, int kl, int nk, unsigned char *__restrict__ fofc,
                int *__restrict__ nfofc) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const double dtodx1 = beta_dt/g.dx[3*m], dtodx2 = beta_dt/g.dx[3*m + 1];
  const double dtodx3 = beta_dt/g.dx[3*m + 2];
  double ut[5];
  for (int n = 0; n < 5; ++n) {
    double divf = dtodx1*(a.flx[0][ix5(5, N3, N2, N1 + 1, m, n, k, j, i + 1)] -
                          a.flx[0][ix5(5, N3, N2, N1 + 1, m, n, k, j, i)]);
    if (g.multi_d)
      divf += dtodx2*(a.flx[1][ix5(5, N3, N2 + 1, N1, m, n, k, j + 1, i)] -
                      a.flx[1][ix5(5, N3, N2 + 1, N1, m, n, k, j, i)]);
    if (g.three_d)
      divf += dtodx3*(a.flx[2][ix5(5, N3 + 1, N2, N1, m, n, k + 1, j, i)] -
                      a.flx[2][ix5(5, N3 + 1, N2, N1, m, n, k, j, i)]);
    const size_t c = ix5(5, N3, N2, N1, m, n, k, j, i);
    ut[n] = gam0*a.u0[c] + gam1*a.u1[c] - divf;
  }
  // trial cell-centred field, mhd_fofc.cpp:88-107
  const double b1old = 0.5*(a.b1[0][ix4(N3, N2, N1 + 1, m, k, j, i)] + a.b1[0][ix4(N3, N2, N1 + 1, m, k, j, i + 1)]);
  const double b2old = 0.5*(a.b1[1][ix4(N3, N2 + 1, N1, m, k, j, i)] + a.b1[1][ix4(N3, N2 + 1, N1, m, k, j + 1, i)]);
  const double b3old = 0.5*(a.b1[2][ix4(N3 + 1, N2, N1, m, k, j, i)] + a.b1[2][ix4(N3 + 1, N2, N1, m, k + 1, j, i)]);
  const size_t cs = (size_t)N3*N2*N1;
  const size_t bc = ix5(3, N3, N2, N1, m, 0, k, j, i);
  double bx = gam0*a.bcc0[bc] + gam1*b1old;
  double by = gam0*a.bcc0[bc + cs] + gam1*b2old;
  double bz = gam0*a.bcc0[bc + 2*cs] + gam1*b3old;
  const size_t e = ix4(N3, N2, N1, m, k, j, i);
  by += dtodx1*(a.ey[0][e + 1] - a.ey[0][e]);              // e3x1
  bz -= dtodx1*(a.ez[0][e + 1] - a.ez[0][e]);              // e2x1
  if (g.multi_d) {
    bx -= dtodx2*(a.ez[1][e + N1] - a.ez[1][e]);           // e3x2
    bz += dtodx2*(a.ey[1][e + N1] - a.ey[1][e]);           // e1x2
  }
  if (g.three_d) {
    bx += dtodx3*(a.ey[2][e + (size_t)N2*N1] - a.ey[2][e]);   // e2x3
    by -= dtodx3*(a.ez[2][e + (size_t)N2*N1] - a.ez[2][e]);   // e1x3
  }
  double wd, wvx, wvy, wvz, we;
  bool dfl = false, efl = false, tfl = false;
  c2p_mhd(eos, ut[0], ut[1], ut[2], ut[3], ut[4], bx, by, bz, wd, wvx, wvy, wvz, we, dfl, efl, tfl);
  if (dfl || efl || tfl) {
    fofc[e] = 1;
    atomicAdd(nfofc, 1);
  }
}

__global__ void __launch_bounds__(BX*BY)
k_fofc_fix_mhd(Geo g, Eos eos, FofcMhd a, int il, int iu, int jl, int ju, int kl, int nk,
               const unsigned char *__restrict__ fofc) {
  const int i = il + blockIdx.x*BX + threadIdx.x;
  const int j = jl + blockIdx.y*BY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = kl + (blockIdx.z - m*nk);
  if (i > iu || j > ju) return;
  if (!fofc[ix4(g.N3, g.N2, g.N1, m, k, j, i)]) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const int ndir = g.three_d ? 3 : (g.multi_d ? 2 : 1);
  for (int dir = 0; dir < ndir; ++dir) {
    const int ivx = 1 + dir, ivy = 1 + (dir + 1)%3, ivz = 1 + (dir + 2)%3;
    const int iby = (dir + 1)%3, ibz = (dir + 2)%3;
    const int d1 = dir == 0, d2 = dir == 1, d3 = dir == 2;
    const int f1 = g.N1 + d1, f2 = g.N2 + d2, f3 = g.N3 + d3;
    const size_t fs = (size_t)f3*f2*f1;
    for (int side = 0; side < 2; ++side) {
      const int kf = k + side*d3, jf = j + side*d2, ic = i + side*d1;
      const double *ql = a.w0 + ix5(5, g.N3, g.N2, g.N1, m, 0, kf - d3, jf - d2, ic - d1);
      const double *qr = a.w0 + ix5(5, g.N3, g.N2, g.N1, m, 0, kf, jf, ic);
      const double *bl = a.bcc0 + ix5(3, g.N3, g.N2, g.N1, m, 0, kf - d3, jf - d2, ic - d1);
      const double *br = a.bcc0 + ix5(3, g.N3, g.N2, g.N1, m, 0, kf, jf, ic);
      const double bxi = a.b0[dir][ix4(f3, f2, f1, m, kf, jf, ic)];
      const Cons1D fl = llf_mhd(eos.gamma, ql[0], ql[ivx*cs], ql[ivy*cs], ql[ivz*cs], ql[4*cs],
                                bl[iby*cs], bl[ibz*cs], qr[0], qr[ivx*cs], qr[ivy*cs], qr[ivz*cs],
                                qr[4*cs], br[iby*cs], br[ibz*cs], bxi);
      double *f = a.flx[dir] + ix5(5, f3, f2, f1, m, 0, kf, jf, ic);
      f[0] = fl.d; f[ivx*fs] = fl.mx; f[ivy*fs] = fl.my; f[ivz*fs] = fl.mz; f[4*fs] = fl.e;
      const size_t ec = ix4(g.N3, g.N2, g.N1, m, kf, jf, ic);
      a.ey[dir][ec] = -fl.by;            // SingleStateLLF_MHD returns flux.by already negated (:83)
      a.ez[dir][ec] = fl.bz;
    }
  }
}

// NewTimeStep of kinematic runs (hydro_newdt.cpp:55-72): dx/|v| per direction.  |v| may be 0: dx/0 =
// inf never wins the min.  Reduced like k_newdt: the division is monotone, so min dx/|v| =
// dx/max|v| ... only for positive max; keep the per-cell form, it is a cold path.
__global__ void __launch_bounds__(BX*BY)
k_kinematic_newdt(Geo g, const double *__restrict__ w0, double *__restrict__ dt3) {
  const int i = g.is + blockIdx.x*BX + threadIdx.x;
  const int j = g.js + blockIdx.y*BY + threadIdx.y;
  const int nk = g.ke - g.ks + 1;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  double d1 = DBL_MAX, d2 = DBL_MAX, d3 = DBL_MAX;
  if (i <= g.ie && j <= g.je) {
    const size_t cs = (size_t)g.N3*g.N2*g.N1;
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
    d1 = g.dx[3*m]/fabs(w0[c + cs]);
    d2 = g.dx[3*m + 1]/fabs(w0[c + 2*cs]);
    d3 = g.dx[3*m + 2]/fabs(w0[c + 3*cs]);
  }
  block_min3_atomic(d1, d2, d3, dt3);
}

// Hydro::CopyCons for rk4 (hydro_tasks.cpp:134-148): u1 += delta*u0, active cells
__global__ void __launch_bounds__(256)
k_rk4_register(Geo g, double delta, const double *__restrict__ u0, double *__restrict__ u1) {
  const int i = g.is + blockIdx.x*256 + threadIdx.x;
  if (i > g.ie) return;
  const int j = g.js + blockIdx.y%g.nx2, k = g.ks + blockIdx.y/g.nx2;
  const size_t c = (((size_t)blockIdx.z*g.N3 + k)*g.N2 + j)*g.N1 + i;   // blockIdx.z = m*nvar + n
  u1[c] += delta*u0[c];
}

}  // namespace akmi

using namespace akmi;

extern "C" {

const char *akmi_last_error(void) { return akmi::g_err; }
int akmi_version(void) { return 100; }

int akmi_copy_cons(const akmi_pack *p, const double *u0, double *u1, void *stream) {
  Geo g = make_geo(p);
  size_t n = (size_t)g.nmb*g.nvar*g.N3*g.N2*g.N1*sizeof(double);
  hipError_t e = hipMemcpyAsync(u1, u0, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("copy_cons: %s", hipGetErrorString(e)); return AKMI_FAIL; }
  return AKMI_COMPLETE;
}

int akmi_rk4_copy_cons(const akmi_pack *p, double delta, const double *u0, double *u1, void *stream) {
  Geo g = make_geo(p);
  dim3 grid((g.nx1 + 255)/256, g.nx2*g.nx3, g.nmb*g.nvar);
  k_rk4_register<<<grid, 256, 0, (hipStream_t)stream>>>(g, delta, u0, u1);
  AKMI_CHECK_LAUNCH("rk4_copy_cons");
  return AKMI_COMPLETE;
}

static int fofc_checks(const akmi_pack *p, int recon, const char *what, bool ideal_only);

// The flux kernels below are thread-per-face; the sweeps of the fused stage compute the same fluxes with
// the reconstruction of a cell done once (akmi_stage.hip, sweeps_store_fluxes) and take over wherever they
// cover the request: no passive scalars (those ride on the stored mass flux in the fused path), not the
// kinematic "advect" solver, not the FOFC-extended ranges.
static bool use_sweeps(const akmi_pack *p, int rsolver, int ext) {
  return !ext && rsolver != AKMI_RS_ADVECT && p->nvar == (p->is_ideal ? 5 : 4);
}

static int hydro_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0, double *flx1,
                        double *flx2, double *flx3, int face_shaped, void *stream, int ext) {
  if (check_scheme(p, recon, "hydro_fluxes") != AKMI_COMPLETE) return AKMI_FAIL;
  if (ext && fofc_checks(p, recon, "hydro_fluxes_fofc", false) != AKMI_COMPLETE) return AKMI_FAIL;
  if (!p->is_ideal && rsolver == AKMI_RS_HLLC) {
    set_error("hydro_fluxes: rsolver = hllc needs the ideal-gas EOS"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  const Scheme sc{recon, rsolver, make_face_eos(p), !p->is_ideal};
  hipStream_t st = (hipStream_t)stream;
  int fsh = face_shaped ? 1 : 0;
  if (use_sweeps(p, rsolver, ext)) {
    const int r = sweeps_store_fluxes(p, recon, rsolver, w0, nullptr, nullptr, nullptr, nullptr, flx1, flx2, flx3,
                                      fsh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
    if (r != -1) return r;
  }
  int rc = launch_hydro_flux<0>(g, sc, w0, flx1, fsh, st, ext);
  if (rc == AKMI_COMPLETE && g.multi_d) rc = launch_hydro_flux<1>(g, sc, w0, flx2, fsh, st, ext);
  if (rc == AKMI_COMPLETE && g.three_d) rc = launch_hydro_flux<2>(g, sc, w0, flx3, fsh, st, ext);
  return rc;
}

int akmi_hydro_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                      double *flx1, double *flx2, double *flx3, int face_shaped,
                      void *stream) {
  return hydro_fluxes(p, recon, rsolver, w0, flx1, flx2, flx3, face_shaped, stream, 0);
}

int akmi_hydro_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                           double *flx1, double *flx2, double *flx3, int face_shaped,
                           void *stream) {
  return hydro_fluxes(p, recon, rsolver, w0, flx1, flx2, flx3, face_shaped, stream, 1);
}

int akmi_hydro_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                    const double *u0, const double *u1, double *flx1, double *flx2, double *flx3,
                    int face_shaped, unsigned char *fofc, int *nfofc, void *stream) {
  if (p->nvar != (p->is_ideal ? 5 : 4)) {
    set_error("hydro_fofc: FOFC with passive scalars is not on this path"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const int il = g.is - 1, iu = g.ie + 1;
  const int jl = g.multi_d ? g.js - 1 : g.js, ju = g.multi_d ? g.je + 1 : g.je;
  const int kl = g.three_d ? g.ks - 1 : g.ks, ku = g.three_d ? g.ke + 1 : g.ke;
  const int nk = ku - kl + 1;
  dim3 grid(cdiv(iu - il + 1, BX), cdiv(ju - jl + 1, BY), nk*g.nmb), block(BX, BY);
  const int fsh = face_shaped ? 1 : 0;
  k_fofc_flag_hyd<<<grid, block, 0, st>>>(g, make_eos(p), gam0, gam1, beta_dt, u0, u1, flx1, flx2,
                                          flx3, fsh, il, iu, jl, ju, kl, nk, fofc, nfofc);
  k_fofc_fix_hyd<<<grid, block, 0, st>>>(g, make_eos(p), w0, flx1, flx2, flx3, fsh, il, iu, jl, ju,
                                         kl, nk, fofc);
  AKMI_CHECK_LAUNCH("hydro_fofc");
  // "reset FOFC flag" (hydro_fofc.cpp:364) once every flagged cell has been processed
  hipError_t e = hipMemsetAsync(fofc, 0, (size_t)g.nmb*g.N3*g.N2*g.N1, st);
  if (e != hipSuccess) { set_error("hydro_fofc: %s", hipGetErrorString(e)); return AKMI_FAIL; }
  return AKMI_COMPLETE;
}

int akmi_rk_update(const akmi_pack *p, double gam0, double gam1, double beta_dt, double *u0,
                   const double *u1, const double *flx1, const double *flx2,
                   const double *flx3, int face_shaped, void *stream) {
  Geo g = make_geo(p);
  // MeshBlocks whose planes of active cells fill at most one workgroup: planes and columns flattened (120 blocks of 16^3:
  // 40 -> 34 us); larger ones keep whole rows and one group of workgroups per plane (960 blocks of 32^3: 1 950 us against
  // 2 040 with the planes flattened and 2 180 with the columns too) -- profiles/r06_lane_mapping.txt
  const int fm = (g.three_d && g.nx1*g.nx2 <= BX*BY) ? (FLAT_K | FLAT_COLS) : 0;
  dim3 grid = flat_cells_grid(fm, g.N1, g.is, g.ie, g.nx2, g.nx3, g.nmb), block(BX, BY);
  k_rk_update<<<grid, block, 0, (hipStream_t)stream>>>(g, gam0, gam1, beta_dt, u0, u1, flx1, flx2,
                                                       flx3, face_shaped ? 1 : 0, fm);
  AKMI_CHECK_LAUNCH("rk_update");
  return AKMI_COMPLETE;
}

int akmi_rk_update_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *u0, double *u1,
                       const double *flx1, const double *flx2, const double *flx3, int face_shaped, void *stream) {
  Geo g = make_geo(p);
  dim3 grid = flat_cells_grid(FLAT_K, g.N1, 0, g.N1 - 1, g.N2, g.N3, g.nmb), block(BX, BY);
  k_rk_update_oop<<<grid, block, 0, (hipStream_t)stream>>>(g, gam0, gam1, beta_dt, u0, u1, flx1, flx2, flx3,
                                                           face_shaped ? 1 : 0);
  AKMI_CHECK_LAUNCH("rk_update_oop");
  return AKMI_COMPLETE;
}

int akmi_hydro_c2p(const akmi_pack *p, double *u0, double *w0, int il, int iu, int jl, int ju,
                   int kl, int ku, int *counters, void *stream) {
  Geo g = make_geo(p);
  int nk = ku - kl + 1;
  dim3 grid = flat_cells_grid(0, g.N1, il, iu, ju - jl + 1, nk, g.nmb), block(BX, BY);
  k_c2p<false><<<grid, block, 0, (hipStream_t)stream>>>(g, make_eos(p), u0, nullptr, nullptr,
      nullptr, w0, nullptr, il, iu, jl, ju, kl, nk, counters);
  AKMI_CHECK_LAUNCH("hydro_c2p");
  return AKMI_COMPLETE;
}

int akmi_mhd_c2p(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                 const double *bx3f, double *w0, double *bcc0, int il, int iu, int jl, int ju,
                 int kl, int ku, int *counters, void *stream) {
  Geo g = make_geo(p);
  int nk = ku - kl + 1;
  dim3 grid = flat_cells_grid(0, g.N1, il, iu, ju - jl + 1, nk, g.nmb), block(BX, BY);
  k_c2p<true><<<grid, block, 0, (hipStream_t)stream>>>(g, make_eos(p), u0, bx1f, bx2f, bx3f, w0,
      bcc0, il, iu, jl, ju, kl, nk, counters);
  AKMI_CHECK_LAUNCH("mhd_c2p");
  return AKMI_COMPLETE;
}

int akmi_hydro_newdt(const akmi_pack *p, const double *w0, double *dt3, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  k_init_dt<<<1, 64, 0, st>>>(dt3);
  dim3 grid = flat_cells_grid(0, g.N1, g.is, g.ie, g.nx2, g.nx3, g.nmb), block(BX, BY);
  k_newdt<false><<<grid, block, 0, st>>>(g, make_eos(p), w0, nullptr, dt3);
  AKMI_CHECK_LAUNCH("hydro_newdt");
  return AKMI_COMPLETE;
}

int akmi_kinematic_newdt(const akmi_pack *p, const double *w0, double *dt3, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  k_init_dt<<<1, 64, 0, st>>>(dt3);
  dim3 grid(cdiv(g.nx1, BX), cdiv(g.je - g.js + 1, BY), (g.ke - g.ks + 1)*g.nmb), block(BX, BY);
  k_kinematic_newdt<<<grid, block, 0, st>>>(g, w0, dt3);
  AKMI_CHECK_LAUNCH("kinematic_newdt");
  return AKMI_COMPLETE;
}

int akmi_mhd_newdt(const akmi_pack *p, const double *w0, const double *bcc0, double *dt3,
                   void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  k_init_dt<<<1, 64, 0, st>>>(dt3);
  dim3 grid = flat_cells_grid(0, g.N1, g.is, g.ie, g.nx2, g.nx3, g.nmb), block(BX, BY);
  k_newdt<true><<<grid, block, 0, st>>>(g, make_eos(p), w0, bcc0, dt3);
  AKMI_CHECK_LAUNCH("mhd_newdt");
  return AKMI_COMPLETE;
}

static int fofc_checks(const akmi_pack *p, int recon, const char *what, bool ideal_only) {
  if (p->nvar != (p->is_ideal ? 5 : 4) || (ideal_only && !p->is_ideal)) {
    set_error("%s: FOFC with passive scalars%s is not on this path", what,
              ideal_only ? " or the isothermal EOS" : "");
    return AKMI_FAIL;
  }
  const int need = recon == AKMI_RECON_PLM ? 3 : (recon >= AKMI_RECON_PPM4 ? 4 : 2);
  if (p->ng < need) {              // src/hydro/hydro.cpp:163-190, src/mhd/mhd.cpp:211-235
    set_error("%s: FOFC and this reconstruction require at least %d ghost zones, but nghost=%d",
              what, need, p->ng);
    return AKMI_FAIL;
  }
  return AKMI_COMPLETE;
}

static int mhd_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                      const double *bcc0, const double *bx1f, const double *bx2f,
                      const double *bx3f, double *flx1, double *flx2, double *flx3, double *e3x1,
                      double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                      void *stream, int ext) {
  if (check_scheme(p, recon, "mhd_fluxes") != AKMI_COMPLETE) return AKMI_FAIL;
  if (ext && fofc_checks(p, recon, "mhd_fluxes_fofc", true) != AKMI_COMPLETE) return AKMI_FAIL;
  Geo g = make_geo(p);
  const Scheme sc{recon, rsolver, make_face_eos(p), !p->is_ideal};
  hipStream_t st = (hipStream_t)stream;
  if (use_sweeps(p, rsolver, ext)) {
    const int r = sweeps_store_fluxes(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, 1, e3x1,
                                      e2x1, e1x2, e3x2, e2x3, e1x3, stream);
    if (r != -1) return r;
  }
  int rc = launch_mhd_flux<0>(g, sc, w0, bcc0, bx1f, flx1, e3x1, e2x1, st, ext);
  if (rc == AKMI_COMPLETE && g.multi_d)
    rc = launch_mhd_flux<1>(g, sc, w0, bcc0, bx2f, flx2, e1x2, e3x2, st, ext);
  if (rc == AKMI_COMPLETE && g.three_d)
    rc = launch_mhd_flux<2>(g, sc, w0, bcc0, bx3f, flx3, e2x3, e1x3, st, ext);
  return rc;
}

int akmi_mhd_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                    const double *bcc0, const double *bx1f, const double *bx2f,
                    const double *bx3f, double *flx1, double *flx2, double *flx3, double *e3x1,
                    double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                    void *stream) {
  return mhd_fluxes(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, e3x1, e2x1,
                    e1x2, e3x2, e2x3, e1x3, stream, 0);
}

int akmi_mhd_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                         const double *bcc0, const double *bx1f, const double *bx2f,
                         const double *bx3f, double *flx1, double *flx2, double *flx3, double *e3x1,
                         double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                         void *stream) {
  return mhd_fluxes(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, e3x1, e2x1,
                    e1x2, e3x2, e2x3, e1x3, stream, 1);
}

int akmi_mhd_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                  const double *bcc0, const double *b0x1f, const double *b0x2f, const double *b0x3f,
                  const double *b1x1f, const double *b1x2f, const double *b1x3f, const double *u0,
                  const double *u1, double *flx1, double *flx2, double *flx3, double *e3x1,
                  double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                  unsigned char *fofc, int *nfofc, void *stream) {
  if (!p->is_ideal || p->nvar != 5) {
    set_error("mhd_fofc: ideal gas without passive scalars only"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const int il = g.is - 1, iu = g.ie + 1;
  const int jl = g.multi_d ? g.js - 1 : g.js, ju = g.multi_d ? g.je + 1 : g.je;
  const int kl = g.three_d ? g.ks - 1 : g.ks, ku = g.three_d ? g.ke + 1 : g.ke;
  const int nk = ku - kl + 1;
  dim3 grid(cdiv(iu - il + 1, BX), cdiv(ju - jl + 1, BY), nk*g.nmb), block(BX, BY);
  FofcMhd a{w0, bcc0, {b0x1f, b0x2f, b0x3f}, {b1x1f, b1x2f, b1x3f}, u0, u1, {flx1, flx2, flx3},
            {e3x1, e1x2, e2x3}, {e2x1, e3x2, e1x3}};
  k_fofc_flag_mhd<<<grid, block, 0, st>>>(g, make_eos(p), gam0, gam1, beta_dt, a, il, iu, jl, ju, kl,
                                          nk, fofc, nfofc);
  k_fofc_fix_mhd<<<grid, block, 0, st>>>(g, make_eos(p), a, il, iu, jl, ju, kl, nk, fofc);
  AKMI_CHECK_LAUNCH("mhd_fofc");
  hipError_t e = hipMemsetAsync(fofc, 0, (size_t)g.nmb*g.N3*g.N2*g.N1, st);   // mhd_fofc.cpp:487-489
  if (e != hipSuccess) { set_error("mhd_fofc: %s", hipGetErrorString(e)); return AKMI_FAIL; }
  return AKMI_COMPLETE;
}

int akmi_history_sums(const akmi_pack *p, int is_mhd, const double *u0, const double *bx1f,
                      const double *bx2f, const double *bx3f, double *out, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const int nh = is_mhd ? 11 : 8;
  if (hipMemsetAsync(out, 0, sizeof(double)*nh, st) != hipSuccess) {
    set_error("history_sums: memset failed"); return AKMI_FAIL;
  }
  dim3 grid(cdiv(g.nx1, BX), cdiv(g.je - g.js + 1, BY), (g.ke - g.ks + 1)*g.nmb), block(BX, BY);
  if (is_mhd) k_history<true><<<grid, block, 0, st>>>(g, p->is_ideal, u0, bx1f, bx2f, bx3f, out);
  else k_history<false><<<grid, block, 0, st>>>(g, p->is_ideal, u0, nullptr, nullptr, nullptr, out);
  AKMI_CHECK_LAUNCH("history_sums");
  return AKMI_COMPLETE;
}

int akmi_mhd_corner_e(const akmi_pack *p, const double *w0, const double *bcc0,
                      const double *e3x1, const double *e2x1, const double *e1x2,
                      const double *e3x2, const double *e2x3, const double *e1x3,
                      const double *flx1, const double *flx2, const double *flx3, double *e1,
                      double *e2, double *e3, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  EccAccess cc{w0, bcc0, g.nvar, g.N3, g.N2, g.N1, (size_t)g.N3*g.N2*g.N1};
  if (!g.multi_d) {
    dim3 grid(cdiv(g.nx1 + 1, 256), g.nmb);
    k_corner_e_1d<<<grid, 256, 0, st>>>(g, e3x1, e2x1, e2, e3);
  } else if (!g.three_d) {
    dim3 grid(cdiv(g.nx1 + 1, BX), cdiv(g.nx2 + 1, BY), g.nmb), block(BX, BY);
    k_corner_e_2d<<<grid, block, 0, st>>>(g, cc, e3x1, e2x1, e1x2, e3x2, flx1, flx2, e1, e2, e3);
  } else {
    if ((size_t)(g.N3 + 1)*(g.N2 + 1)*(g.N1 + 1)*sizeof(double) < ((size_t)1 << 32)) {
      const long np = (long)(g.nx2 + 1)*g.N1;                      // flattened rows js..je+1
      const long per_wg = (long)(CX - 1)*CY;
      const unsigned nb = (unsigned)((np + 1 + per_wg - 1)/per_wg);
      const int nk = g.nx3 + 1;
      // ~2048 workgroups per launch, chunks of EQUAL length (a chunk primes its k-1 operands: 33 planes as 32 + 1 cost
      // a second workgroup per tile for one plane -- 960 blocks of 32^3: 1978 -> see profiles/r04_c5f_config5.txt)
      long nch = 2048/((long)nb*g.nmb);
      nch = nch < 1 ? 1 : (nch > (nk + 3)/4 ? (nk + 3)/4 : nch);
      long ckl = (nk + nch - 1)/nch;
      ckl = ckl > 48 ? 48 : ckl;
      const int nchunk = cdiv(nk, (int)ckl);
      dim3 grid(nb, 1, nchunk*g.nmb), block(CX, CY);
      k_corner_e_3d_march<<<grid, block, 0, st>>>(g, w0, bcc0, e3x1, e2x1, e1x2, e3x2, e2x3, e1x3, flx1, flx2,
                                                 flx3, e1, e2, e3, (int)ckl, nchunk);
    } else {
      dim3 grid(cdiv(g.nx1 + 1, BX), cdiv(g.nx2 + 1, BY), (g.nx3 + 1)*g.nmb), block(BX, BY);
      k_corner_e_3d<<<grid, block, 0, st>>>(g, cc, e3x1, e2x1, e1x2, e3x2, e2x3, e1x3, flx1, flx2,
                                           flx3, e1, e2, e3);
    }
  }
  AKMI_CHECK_LAUNCH("corner_e");
  return AKMI_COMPLETE;
}

int akmi_mhd_ct(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *e1,
                const double *e2, const double *e3, double *b0x1f, double *b0x2f, double *b0x3f,
                const double *b1x1f, const double *b1x2f, const double *b1x3f, void *stream) {
  Geo g = make_geo(p);
  dim3 grid = flat_cells_grid(FLAT_K, g.N1 + 1, g.is, g.ie + 1, g.je - g.js + 2, g.ke - g.ks + 2, g.nmb), block(BX, BY);
  k_ct<<<grid, block, 0, (hipStream_t)stream>>>(g, gam0, gam1, beta_dt, e1, e2, e3, b0x1f, b0x2f,
                                                b0x3f, b1x1f, b1x2f, b1x3f);
  AKMI_CHECK_LAUNCH("ct");
  return AKMI_COMPLETE;
}

int akmi_mhd_ct_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *e1, const double *e2,
                    const double *e3, const double *b0x1f, const double *b0x2f, const double *b0x3f, double *b1x1f,
                    double *b1x2f, double *b1x3f, void *stream) {
  Geo g = make_geo(p);
  dim3 grid = flat_cells_grid(FLAT_K, g.N1 + 1, 0, g.N1, g.N2 + 1, g.N3 + 1, g.nmb), block(BX, BY);
  k_ct_oop<<<grid, block, 0, (hipStream_t)stream>>>(g, gam0, gam1, beta_dt, e1, e2, e3, b0x1f, b0x2f, b0x3f, b1x1f,
                                                    b1x2f, b1x3f);
  AKMI_CHECK_LAUNCH("ct_oop");
  return AKMI_COMPLETE;
}

}  // extern "C"
