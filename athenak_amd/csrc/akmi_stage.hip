// akmi_stage.hip -- per-stage fast path of a MeshBlockPack (C ABI: akmi_*_stage_update,
// akmi_*_c2p_newdt).  Results are bit-identical to the task chain of akmi_tasks.hip.
//
// Pass A  (akmi_*_stage_update): flux sweeps with everything memory-bound folded into the
//   VALU-bound Riemann kernels:
//     sweep x1 .. x(D-1): reconstruct + Riemann, write face fluxes (+ face EMFs); the x1 sweep
//                         also emits the cell-centred E = -(v x B) that CornerE needs
//     sweep xD (last)   : reconstruct + Riemann, exchange the normal flux of a tile through
//                         LDS and apply the RK update  u0 = gam0*u0 + gam1*u1 - beta_dt*divF
//                         in the same kernel (stage 1 also stores u1 <- u0: CopyCons folded)
//     corner E + CT     : GS05/07 upwind corner EMFs (select-based, no divergent loads) and the
//                         face-B update in one k-marching kernel in 3-D (stage 1 also stores
//                         b1 <- b0); 1-D/2-D: akmi_mhd_corner_e + k_ct_copy
// Pass B  (akmi_*_c2p_newdt): ConsToPrim over all cells + (last stage) CFL scan in ONE
//   kernel: the three direction maxima of |v|+c_f are reduced per wavefront (DPP shuffles),
//   per workgroup (LDS) and the workgroup's dx/max enters a 64-bit atomicMin.  Division is
//   monotone, so min_cells fl(dx/a) == fl(dx/max_cells a): identical to the reference scan.
#include <cstdlib>
#include <map>
#include "akmi_common.hpp"

using namespace akmi;

namespace akmi {

constexpr int SX = 64, SY = 4;     // plain flux kernels: one wave per row, 4 rows
constexpr int TX = 64;             // 1-D tile width

struct StageWs {
  double *flx1, *flx2, *flx3;
  double *efc[6];
  double *ecc[3];
  double *e1, *e2, *e3;
  double *acc;                    // dF1/dx1 + dF2/dx2 per cell and variable (3-D path)
  size_t total;
};

static StageWs carve(const Geo &g, int is_mhd, void *ws) {
  StageWs w;
  double *p = (double *)ws;
  size_t nmb = g.nmb, nv = g.nvar;
  size_t n1 = nmb*nv*g.N3*g.N2*(g.N1 + 1), n2 = nmb*nv*g.N3*(g.N2 + 1)*g.N1,
         n3 = nmb*nv*(g.N3 + 1)*g.N2*g.N1;
  size_t off = 0;
  auto take = [&](size_t n) { double *r = p ? p + off : nullptr; off += (n + 31) & ~(size_t)31; return r; };
  w.flx1 = take(n1); w.flx2 = take(n2); w.flx3 = take(n3);
  w.acc = take(nmb*nv*g.N3*g.N2*g.N1);
  for (int q = 0; q < 6; ++q) w.efc[q] = nullptr;
  for (int q = 0; q < 3; ++q) w.ecc[q] = nullptr;
  w.e1 = w.e2 = w.e3 = nullptr;
  if (is_mhd) {
    size_t nc = nmb*g.N3*g.N2*g.N1;
    for (int q = 0; q < 6; ++q) w.efc[q] = take(nc);
    for (int q = 0; q < 3; ++q) w.ecc[q] = take(nc);
    if (!g.three_d) {       // 3-D: the corner EMFs never leave k_corner_ct
      w.e1 = take(nmb*(g.N3 + 1)*(g.N2 + 1)*g.N1);
      w.e2 = take(nmb*(g.N3 + 1)*g.N2*(g.N1 + 1));
      w.e3 = take(nmb*g.N3*(g.N2 + 1)*(g.N1 + 1));
    }
  }
  w.total = off*sizeof(double);
  return w;
}

// ---------------------------------------------------------------------------------------
// face flux from the cell stencil (registers only).  Returns flux in sweep-aligned order.
template <int DIR, int RECON, bool MHD, int RS>
__device__ __forceinline__ void face_flux(const Geo &g, const FaceEos &eos,
    const double *__restrict__ w0, const double *__restrict__ bcc0,
    const double *__restrict__ bxf, int f3, int f2, int f1, int m, int k, int j, int i,
    double &fd, double &fx, double &fy, double &fz, double &fe, double &fby, double &fbz) {
  constexpr int ivx = 1 + DIR, ivy = 1 + (DIR + 1)%3, ivz = 1 + (DIR + 2)%3;
  const long s = (DIR == 0) ? 1 : (DIR == 1 ? (long)g.N1 : (long)g.N1*g.N2);
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  // wave-uniform bases (m comes from blockIdx) + one 32-bit in-variable offset per lane
  const double *q = w0 + (size_t)m*g.nvar*cs;
  const unsigned off = (((unsigned)k*(unsigned)g.N2 + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;   // bytes
  double ld, lx, ly, lz, le, rd, rx, ry, rz, re;
  // every stencil (and the face field) requested before the first reconstruction: the limiters branch, so with the
  // loads inside face_states_u each variable waited for its own (see sweep_x1_shared)
  constexpr int W = stencil_w<RECON>(), LO = stencil_lo<RECON>();
  constexpr int NVs = MHD ? 7 : 5;
  double sq[NVs][W];
  const double *bq = MHD ? bcc0 + (size_t)m*3*cs : nullptr;
  const double *vb[7] = {q, q + ivx*cs, q + ivy*cs, q + ivz*cs, q + 4*cs,
                         MHD ? bq + ((DIR + 1)%3)*cs : nullptr, MHD ? bq + ((DIR + 2)%3)*cs : nullptr};
#pragma unroll
  for (int n = 0; n < NVs; ++n) {
    if (rs_iso<RS>() && n == 4) continue;
#pragma unroll
    for (int d = 0; d < W; ++d) sq[n][d] = ldu(vb[n] + (long)(d - LO)*s, off);
  }
  [[maybe_unused]] double bxi_g = 0.0;
  if constexpr (MHD) bxi_g = ldu(bxf + (size_t)m*f3*f2*f1,
                                 (((unsigned)k*(unsigned)f2 + (unsigned)j)*(unsigned)f1 + (unsigned)i)*8u);
  __builtin_amdgcn_sched_barrier(0);
  face_states_v<RECON, 1>(sq[0], eos, ld, rd);
  face_states_v<RECON, 0>(sq[1], eos, lx, rx);
  face_states_v<RECON, 0>(sq[2], eos, ly, ry);
  face_states_v<RECON, 0>(sq[3], eos, lz, rz);
  if constexpr (rs_iso<RS>()) { le = re = 0.0; }           // isothermal: no energy variable
  else face_states_v<RECON, 2>(sq[4], eos, le, re);
  if constexpr (MHD) {
    double lby, lbz, rby, rbz;
    face_states_v<RECON, 0>(sq[5], eos, lby, rby);
    face_states_v<RECON, 0>(sq[6], eos, lbz, rbz);
    const double bxi = bxi_g;
    Cons1D fl = riemann_mhd_e<RS>(eos, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
    fd = fl.d; fx = fl.mx; fy = fl.my; fz = fl.mz; fe = fl.e; fby = fl.by; fbz = fl.bz;
  } else {
    riemann_hyd_e<RS>(eos, ld, lx, ly, lz, le, rd, rx, ry, rz, re, fd, fx, fy, fz, fe);
    fby = fbz = 0.0;
  }
}

// Window sizes of the marching kernels per reconstruction (cells kept per variable)
template <int RECON> struct RollCfg;
template <> struct RollCfg<0> { static constexpr int NW = 1; };   // dc
template <> struct RollCfg<1> { static constexpr int NW = 2; };   // plm
template <> struct RollCfg<2> { static constexpr int NW = 4; };   // ppm4
template <> struct RollCfg<3> { static constexpr int NW = 4; };   // ppmx
template <> struct RollCfg<4> { static constexpr int NW = 4; };   // wenoz
template <> struct RollCfg<5> { static constexpr int NW = 4; };   // teno

struct SweepArgs {
  const double *w0, *bcc0, *bxf;
  double *flx, *ey, *ez;          // this direction's flux (face-shaped) and face EMFs
  double *ecc1, *ecc2, *ecc3;     // cell-centred EMFs (x1 sweep only)
  int il, iu, jl, ju, kl, ku;     // face ranges of this sweep
  int f3, f2, f1;
};

// plain sweep (not the last direction): thread per face.  ECC: also emit e_cc for the right
// cell of every face and for the extra column i = il-1 (mhd_corner_e.cpp:309-317 range
// [is-1,ie+1] x [js-1,je+1] x [ks-1,ke+1] == the CT-extended x1 sweep, right cells).
// x1 sweep with PLM / the five-point schemes: the reconstruction of a cell serves both of its faces
// (ReconCellT, recon.hpp:40-118: cell c-1 writes wl(c), cell c writes wr(c)), so a lane reconstructs
// ITS cell only (PLM: one division per variable instead of two) and takes the left state of its face
// from the lane below through a wave shuffle.  Waves overlap by one lane (lane 0 only
// provides): 63 faces per wave.  Same operands, same operations -> same bits.
#ifndef AKMI_X1_SHARE
#define AKMI_X1_SHARE 1
#endif
template <int DIR, int RECON>
constexpr bool x1_share() { return DIR == 0 && RECON >= 1 && AKMI_X1_SHARE; }
// k_sweep: the planes [kl, ku] flattened into the lane index as well (120 blocks of 16^3, PPM4 + HLLD: x1 sweep 40.7 -> 32,
// x2 sweep 49.4 -> 36 us) -- profiles/r06_lane_mapping.txt

template <int RECON, bool MHD, bool ECC, int RS>
__device__ __forceinline__ void sweep_x1_shared(const Geo &g, const FaceEos &eos, const SweepArgs &a,
                                                int nk) {
  // lanes over the flattened (row, column) with the columns a row needs: cells il-1 .. iu (the first one only provides
  // the left state of face il) -- not the N1 of the row: 34 of the 40 columns of a 32^3 MeshBlock with four ghost cells
  // The flattening runs over the planes [kl, ku] of the MeshBlock as well (a plane of a 16^3 block is 18 x 18 positions:
  // 1.3 workgroups of 252, i.e. two workgroups 64 % full when every plane starts a workgroup of its own).
  const long p = ((long)blockIdx.x*SY + threadIdx.y)*(SX - 1) + (long)threadIdx.x - 1;
  const unsigned pc = p < 0 ? 0u : (unsigned)p;
  const unsigned row_w = (unsigned)(a.iu - a.il + 2), plane = row_w*(unsigned)(a.ju - a.jl + 1);
  const unsigned kk = pc/plane, pr = pc - kk*plane;
  const unsigned jj = pr/row_w;
  const int i = a.il - 1 + (int)(pr - jj*row_w);
  const int j = a.jl + (int)jj;
  const int m = blockIdx.z;
  const int k = a.kl + (int)kk;
  // a lane owns cell i: its stencil (i-1..i+1, five-point schemes i-2..i+2) has to be inside the row
  constexpr int HW = RECON == 1 ? 1 : 2;
  const bool valid = p >= 0 && (int)kk < nk && i >= HW && i <= g.N1 - 1 - HW;
  constexpr int NV = MHD ? 7 : 5;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  // addresses = wave-uniform base (block m, variable n: scalar unit) + ONE 32-bit byte offset per lane
  // (global_load v, v_off, s[base]): no 64-bit integer multiplies on the vector unit.  A block-variable
  // is below 4 GB (checked at launch).
  const unsigned oc = (((unsigned)k*(unsigned)g.N2 + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;           // cell (k,j,i)
  const unsigned of = (((unsigned)k*(unsigned)a.f2 + (unsigned)j)*(unsigned)a.f1 + (unsigned)i)*8u;   // face (k,j,i)
  const double *wm = a.w0 + (size_t)m*g.nvar*cs;
  const double *bm = MHD ? a.bcc0 + (size_t)m*3*cs : nullptr;
  double qln[NV], qr[NV];                 // left state of face i+1, right state of face i
#pragma unroll
  for (int n = 0; n < NV; ++n) { qln[n] = 0.0; qr[n] = 0.0; }
  double vx = 0.0, vy = 0.0, vz = 0.0, by = 0.0, bz = 0.0;
  [[maybe_unused]] double bxi_pre = 0.0;
  if (valid) {
    // the five-point reconstructions branch (limiters), so every variable is a basic block of its own and its
    // stencil loads were issued there: load -> s_waitcnt vmcnt(0) -> 43 VALU, seven times over
    // (profiles/r03_isa_audit.txt, second audit).  All stencils are requested first; the waits count down.
    double sq[NV][RECON == 1 ? 3 : 5];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (rs_iso<RS>() && n == 4) continue;
      const double *q = (n < 5) ? wm + n*cs : bm + (n - 4)*cs;
#pragma unroll
      for (int d = 0; d < (RECON == 1 ? 3 : 5); ++d) sq[n][d] = ldu(q + d - (RECON == 1 ? 1 : 2), oc);
    }
    if constexpr (MHD) bxi_pre = ldu(a.bxf + (size_t)m*((size_t)a.f3*a.f2*a.f1), of);   // face (k,j,i) exists for a valid lane
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (rs_iso<RS>() && n == 4) continue;             // isothermal: slot 4 (energy) stays unused
      const double *q = (n < 5) ? wm + n*cs : bm + (n - 4)*cs;   // by, bz
      constexpr int C = RECON == 1 ? 1 : 2;
      const double qm = sq[n][C - 1], q0 = sq[n][C], qp = sq[n][C + 1];
      if constexpr (RECON == 1) {
        plm(qm, q0, qp, qln[n], qr[n]);
      } else {
        const double qmm = sq[n][0], qpp = sq[n][4];
        recon5<RECON>(qmm, qm, q0, qp, qpp, qln[n], qr[n]);
        // the floors of recon.hpp:72-103 act on each state separately: per cell == per face
        if (n == 0) floor_lr<RECON, 1>(eos, qln[n], qr[n]);
        if (n == 4) floor_lr<RECON, 2>(eos, qln[n], qr[n]);
      }
      if (n == 1) vx = q0;
      if (n == 2) vy = q0;
      if (n == 3) vz = q0;
      if (n == 5) by = q0;
      if (n == 6) bz = q0;
    }
  }
  double L[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) L[n] = __shfl_up(qln[n], 1, 64);
  if (!valid) return;
  if constexpr (ECC) {
    if (i >= a.il - 1 && i <= a.iu) {
      const double bx = ldu(bm, oc);
      stu(a.ecc1 + (size_t)m*cs, oc, vz*by - vy*bz);
      stu(a.ecc2 + (size_t)m*cs, oc, vx*bz - vz*bx);
      stu(a.ecc3 + (size_t)m*cs, oc, vy*bx - vx*by);
    }
  }
  if (threadIdx.x == 0 || i < a.il || i > a.iu) return;     // lane 0 only provides
  double fd, fx, fy, fz, fe, fby = 0.0, fbz = 0.0;
  const size_t fs = (size_t)a.f3*a.f2*a.f1;
  if constexpr (MHD) {
    const double bxi = bxi_pre;
    Cons1D fl = riemann_mhd_e<RS, true>(eos, L[0], L[1], L[2], L[3], L[4], L[5], L[6], qr[0], qr[1],
                                        qr[2], qr[3], qr[4], qr[5], qr[6], bxi);
    fd = fl.d; fx = fl.mx; fy = fl.my; fz = fl.mz; fe = fl.e; fby = fl.by; fbz = fl.bz;
  } else {
    riemann_hyd_e<RS>(eos, L[0], L[1], L[2], L[3], L[4], qr[0], qr[1], qr[2], qr[3], qr[4], fd,
                      fx, fy, fz, fe);
  }
  double *f = a.flx + (size_t)m*g.nvar*fs;
  stu(f, of, fd); stu(f + fs, of, fx); stu(f + 2*fs, of, fy); stu(f + 3*fs, of, fz);
  if constexpr (!rs_iso<RS>()) stu(f + 4*fs, of, fe);
  if constexpr (MHD) {
    stu(a.ey + (size_t)m*cs, oc, -fby);
    stu(a.ez + (size_t)m*cs, oc, fbz);
  }
}

template <int DIR, int RECON, bool MHD, bool ECC, int RS>
__global__ void __launch_bounds__(SX*SY)
k_sweep(Geo g, FaceEos eos, SweepArgs a, int nk) {
  if constexpr (x1_share<DIR, RECON>()) {
    sweep_x1_shared<RECON, MHD, ECC, RS>(g, eos, a, nk);
    return;
  }
  // lanes run over the flattened rows [jl,ju] x [0,N1): contiguous in memory, and the few
  // ghost columns outside [il,iu] cost 2-4 idle lanes per 260 instead of a mostly empty wave
  // lanes over the flattened (row, column) with the columns of the sweep only (ECC: one more on the low side), not the
  // N1 of a row: a thread-per-face sweep has no coupling between lanes (18 of 24 columns at 16^3 with four ghost cells)
  // ... and over the planes [kl, ku] of the MeshBlock (a plane of a 16^3 block is 17 x 18 faces: two workgroups 60 % full
  // when every plane starts a workgroup of its own)
  const unsigned p = (blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;
  const int row_0 = a.il - (ECC ? 1 : 0);
  const unsigned row_w = (unsigned)(a.iu - row_0 + 1), plane = row_w*(unsigned)(a.ju - a.jl + 1);
  unsigned kk, jj, pr;
  if (DIR != 2) {
    kk = p/plane; pr = p - kk*plane;
    jj = pr/row_w; pr -= jj*row_w;
  } else {
    // x3 sweep: (j; k; i) -- the lanes of a wave that are not neighbours in i are neighbours in k, the direction of the
    // stencil, so the rows a wave loads overlap as they do in the x2 sweep (a wave of the (k; j; i) order reads six planes x
    // its own rows and shares nothing: 57 us against 37 for the x2 sweep on 120 blocks of 16^3)
    const unsigned slab = row_w*(unsigned)nk;
    jj = p/slab; pr = p - jj*slab;
    kk = pr/row_w; pr -= kk*row_w;
  }
  const int i = row_0 + (int)pr;
  const int j = a.jl + (int)jj;
  const int m = blockIdx.z;
  const int k = a.kl + (int)kk;
  if ((int)kk >= nk || j > a.ju) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const unsigned oc = (((unsigned)k*(unsigned)g.N2 + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;   // cell (k,j,i)
  if constexpr (ECC) {
    const double *wm = a.w0 + (size_t)m*g.nvar*cs, *bm = a.bcc0 + (size_t)m*3*cs;
    const double vx = ldu(wm + cs, oc), vy = ldu(wm + 2*cs, oc), vz = ldu(wm + 3*cs, oc);
    const double bx = ldu(bm, oc), by = ldu(bm + cs, oc), bz = ldu(bm + 2*cs, oc);
    stu(a.ecc1 + (size_t)m*cs, oc, vz*by - vy*bz);
    stu(a.ecc2 + (size_t)m*cs, oc, vx*bz - vz*bx);
    stu(a.ecc3 + (size_t)m*cs, oc, vy*bx - vx*by);
    if (i < a.il) return;
  }
  constexpr int ivx = 1 + DIR, ivy = 1 + (DIR + 1)%3, ivz = 1 + (DIR + 2)%3;
  double fd, fx, fy, fz, fe, fby, fbz;
  face_flux<DIR, RECON, MHD, RS>(g, eos, a.w0, a.bcc0, a.bxf, a.f3, a.f2, a.f1, m, k, j, i, fd,
                                 fx, fy, fz, fe, fby, fbz);
  const size_t fs = (size_t)a.f3*a.f2*a.f1;
  const unsigned of = (((unsigned)k*(unsigned)a.f2 + (unsigned)j)*(unsigned)a.f1 + (unsigned)i)*8u;       // face (k,j,i)
  double *f = a.flx + (size_t)m*g.nvar*fs;
  stu(f, of, fd); stu(f + ivx*fs, of, fx); stu(f + ivy*fs, of, fy); stu(f + ivz*fs, of, fz);
  if constexpr (!rs_iso<RS>()) stu(f + 4*fs, of, fe);
  if constexpr (MHD) {
    stu(a.ey + (size_t)m*cs, oc, -fby);
    stu(a.ez + (size_t)m*cs, oc, fbz);
  }
}

#ifndef AKMI_POW2DX
#define AKMI_POW2DX 1           // cell sizes that are powers of two: x/dx as one v_ldexp_f64 (wave-uniform choice at run time)
#endif
struct UpdArgs {
  double gam0, gam1, beta_dt;
  double *u0, *u1;
  const double *flx1, *flx2;      // fluxes of the earlier sweeps (face-shaped)
  int copy_u1;
  double *acc;                    // partial divergence (written by the x2 march, read by x3)
  const double *dtp;              // non-null: beta_dt holds the RK weight beta and dt is read from
                                  // device memory (a captured cycle replayed with a new time step)
};
// copy_u1 / copy_b1: 0 = the second register (u1, b1) holds the state of the start of the cycle;
// 1 = first stage, CopyCons folded in: the register receives the old state, u0 / b0 the new one;
// 2 = first stage OUT OF PLACE: u0 / b0 are only read, the new state goes to u1 / b1 and the caller
//     swaps the two registers afterwards -- the copy (5 + 3 arrays written) disappears.  Only the cells /
//     faces the update touches are written: the ghost zones of the new register are filled by the halo
//     exchange and the boundary conditions that follow every stage.
__device__ __forceinline__ void rk_store(double *__restrict__ u0, double *__restrict__ u1, int copy, size_t c,
                                         double old, double res) {
  if (copy == 2) { u1[c] = res; return; }
  if (copy) u1[c] = old;
  u0[c] = res;
}
// beta*dt: the product the host forms in RKUpdate (hydro_update.cpp:35), same operands, same rounding
__device__ __forceinline__ void rk_store_u(double *__restrict__ u0, double *__restrict__ u1, int copy, unsigned ob,
                                           double old, double res) {
  if (copy == 2) { stu(u1, ob, res); return; }
  if (copy) stu(u1, ob, old);
  stu(u0, ob, res);
}
__device__ __forceinline__ double beta_dt_of(double beta_dt, const double *dtp) {
  return dtp ? beta_dt*(*dtp) : beta_dt;
}

// last-direction sweep with the RK update fused, as a MARCH along the sweep direction:
// each thread owns one transverse position (lanes run over the contiguous index, so every
// load/store of a wave is a coalesced row segment), walks ML faces along the sweep, keeps the
// previous face flux in registers and finishes cell (t-1) as soon as face t is known:
//   divf = dF1/dx1; divf += dF2/dx2; divf += dF3/dx3; u0 = gam0*u0 + gam1*u1 - beta_dt*divf
// (hydro_update.cpp:55-80 order).  No LDS, no barrier: waves run free, so the memory-bound
// update of one wave hides under the Riemann arithmetic of the others.  One face per chunk
// (1/ML) is computed twice.
#ifndef AKMI_ML
#define AKMI_ML 32
#endif
#ifndef AKMI_PREFETCH_UPD
#define AKMI_PREFETCH_UPD 1
#endif
// The x3 march moves the most bytes of the stage and waits on memory 72 % of its wave cycles
// (profiles/r02_v3_valu_counters.txt).  With the register budget of TWO waves per SIMD it can hold the
// update operands u1 (AKMI_PREFETCH_U1) and the cells of the next step (AKMI_PREFETCH_W) in flight across
// the Riemann solve: 1165 -> 907 us at 256^3, 202 VGPRs, no scratch (profiles/r02_prefetch_w.txt).  At three
// waves the same prefetches spill (x3 march 1.07 -> 1.20-1.35 ms), and the x2 march loses at two waves.
#ifndef AKMI_PREFETCH_U1
#define AKMI_PREFETCH_U1 1
#endif
#ifndef AKMI_MARCH_WAVES
#define AKMI_MARCH_WAVES 3
#endif
#ifndef AKMI_PREFETCH_W
#define AKMI_PREFETCH_W 1       // x3 PLM march: load the cells of step t+1 during step t (2*NV more VGPRs)
#endif
#ifndef AKMI_PPM_WREG
#define AKMI_PPM_WREG 1         // marches with five-point reconstructions keep their window in registers
#endif
#ifndef AKMI_PREFETCH_W2
#define AKMI_PREFETCH_W2 0      // the same in the x2 march
#endif
#ifndef AKMI_X3_WAVES
#define AKMI_X3_WAVES 2         // register budget of the x3 march (waves per SIMD)
#endif
#ifndef AKMI_X2_WAVES
#define AKMI_X2_WAVES AKMI_MARCH_WAVES
#endif
#ifndef AKMI_PREFETCH_X1
#define AKMI_PREFETCH_X1 0
#endif
#ifndef AKMI_PREFETCH_WP2
#define AKMI_PREFETCH_WP2 1     // five-point schemes (two waves per SIMD): next cells prefetched in the x2 march,
#endif
#ifndef AKMI_PREFETCH_WP3
#define AKMI_PREFETCH_WP3 1     // ... in the x3 march,
#endif
#ifndef AKMI_PREFETCH_X1P
#define AKMI_PREFETCH_X1P 1     // ... the x1 flux difference fetched before the solve in the x2 march
#endif
#ifndef AKMI_DPP
#define AKMI_DPP 1            // neighbour lanes by DPP moves instead of ds_bpermute (lane_below / lane_above)
#endif
#if AKMI_DPP
#define AKMI_LANE_BELOW(x) lane_below(x)
#define AKMI_LANE_ABOVE(x) lane_above(x)
#else
#define AKMI_LANE_BELOW(x) __shfl_up(x, 1, 64)
#define AKMI_LANE_ABOVE(x) __shfl_down(x, 1, 64)
#endif
#ifndef AKMI_X12S_FM
#define AKMI_X12S_FM 1         // short sqrt / reciprocal forms (akmi_numerics.hpp) in the two solves of k_sweep12s
#endif
#ifndef AKMI_MARCH_FM
#define AKMI_MARCH_FM 0        // ... in the marches: loses (registers, basic blocks), profiles/r03_ab1.txt
#endif
#ifndef AKMI_SMALL_FACE_SWEEPS
#define AKMI_SMALL_FACE_SWEEPS 700000   // task path: packs up to this many cells take thread-per-face x2/x3 sweeps (0: never); the crossover
                                        // measured in round 3 (88^3 = 681 k cells still faster per face, profiles/r03_small_packs.txt); the HOSTS switch
                                        // small MHD packs to the task chain at AKMI_SMALL_PACK_CELLS (include/akmi.h)
#endif
#ifndef AKMI_X2_EO
#define AKMI_X2_EO 0            // wave-uniform early-outs of HLLD in the x2 / x3 march (registers!)
#endif
#ifndef AKMI_X3_EO
#define AKMI_X3_EO 0
#endif
constexpr int ML = AKMI_ML;            // faces marched per thread (chunk length), full-size packs

// chunk length of a marching kernel: ML when that still gives several workgroups per CU, shorter
// marches (more, smaller chunks; one face per chunk is recomputed) for small packs, which would
// otherwise leave most of the 256 CUs idle behind a few long serial chains.  Results do not
// depend on the chunking.
// Tile shape of k_corner_ct for e1 x e2 edge positions, among the shapes of at most 512 threads:
// lanes launched (tiles x padded workgroup size), weighted by what the scan in
// profiles/r01_v12_ct_tiles.txt shows -- short rows cost coalescing (~ 1 + 20/tw per lane), few
// rows re-read the j-1 neighbours more often (~ 1 + 0.6/(th-1)), and a workgroup of 7 waves leaves
// 2 of the 16 wave slots of a CU empty (106 VGPRs: 4 waves per SIMD).
constexpr int CT_THREADS = 512;        // 8 waves: two workgroups per CU hide each other's barriers
struct CtTile { int tw, th, n1, n2, threads; };
static CtTile ct_tile(int e1, int e2) {
  static int f_tw = -1, f_th = 0;
  if (f_tw < 0) {                                    // AKMI_CT_TILE=tw,th pins the shape (experiments)
    const char *e = getenv("AKMI_CT_TILE");
    f_tw = 0;
    if (e && sscanf(e, "%d,%d", &f_tw, &f_th) != 2) f_tw = 0;
    if (f_tw < 2 || f_th < 2 || f_tw*f_th > CT_THREADS) f_tw = 0;
  }
  CtTile best{0, 0, 0, 0, 0};
  double best_cost = -1.0;
  for (int n1 = 1; n1 <= e1; ++n1) {
    const int tw = (e1 + n1 - 1)/n1 + 1;
    if (tw > 130) continue;
    if (tw < 18 && n1 > 1) break;
    for (int th = 3; th <= 32; ++th) {
      if (tw*th > CT_THREADS) break;
      if (f_tw > 0 && (tw != f_tw || th != f_th)) continue;
      const int n2 = (e2 + th - 2)/(th - 1);
      const int threads = (tw*th + 63)/64*64;
      const int waves = threads/64;
      const double cost = (double)n1*n2*threads*(6.3 + 130.0/tw)*(1.0 + 0.6/(th - 1))*16.0/(16/waves*waves);
      if (best_cost < 0 || cost < best_cost) { best = CtTile{tw, th, n1, n2, threads}; best_cost = cost; }
    }
  }
  if (best_cost < 0) {                               // pinned shape not among the candidates
    const int tw = f_tw, th = f_th;
    best = CtTile{tw, th, (e1 + tw - 2)/(tw - 1), (e2 + th - 2)/(th - 1), (tw*th + 63)/64*64};
  }
  return best;
}

static int march_len(long col_blocks, int ncells, int nmb, int lmax, int wgs_per_cu = 0) {
  const long want = 2048;                       // workgroups per launch (scan: profiles/r01_small_packs.txt)
  long ml = col_blocks*(long)ncells*nmb/want;
  if (ml > lmax) ml = lmax;
  if (ml < 4) ml = 4;
  static const int tail = getenv("AKMI_TAIL") ? atoi(getenv("AKMI_TAIL")) : 1;
  // tail == 1: the rule below for every launch that asks for it (round 3: with the residency the kernel really
  // has -- two workgroups per CU for the x3 march -- full-length marches gain too: 256^3 x3 march 965 -> 907-914 us
  // at 14 instead of 32 faces, 8 chunks x 263 workgroups being 4.1 rounds of 512; profiles/r03_ab2.txt);
  // tail == 2: the round-2 behaviour (short marches only)
  // (packs of many blocks keep the plain rule: with tens of rounds the fill of the last one does not matter and
  //  every extra chunk costs its priming -- 960 blocks of 32^3, PPM4: x2/x3 marches 1750 -> 2330 us with two
  //  chunks of 17 instead of 32 + 1, profiles/r03_config5.txt)
  const double rounds_plain = (double)(col_blocks*((ncells + ml - 1)/ml)*nmb)/(256.0*(wgs_per_cu > 0 ? wgs_per_cu : 1));
  if (wgs_per_cu > 0 && tail && (ml < lmax || (tail == 1 && rounds_plain < 8.0))) {
    // equal-length workgroups run in rounds of (256 CUs x resident workgroups): pick the chunk
    // length whose last round is fullest, charging the face each chunk recomputes.  Used by the
    // x2/x3 marches of small packs (128^3: -7 % per march); full-length marches, k_corner_ct and
    // k_hydro_stage3d measured no better than the plain rule (profiles/r01_v14_tail_ab.txt)
    const double resident = 256.0*wgs_per_cu;
    double best = -1.0;
    long best_ml = ml;
    for (long c = (ml < lmax ? lmax : lmax + lmax/4); c >= 4; --c) {
      if (ml < lmax && c > lmax) continue;
      const long nch = (ncells + c - 1)/c;
      const double rounds = (double)(col_blocks*nch*nmb)/resident;
      const double full = rounds <= 1.0 ? 1.0 : rounds/(double)(long)(rounds + 0.999999);
      // below one round the launch is latency-bound: prefer more, shorter chunks up to a round
      const double fill = rounds < 1.0 ? rounds : 1.0;
      const double score = full*fill/(1.0 + 1.0/(double)c);
      if (score > best) { best = score; best_ml = c; }
    }
    ml = best_ml;
  }
  return (int)ml;
}

// MODE 2: no update at all, the five flux components of every face are stored (the flux kernels of the
// task-granular entry points, sweeps_store_fluxes below).
// MODE 0: last direction -- finish the RK update.  MODE 1 (x2 sweep of 3-D runs): store the
// partial divergence acc = dF1/dx1 + dF2/dx2 for the x3 march, which then needs one array
// instead of two face pairs per variable (USEACC).  Rounding sequence unchanged.
template <int DIR, int RECON, bool MHD, int MODE, bool USEACC, int RS, bool P2>
__device__ __forceinline__ void sweep_update_body(const Geo &g, const FaceEos &eos, const SweepArgs &a,
                                                  const UpdArgs &u, int ml, double *sm) {
  static_assert(DIR == 1 || DIR == 2, "marching kernel is for the x2/x3 sweeps");
  constexpr bool STORE = MODE == 2;                                  // all flux components of every face go to memory
  int i, j, k, m, s0;
  bool lane_ok;
  // lanes over the flattened (row, i).  The storing marches (MODE 2: refined meshes, small MeshBlocks) run them over the
  // columns of the sweep only -- [il, iu] instead of the N1 of the row: with four ghost cells a 32^3 block has 40 columns
  // of which 34 carry a face, and nothing couples the lanes of a march.  The other modes keep the N1 rows (at 256^3
  // 258 of 260 lanes carry a face).
  const int row_w = STORE ? a.iu - a.il + 1 : g.N1, row_0 = STORE ? a.il : 0;
  if constexpr (DIR == 2) {
    const long p = ((long)blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;   // rows [jl,ju] x row_w
    const int jj = (int)(p/row_w);
    i = row_0 + (int)(p - (long)jj*row_w);
    j = a.jl + jj;
    m = blockIdx.z;
    s0 = a.kl + blockIdx.y*ml;
    k = s0;
    lane_ok = (j <= a.ju) && (i >= a.il) && (i <= a.iu);
  } else {
    const long p = ((long)blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;   // planes [kl,ku] x row_w
    const int kk = (int)(p/row_w);
    i = row_0 + (int)(p - (long)kk*row_w);
    k = a.kl + kk;
    m = blockIdx.z;
    s0 = a.jl + blockIdx.y*ml;
    j = s0;
    lane_ok = (k <= a.ku) && (i >= a.il) && (i <= a.iu);
  }
  if (!lane_ok) return;
  constexpr int ivx = 1 + DIR, ivy = 1 + (DIR + 1)%3, ivz = 1 + (DIR + 2)%3;
  constexpr int iby = (DIR + 1)%3, ibz = (DIR + 2)%3;
  constexpr int NV = MHD ? 7 : 5;
  constexpr bool ISO = rs_iso<RS>();       // slot 4 (energy) of the variable arrays unused
  constexpr int NW = RollCfg<RECON>::NW;
  constexpr int NT = SX*SY;
  // marching state of this thread, parked in LDS ("LDS as an extension of the register file":
  // slot s of thread t lives at sm[s*NT + t], consecutive lanes -> consecutive banks):
  //   W(n,c)  last NW cells of variable n       PL(n)  pending left state of the next face
  //   FP(n)   flux of the previous face
  // Keeping it out of the VGPRs lets the kernel run at the occupancy of the plain Riemann
  // kernel while every cell is loaded from HBM exactly once per sweep.
  double *my = sm + threadIdx.y*SX + threadIdx.x;
  // five-point reconstructions (NW = 4) run with the register budget of two waves per SIMD: their
  // window stays in registers (WREG) and only PL/FP are parked -- 12 instead of 40 LDS slots per thread,
  // 24 KB instead of 80 KB per workgroup, i.e. two workgroups per CU instead of one
  constexpr bool WREG = (RECON >= 2) && AKMI_PPM_WREG;
  constexpr int WB = WREG ? 0 : NV*NW;               // LDS slots taken by the window
  double Wr[WREG ? NV : 1][WREG ? NW : 1];
#define W_(n, c) (*(WREG ? &Wr[WREG ? (n) : 0][WREG ? (c) : 0] : &my[((n)*NW + (c))*NT]))
#define PL_(n) my[(WB + (n))*NT]
#define FP_(n) my[(WB + NV + (n))*NT]
  const int shi = (DIR == 1) ? a.ju : a.ku;           // last face along the sweep
  // beta*dt once per thread, in scalar registers (inside the update loop a device-resident dt was re-read,
  // and waited for, once per variable and step)
  const double bdt = (MODE == 0) ? to_sgpr(beta_dt_of(u.beta_dt, u.dtp)) : 0.0;
  const bool col_active = (i >= g.is) && (i <= g.ie) &&
                          ((DIR == 1) ? (k >= g.ks && k <= g.ke) : (j >= g.js && j <= g.je));
  const int clo = (DIR == 1) ? g.js : g.ks, chi = (DIR == 1) ? g.je : g.ke;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const bool p2 = P2 && is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);      // wave-uniform: x/dx == ldexp(x, n) bit for bit
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const long st = (DIR == 1) ? (long)g.N1 : (long)g.N1*g.N2;
  // wave-uniform variable bases in sweep-aligned order d, vx, vy, vz, e, (by, bz) + one
  // per-lane element offset that advances by st per step
  // (addresses = uniform base on the scalar unit + 32-bit byte offset of the lane: global_load v, v_off,
  // s[base]; a block-variable is below 4 GB, checked at launch)
  const double *wb = a.w0 + (size_t)m*g.nvar*cs;
  const double *bb = MHD ? a.bcc0 + (size_t)m*3*cs : nullptr;
  unsigned off = (((unsigned)k*(unsigned)g.N2 + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;   // cell s
  const unsigned st8 = (unsigned)st*8u;
  auto base = [&](int n) -> const double * {
    return n == 0 ? wb : n == 1 ? wb + ivx*cs : n == 2 ? wb + ivy*cs : n == 3 ? wb + ivz*cs
         : n == 4 ? wb + 4*cs : n == 5 ? bb + iby*cs : bb + ibz*cs;
  };
  // prime the window: left state of the first face comes from cell s0-1
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    if (ISO && n == 4) continue;
    const double *q = base(n);
    double pl, dummy;
    if constexpr (RECON == 1) {
      const double qa = ldu(q - 2*st, off), qb = ldu(q - st, off), qc = ldu(q, off);
      plm(qa, qb, qc, pl, dummy);
      W_(n, 0) = qb; W_(n, 1) = qc;
    } else if constexpr (RECON >= 2) {
      const double qa = ldu(q - 3*st, off), qb = ldu(q - 2*st, off), qc = ldu(q - st, off), qd = ldu(q, off),
                   qe = ldu(q + st, off);
      recon5<RECON>(qa, qb, qc, qd, qe, pl, dummy);
      if (n == 0) floor_lr<RECON, 1>(eos, pl, dummy);
      if (n == 4) floor_lr<RECON, 2>(eos, pl, dummy);
      W_(n, 0) = qb; W_(n, 1) = qc; W_(n, 2) = qd; W_(n, 3) = qe;
    } else {
      pl = ldu(q - st, off);
      W_(n, 0) = ldu(q, off);
    }
    PL_(n) = pl;
  }
  // face (k,j,i) of this direction's face-shaped arrays (f3,f2,f1), advancing with the march
  const size_t fs = (size_t)a.f3*a.f2*a.f1;
  unsigned foff = (((unsigned)k*(unsigned)a.f2 + (unsigned)j)*(unsigned)a.f1 + (unsigned)i)*8u;
  const unsigned fst8 = ((DIR == 1) ? (unsigned)a.f1 : (unsigned)a.f1*(unsigned)a.f2)*8u;
  const double *bxm = MHD ? a.bxf + (size_t)m*fs : nullptr;
  double *mfm = a.flx ? a.flx + (size_t)m*g.nvar*fs : nullptr;       // variable 0: the mass flux
  // x1 fluxes (N3,N2,N1+1) of the cell row the step finishes (x2 march without acc)
  const size_t fs1 = (size_t)g.N3*g.N2*(g.N1 + 1);
  unsigned o1 = (((unsigned)k*(unsigned)g.N2 + (unsigned)j)*(unsigned)(g.N1 + 1) + (unsigned)i)*8u;
  const unsigned st18 = (unsigned)(g.N1 + 1)*8u;
  constexpr bool PW = (RECON == 1 && ((AKMI_PREFETCH_W && DIR == 2) || (AKMI_PREFETCH_W2 && DIR == 1))) ||
                      (RECON >= 2 && ((AKMI_PREFETCH_WP3 && DIR == 2) || (AKMI_PREFETCH_WP2 && DIR == 1)));
  constexpr int LA = RECON == 1 ? 1 : 2;            // the cell a step loads is LA cells ahead of cell s
  double nx[NV];                         // PW: cells s+1 of the coming step, loaded one step ahead
  // (the face field of the next face prefetched as well: measured, the x3 march loses 865 -> 942 us,
  //  profiles/r03_ab2.txt; requested before the update operands: no effect, profiles/r03_ab4.txt -- both removed)
  if constexpr (PW) {
#pragma unroll
    for (int n = 0; n < NV; ++n) nx[n] = (ISO && n == 4) ? 0.0 : ldu(base(n) + LA*st, off);
  }
  for (int t = 0; t <= ml; ++t) {
    const int s = s0 + t;
    if (s > shi) break;
    if constexpr (DIR == 1) j = s; else k = s;
    double nx2[NV];
    if constexpr (PW) {
      const bool more = (t < ml) && (s < shi);           // a next step exists: its cell s+2 is inside the array
#pragma unroll
      for (int n = 0; n < NV; ++n) nx2[n] = (more && !(ISO && n == 4)) ? ldu(base(n) + (LA + 1)*st, off) : 0.0;
    }
    double L[NV], R[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (ISO && n == 4) { L[n] = R[n] = 0.0; continue; }      // isothermal: no energy variable
      const double *q = base(n);
      double qln;
      L[n] = PL_(n);
      if constexpr (RECON == 1) {
        double qp;
        if constexpr (PW) qp = nx[n]; else qp = ldu(q + st, off);
        const double w0 = W_(n, 0), w1 = W_(n, 1);
        plm(w0, w1, qp, qln, R[n]);
        W_(n, 0) = w1; W_(n, 1) = qp;
      } else if constexpr (RECON >= 2) {
        double qp;
        if constexpr (PW) qp = nx[n]; else qp = ldu(q + 2*st, off);
        const double w0 = W_(n, 0), w1 = W_(n, 1), w2 = W_(n, 2), w3 = W_(n, 3);
        recon5<RECON>(w0, w1, w2, w3, qp, qln, R[n]);
        if (n == 0) floor_lr<RECON, 1>(eos, qln, R[n]);
        if (n == 4) floor_lr<RECON, 2>(eos, qln, R[n]);
        W_(n, 0) = w1; W_(n, 1) = w2; W_(n, 2) = w3; W_(n, 3) = qp;
      } else {
        const double w0 = W_(n, 0);
        R[n] = w0;
        qln = w0;
        W_(n, 0) = ldu(q + st, off);
      }
      PL_(n) = qln;
    }
    const unsigned oc = off;                            // cell s == face s of the cell-shaped EMF arrays
    const unsigned ocm = off - st8;                     // cell s-1, the one this face finishes
    off += st8;
    // x3 march: fetch the update operands of the cell this face finishes BEFORE the Riemann solve,
    // so that their latency is covered by ~1000 VALU instructions instead of following them
    // (this kernel moves the most bytes of the stage and runs at 3 waves/SIMD)
    constexpr bool PRE = (DIR == 2) && USEACC && (MODE == 0) && AKMI_PREFETCH_UPD;
    const int sc = s - 1;                               // cell finished by this face
    const bool upd = MODE != 2 && t > 0 && col_active && sc >= clo && sc <= chi;
    double pa[5], pu[5], pu1[5];
    if constexpr (PRE) {
      if (upd) {
        const size_t mb = (size_t)m*g.nvar*cs;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          if (ISO && n == 4) continue;
          pa[n] = ldu(u.acc + mb + n*cs, ocm); pu[n] = ldu(u.u0 + mb + n*cs, ocm);
        }
        if (AKMI_PREFETCH_U1 && !u.copy_u1) {
#pragma unroll
          for (int n = 0; n < 5; ++n) { if (ISO && n == 4) continue; pu1[n] = ldu(u.u1 + mb + n*cs, ocm); }
        }
      }
    }
    // x2 march of 3-D runs: same idea for the x1 flux difference of the finished cell
    constexpr bool PRE1 = (DIR == 1) && (MODE == 1) && !USEACC && (RECON >= 2 ? AKMI_PREFETCH_X1P : AKMI_PREFETCH_X1);
    if constexpr (PRE1) {
      if (upd) {
        const double *f1 = u.flx1 + (size_t)m*g.nvar*fs1 - (g.N1 + 1);       // row sc = s-1
        // the two fluxes stay as loaded until the update below: a subtraction or `p2 ? ldexp : /` here would put a
        // wait (and, through the wave-uniform branch, one basic block per variable: load -> vmcnt(0), five times)
        // in front of the solve that is meant to cover the latency (second audit, profiles/r03_isa_audit.txt)
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          if (ISO && n == 4) continue;
          pu[n] = ldu(f1 + n*fs1 + 1, o1); pa[n] = ldu(f1 + n*fs1, o1);
        }
      }
    }
    double fd, fx, fy, fz, fe;
    if constexpr (MHD) {
      const double bxi = ldu(bxm, foff);
      Cons1D fl = riemann_mhd_e<RS, (DIR == 1 ? AKMI_X2_EO : AKMI_X3_EO) != 0, AKMI_MARCH_FM != 0>(
          eos, L[0], L[1], L[2], L[3], L[4], L[5], L[6], R[0], R[1], R[2], R[3], R[4], R[5], R[6], bxi);
      fd = fl.d; fx = fl.mx; fy = fl.my; fz = fl.mz; fe = fl.e;
      if (t < ml || s == shi) {
        // CornerE needs the sign of the mass flux and the two face EMFs of this direction
        stu(mfm, foff, fd);
        stu(a.ey + (size_t)m*cs, oc, -fl.by);
        stu(a.ez + (size_t)m*cs, oc, fl.bz);
      }
    } else {
      riemann_hyd_e<RS>(eos, L[0], L[1], L[2], L[3], L[4], R[0], R[1], R[2], R[3], R[4], fd, fx,
                        fy, fz, fe);
      // passive scalars are advected by the mass flux (k_scalar_update): keep it
      if ((STORE || g.nvar > (ISO ? 4 : 5)) && (t < ml || s == shi)) stu(mfm, foff, fd);
    }
    double fv[5];
    fv[0] = fd; fv[ivx] = fx; fv[ivy] = fy; fv[ivz] = fz; fv[4] = fe;
    if constexpr (STORE) {                // task-granular callers: the whole flux of the face goes to memory
      if (t < ml || s == shi) {
#pragma unroll
        for (int n = 1; n < 5; ++n) { if (ISO && n == 4) continue; stu(mfm + n*fs, foff, fv[n]); }
      }
    }
    foff += fst8;
    if (upd) {
      const int kc = (DIR == 2) ? sc : k, jc = (DIR == 1) ? sc : j;
      const size_t mb = (size_t)m*g.nvar*cs;
      // operands of all variables first (one group of loads, counted waits), then the arithmetic with the
      // power-of-two choice hoisted out of the loop over the variables: with `p2 ? ldexp : /` inside that loop
      // every variable was a basic block of its own with its loads and an s_waitcnt vmcnt(0) at the top
      // (the x3 march with its operands prefetched before the solve keeps the plain loop: nothing is loaded in it, and
      //  the two-copy form costs it 870 -> 940 us, profiles/r03_ab5.txt)
      if constexpr (PRE) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          if (ISO && n == 4) continue;
          const double dlast = fv[n] - FP_(n);
          double divf = pa[n];
          divf += p2 ? ldexp(dlast, n3) : dlast/dx3;
          const double u0v = pu[n];
          double u1v;
          if constexpr (AKMI_PREFETCH_U1) u1v = u.copy_u1 ? u0v : pu1[n];
          else u1v = u.copy_u1 ? u0v : ldu(u.u1 + mb + n*cs, ocm);
          rk_store_u(u.u0 + mb + n*cs, u.u1 + mb + n*cs, u.copy_u1, ocm, u0v, u.gam0*u0v + u.gam1*u1v - bdt*divf);
        }
      } else {
      double t1[5], t2[5], u0v[5], u1v[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        if (ISO && n == 4) continue;
        t2[n] = 0.0;
        if constexpr (PRE1) {
          t1[n] = pu[n] - pa[n];                          // x1 flux difference, fetched before the solve
        } else if constexpr (PRE) {
          t1[n] = pa[n];                                  // acc, fetched before the solve
        } else if constexpr (USEACC) {
          t1[n] = ldu(u.acc + mb + n*cs, ocm);
        } else if constexpr (DIR == 1) {
          const double *f1 = u.flx1 + (size_t)m*g.nvar*fs1 + n*fs1 - (g.N1 + 1);     // row sc = s-1
          t1[n] = ldu(f1 + 1, o1) - ldu(f1, o1);
        } else {
          t1[n] = u.flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, n, kc, jc, i + 1)] -
                  u.flx1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, n, kc, jc, i)];
          t2[n] = u.flx2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, n, kc, jc + 1, i)] -
                  u.flx2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, n, kc, jc, i)];
        }
        if constexpr (MODE != 1) {
          if constexpr (PRE) u0v[n] = pu[n]; else u0v[n] = ldu(u.u0 + mb + n*cs, ocm);
          if constexpr (PRE && AKMI_PREFETCH_U1) u1v[n] = u.copy_u1 ? u0v[n] : pu1[n];
          else u1v[n] = u.copy_u1 ? u0v[n] : ldu(u.u1 + mb + n*cs, ocm);
        }
      }
      auto finish = [&](auto P2c) {
        constexpr bool P2v = decltype(P2c)::value;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          if (ISO && n == 4) continue;
          const double dlast = fv[n] - FP_(n);
          double divf;
          if constexpr ((PRE && !PRE1) || (USEACC && !PRE1)) divf = t1[n];           // acc: already a divergence
          else divf = P2v ? ldexp(t1[n], n1) : t1[n]/dx1;
          if constexpr (DIR == 1) {
            divf += P2v ? ldexp(dlast, n2) : dlast/dx2;
          } else {
            if constexpr (!USEACC) divf += P2v ? ldexp(t2[n], n2) : t2[n]/dx2;
            divf += P2v ? ldexp(dlast, n3) : dlast/dx3;
          }
          if constexpr (MODE == 1) {
            stu(u.acc + mb + n*cs, ocm, divf);
          } else {
            rk_store_u(u.u0 + mb + n*cs, u.u1 + mb + n*cs, u.copy_u1, ocm, u0v[n],
                       u.gam0*u0v[n] + u.gam1*u1v[n] - bdt*divf);
          }
        }
      };
      if (p2) finish(IC<1>{}); else finish(IC<0>{});
      }
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) { if (ISO && n == 4) continue; FP_(n) = fv[n]; }
    o1 += st18;
    if constexpr (PW) {
#pragma unroll
      for (int n = 0; n < NV; ++n) nx[n] = nx2[n];
    }
  }
#undef W_
#undef PL_
#undef FP_
}

// the kernel proper: when the cell sizes of the block are powers of two, x/dx == x*(1/dx) bit for
// bit (both are the correctly rounded value of the same real number, and 1/dx is exact), so the
// divisions by dx become products; decided per block, two copies of the loop
template <int DIR, int RECON, bool MHD, int MODE, bool USEACC, int RS>
__global__ void __launch_bounds__(SX*SY, (RECON >= 2 ? 2 : (DIR == 2 ? AKMI_X3_WAVES : AKMI_X2_WAVES)))
k_sweep_update(Geo g, FaceEos eos, SweepArgs a, UpdArgs u, int ml) {
  constexpr int NV = MHD ? 7 : 5;
  __shared__ double sm[(((RECON >= 2) && AKMI_PPM_WREG ? 0 : NV*RollCfg<RECON>::NW) + NV + 5)*SX*SY];
  const int m = blockIdx.z;
  constexpr bool TRY = AKMI_POW2DX != 0;      // power-of-two cell sizes: x/dx by v_ldexp_f64 (wave-uniform run-time choice)
  sweep_update_body<DIR, RECON, MHD, MODE, USEACC, RS, TRY>(g, eos, a, u, ml, sm);
}


// 1-D problems: the sweep direction is the lane direction, so neighbouring faces are
// exchanged through LDS inside a TX-wide tile (overlap of one face between tiles).
template <int RECON, bool MHD, int RS>
__global__ void __launch_bounds__(TX)
k_sweep_update_1d(Geo g, FaceEos eos, SweepArgs a, UpdArgs u) {
  __shared__ double sF[5][TX];
  const int i = a.il + blockIdx.x*(TX - 1) + threadIdx.x;
  const int j = a.jl, k = a.kl, m = blockIdx.z;
  const bool face_ok = (i <= a.iu);
  double fd = 0, fx = 0, fy = 0, fz = 0, fe = 0, fby = 0, fbz = 0;
  if (face_ok) {
    face_flux<0, RECON, MHD, RS>(g, eos, a.w0, a.bcc0, a.bxf, a.f3, a.f2, a.f1, m, k, j, i, fd, fx,
                                 fy, fz, fe, fby, fbz);
    if constexpr (MHD) {
      a.flx[ix5(g.nvar, a.f3, a.f2, a.f1, m, 0, k, j, i)] = fd;
      const size_t ec = ix4(g.N3, g.N2, g.N1, m, k, j, i);
      a.ey[ec] = -fby;
      a.ez[ec] = fbz;
    } else {
      if (g.nvar > (rs_iso<RS>() ? 4 : 5)) a.flx[ix5(g.nvar, a.f3, a.f2, a.f1, m, 0, k, j, i)] = fd;
    }
  }
  const double fv[5] = {fd, fx, fy, fz, fe};
#pragma unroll
  for (int n = 0; n < 5; ++n) sF[n][threadIdx.x] = fv[n];
  __syncthreads();
  if (threadIdx.x >= TX - 1 || i > g.ie) return;
  const double dx1 = g.dx[3*m];
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
#pragma unroll
  for (int n = 0; n < 5; ++n) {
    if (rs_iso<RS>() && n == 4) continue;            // isothermal: no energy variable
    const double divf = (sF[n][threadIdx.x + 1] - fv[n])/dx1;
    const double u0v = u.u0[c + n*cs];
    const double u1v = u.copy_u1 ? u0v : u.u1[c + n*cs];
    rk_store(u.u0, u.u1, u.copy_u1, c + n*cs, u0v, u.gam0*u0v + u.gam1*u1v - beta_dt_of(u.beta_dt, u.dtp)*divf);
  }
}

// ---------------------------------------------------------------------------------------
// Passive scalars of the fused stage (hydro_fluxes.cpp:135-147, mhd_fluxes.cpp:153-166 + RKUpdate): a
// scalar's flux is the mass flux times its upwind reconstructed value, so the Riemann kernels only
// have to leave the mass fluxes behind (the MHD sweeps store them anyway, for CornerE); this kernel
// reconstructs each scalar at the six faces of a cell, forms the divergence in the reference's order
// and applies the RK update, CopyCons folded in.  Same operations as the task-granular kernels.
template <int RECON>
__global__ void __launch_bounds__(SX*SY)
k_scalar_update(Geo g, FaceEos eos, const double *__restrict__ w0, const double *__restrict__ m1,
                const double *__restrict__ m2, const double *__restrict__ m3, UpdArgs u, int k0, int nk,
                int nf) {
  // nf: number of fluid variables (5 ideal gas, 4 isothermal); the scalars follow them
  const long p = ((long)blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;   // rows [js,je] x N1
  const int jj = (int)(p/g.N1);
  const int i = (int)(p - (long)jj*g.N1);
  const int j = g.js + jj;
  const int m = blockIdx.z/nk;
  const int k = k0 + (blockIdx.z - m*nk);
  if (i < g.is || i > g.ie || j > g.je) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const double bdt = beta_dt_of(u.beta_dt, u.dtp);
  const double f1l = m1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, 0, k, j, i)];
  const double f1h = m1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, 0, k, j, i + 1)];
  double f2l = 0.0, f2h = 0.0, f3l = 0.0, f3h = 0.0;
  if (g.multi_d) {
    f2l = m2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, 0, k, j, i)];
    f2h = m2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, 0, k, j + 1, i)];
  }
  if (g.three_d) {
    f3l = m3[ix5(g.nvar, g.N3 + 1, g.N2, g.N1, m, 0, k, j, i)];
    f3h = m3[ix5(g.nvar, g.N3 + 1, g.N2, g.N1, m, 0, k + 1, j, i)];
  }
  const long s2 = g.N1, s3 = (long)g.N1*g.N2;
  for (int n = nf; n < g.nvar; ++n) {
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, n, k, j, i);
    const double *q = w0 + c;
    double sl, sr;
    face_states<RECON, 0>(q, 1, eos, sl, sr);
    const double a_lo = f1l*((f1l >= 0.0) ? sl : sr);
    face_states<RECON, 0>(q + 1, 1, eos, sl, sr);
    const double a_hi = f1h*((f1h >= 0.0) ? sl : sr);
    double divf = (a_hi - a_lo)/dx1;
    if (g.multi_d) {
      face_states<RECON, 0>(q, s2, eos, sl, sr);
      const double b_lo = f2l*((f2l >= 0.0) ? sl : sr);
      face_states<RECON, 0>(q + s2, s2, eos, sl, sr);
      const double b_hi = f2h*((f2h >= 0.0) ? sl : sr);
      divf += (b_hi - b_lo)/dx2;
    }
    if (g.three_d) {
      face_states<RECON, 0>(q, s3, eos, sl, sr);
      const double c_lo = f3l*((f3l >= 0.0) ? sl : sr);
      face_states<RECON, 0>(q + s3, s3, eos, sl, sr);
      const double c_hi = f3h*((f3h >= 0.0) ? sl : sr);
      divf += (c_hi - c_lo)/dx3;
    }
    const double u0v = u.u0[c];
    const double u1v = u.copy_u1 ? u0v : u.u1[c];
    rk_store(u.u0, u.u1, u.copy_u1, c, u0v, u.gam0*u0v + u.gam1*u1v - bdt*divf);
  }
}

// ---------------------------------------------------------------------------------------
// Corner EMFs (mhd_corner_e.cpp:338-414) from face EMFs, cell-centred EMFs and mass-flux
// signs.  All operands are loaded unconditionally and chosen with selects.
#define CCE(a, k, j, i) a[ix4(g.N3, g.N2, g.N1, m, k, j, i)]
__device__ __forceinline__ double upw(bool pos, double fa, double ca, double fb, double cb) {
  return pos ? (fa - ca) : (fb - cb);
}

// ---------------------------------------------------------------------------------------
// CornerE + CT in ONE kernel (3-D).  The corner EMFs never go to memory: a workgroup owns a
// (j,i) tile with a one-column/one-row overlap, marches along k, exchanges the three edge values
// of a plane through LDS (i+1 and j+1 neighbours) and keeps its own previous plane in registers
// (k+1 differences), so that
//     x3-faces of plane k         are updated at step k      (need e1,e2 of plane k),
//     x1-/x2-faces of plane k-1   are updated at step k      (need e3 of plane k-1, e1/e2 of both).
// The k-1 operands of the corner formulas are the previous step's k operands (25 instead of 33
// loads per corner).  Every face is read and written by exactly one thread (the overlap
// column/row only computes edges), so the in-place update of b0 has no cross-workgroup hazard.
// Arithmetic: the expressions of akmi_mhd_corner_e (akmi_tasks.hip) and k_ct_copy, unchanged.
#ifndef AKMI_CKL
#define AKMI_CKL 32
#endif
#ifndef AKMI_CT_XCD_ROWS
#define AKMI_CT_XCD_ROWS 1
#endif
#ifndef AKMI_C2P_PAIRS
#define AKMI_C2P_PAIRS 1
#endif
#ifndef AKMI_C2P_PAIRS_MIN
#define AKMI_C2P_PAIRS_MIN 4000000l
#endif
constexpr int CKL = AKMI_CKL;          // cell planes per k-chunk (one plane of edges recomputed)

// The tile of edge positions is tw x th threads (owners: (tw-1) x (th-1)), lanes flattened over
// (ty, tx); the launcher picks the shape that wastes the fewest lanes for the block size (a 64-wide
// tile needs two columns of tiles for the 65 edge columns of a 64^3 MeshBlock, a 34 x 15 tile does
// 33 / 65 / 257 columns in 1 / 2 / 8).
// Workgroup -> tile order.  Workgroup b of a launch runs on XCD b % 8 (observed placement; nothing below depends on
// it for correctness), and each XCD has its own L2.  A tile kernel whose neighbouring tiles re-read each other's edge
// rows wants neighbours on the SAME XCD at about the same time: XCD x takes the x-th contiguous eighth of the tile
// list, in list order.  The map is a bijection of [0, n) for every n.
__device__ __forceinline__ unsigned xcd_order(unsigned b, unsigned n) {
  const unsigned xcd = b & 7u, slot = b >> 3, q = n >> 3, rem = n & 7u;
  return (xcd < rem ? xcd*(q + 1u) : rem*(q + 1u) + (xcd - rem)*q) + slot;
}

template <bool P2>
__device__ __forceinline__ void corner_ct_body(const Geo &g, const double *__restrict__ e3x1, const double *__restrict__ e2x1,
            const double *__restrict__ e1x2, const double *__restrict__ e3x2,
            const double *__restrict__ e2x3, const double *__restrict__ e1x3,
            const double *__restrict__ c1, const double *__restrict__ c2,
            const double *__restrict__ c3, const double *__restrict__ flx1,
            const double *__restrict__ flx2, const double *__restrict__ flx3, double gam0,
            double gam1, double beta_dt, double *__restrict__ b0x1f, double *__restrict__ b0x2f,
            double *__restrict__ b0x3f, double *__restrict__ b1x1f, double *__restrict__ b1x2f,
            double *__restrict__ b1x3f, int copy_b1, int kA, int kB, int top, int nchunk,
            int ckl, int tw, int th, const double *dtp) {
  beta_dt = beta_dt_of(beta_dt, dtp);
  extern __shared__ double ct_lds[];     // e1, e2: 2 planes each, e3: 3 planes of th x tw
  const int plane = tw*th;
#define S1(p, y, x) ct_lds[(p)*plane + (y)*tw + (x)]
#define S2(p, y, x) ct_lds[(2 + (p))*plane + (y)*tw + (x)]
#define S3(p, y, x) ct_lds[(4 + (p))*plane + (y)*tw + (x)]
  const int ty = threadIdx.x/tw, tx = threadIdx.x - ty*tw;
  const bool in_tile = ty < th;          // the workgroup is padded to whole waves
  // (xcd_order, which pays in the hydro tile kernel, costs this bandwidth-bound one 2 %: 659 -> 675 us, profiles/r05_mhd_ab.txt)
  // Workgroup -> tile: the tiles of one tile ROW (same j range, same k-chunk: neighbours in x1, whose edge columns and
  // unaligned row ends share 64-/128-byte pieces of every operand row) go to ONE XCD, consecutive rows to consecutive
  // XCDs, so that all eight L2s stream through the same planes at the same time: 625 -> 617 us at 256^3
  // (profiles/r05_corner_ct_xcd_rows.txt; a whole k-chunk per XCD, xcd_order, costs 2 %).
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (AKMI_CT_XCD_ROWS) {
    const unsigned n1 = gridDim.x, nrows = gridDim.y*gridDim.z, full = nrows & ~7u;
    const unsigned b = bx + n1*(by + gridDim.y*bz);
    if (b < n1*full) {                                           // whole groups of eight rows; the rest keeps its place
      const unsigned xcd = b & 7u, slot = b >> 3;
      const unsigned rl = slot/n1;
      bx = slot - rl*n1;
      const unsigned R = rl*8u + xcd;
      bz = R/gridDim.y; by = R - bz*gridDim.y;
    }
  }
  const int i = g.is + bx*(tw - 1) + tx;
  const int j = g.js + by*(th - 1) + ty;
  const int m = bz/nchunk;
  const int ch = bz - m*nchunk;
  const int k0 = kA + ch*ckl;                                   // first cell plane of this chunk
  const int k1 = (k0 + ckl - 1 < kB) ? k0 + ckl - 1 : kB;       // last cell plane
  const bool wtop = top && (k1 == kB);                          // this chunk owns the x3-faces kB+1
  const bool edge_ok = in_tile && (i <= g.ie + 1) && (j <= g.je + 1);
  const bool own = edge_ok && (tx < tw - 1) && (ty < th - 1);
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  // cell sizes that are powers of two: x/dx == x*(1/dx) bit for bit (wave-uniform choice)
  const bool p2 = P2 && is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);
  const int ndx1 = pow2_shift(dx1), ndx2 = pow2_shift(dx2), ndx3 = pow2_shift(dx3);
#define DIVX(x, q) (p2 ? ldexp((x), n##q) : (x)/q)
  // addresses: scalar base of (block, array) + a 32-bit byte offset per lane and array shape, advanced by one
  // plane per step (no 64-bit index arithmetic on the vector unit, fewer address registers)
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const long PS = (long)g.N2*g.N1, PS1 = (long)g.N2*(g.N1 + 1), PS2 = (long)(g.N2 + 1)*g.N1;   // plane strides
  unsigned oc = (((unsigned)k0*(unsigned)g.N2 + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;            // (.., N2, N1)
  unsigned o1 = (((unsigned)k0*(unsigned)g.N2 + (unsigned)j)*(unsigned)(g.N1 + 1) + (unsigned)i)*8u;      // (.., N2, N1+1)
  unsigned o2 = (((unsigned)k0*(unsigned)(g.N2 + 1) + (unsigned)j)*(unsigned)g.N1 + (unsigned)i)*8u;      // (.., N2+1, N1)
  const double *f1m = flx1 + (size_t)m*g.nvar*g.N3*PS1, *f2m = flx2 + (size_t)m*g.nvar*g.N3*PS2,
               *f3m = flx3 + (size_t)m*g.nvar*(g.N3 + 1)*PS;
  const double *x31 = e3x1 + (size_t)m*cs, *x21 = e2x1 + (size_t)m*cs, *x12 = e1x2 + (size_t)m*cs,
               *x32 = e3x2 + (size_t)m*cs, *x23 = e2x3 + (size_t)m*cs, *x13 = e1x3 + (size_t)m*cs;
  const double *c1m = c1 + (size_t)m*cs, *c2m = c2 + (size_t)m*cs, *c3m = c3 + (size_t)m*cs;
  double *b01 = b0x1f + (size_t)m*g.N3*PS1, *b02 = b0x2f + (size_t)m*g.N3*PS2, *b03 = b0x3f + (size_t)m*(g.N3 + 1)*PS;
  double *b11 = b1x1f + (size_t)m*g.N3*PS1, *b12 = b1x2f + (size_t)m*g.N3*PS2, *b13 = b1x3f + (size_t)m*(g.N3 + 1)*PS;
  double e1p = 0.0, e2p = 0.0, e3p = 0.0;                       // own edges of the previous plane
  // operands of the corner formulas that belong to plane k-1 (rolled from step to step)
  double x2_km = 0.0, x1_km = 0.0, c1_mm = 0.0, c1_m0 = 0.0, c2_mm = 0.0, c2_m0 = 0.0;
  bool f1_km = false, f2_km = false;          // mass flux >= 0 on the x1 / x2 face of plane k-1
  if (edge_ok) {
    f1_km = ldu(f1m - PS1, o1) >= 0.0;
    f2_km = ldu(f2m - PS2, o2) >= 0.0;
    x2_km = ldu(x12 - PS, oc);
    x1_km = ldu(x21 - PS, oc);
    c1_mm = ldu(c1m - PS - g.N1, oc); c1_m0 = ldu(c1m - PS, oc);
    c2_mm = ldu(c2m - PS - 1, oc); c2_m0 = ldu(c2m - PS, oc);
  }
  for (int k = k0; k <= k1 + 1; ++k) {
    const int t = k - k0;
    const int pp2 = t & 1, p3 = t % 3;
    double e1 = 0.0, e2 = 0.0, e3 = 0.0;
    if (edge_ok) {
      bool f1_k, f1_jm, f2_k, f2_im, f3_k, f3_jm, f3_im;     // mass flux >= 0 on the faces round the corner
      f1_k = ldu(f1m, o1) >= 0.0;
      f1_jm = ldu(f1m - (g.N1 + 1), o1) >= 0.0;
      f2_k = ldu(f2m, o2) >= 0.0;
      f2_im = ldu(f2m - 1, o2) >= 0.0;
      f3_k = ldu(f3m, oc) >= 0.0;
      f3_jm = ldu(f3m - g.N1, oc) >= 0.0;
      f3_im = ldu(f3m - 1, oc) >= 0.0;
      const double c1_0m = ldu(c1m - g.N1, oc), c1_00 = ldu(c1m, oc);
      const double c2_0m = ldu(c2m - 1, oc), c2_00 = ldu(c2m, oc);
      const double x2_k = ldu(x12, oc), x1_k = ldu(x21, oc);
      {  // E1 (mhd_corner_e.cpp:340-363)
        const double x3_jm = ldu(x13 - g.N1, oc), x3_j = ldu(x13, oc);
        double e1_l3 = upw(f2_km, x3_jm, c1_mm, x3_j, c1_m0);
        double e1_r3 = upw(f2_k, x3_jm, c1_0m, x3_j, c1_00);
        double e1_l2 = upw(f3_jm, x2_km, c1_mm, x2_k, c1_0m);
        double e1_r2 = upw(f3_k, x2_km, c1_m0, x2_k, c1_00);
        e1 = 0.25*(e1_l3 + e1_r3 + e1_l2 + e1_r2 + x2_km + x2_k + x3_jm + x3_j);
      }
      {  // E2 (:365-388)
        const double x3_im = ldu(x23 - 1, oc), x3_i = ldu(x23, oc);
        double e2_l3 = upw(f1_km, x3_im, c2_mm, x3_i, c2_m0);
        double e2_r3 = upw(f1_k, x3_im, c2_0m, x3_i, c2_00);
        double e2_l1 = upw(f3_im, x1_km, c2_mm, x1_k, c2_0m);
        double e2_r1 = upw(f3_k, x1_km, c2_m0, x1_k, c2_00);
        e2 = 0.25*(e2_l3 + e2_r3 + e2_l1 + e2_r1 + x3_im + x3_i + x1_km + x1_k);
      }
      {  // E3 (:390-413)
        const double x2_im = ldu(x32 - 1, oc), x2_i = ldu(x32, oc);
        const double x1_jm = ldu(x31 - g.N1, oc), x1_j = ldu(x31, oc);
        const double c_mm = ldu(c3m - g.N1 - 1, oc), c_m0 = ldu(c3m - g.N1, oc);
        const double c_0m = ldu(c3m - 1, oc), c_00 = ldu(c3m, oc);
        double e3_l2 = upw(f1_jm, x2_im, c_mm, x2_i, c_m0);
        double e3_r2 = upw(f1_k, x2_im, c_0m, x2_i, c_00);
        double e3_l1 = upw(f2_im, x1_jm, c_mm, x1_j, c_0m);
        double e3_r1 = upw(f2_k, x1_jm, c_m0, x1_j, c_00);
        e3 = 0.25*(e3_l1 + e3_r1 + e3_l2 + e3_r2 + x2_im + x2_i + x1_jm + x1_j);
      }
      f1_km = f1_k; f2_km = f2_k; x2_km = x2_k; x1_km = x1_k;
      c1_mm = c1_0m; c1_m0 = c1_00; c2_mm = c2_0m; c2_m0 = c2_00;
    }
    if (in_tile) { S1(pp2, ty, tx) = e1; S2(pp2, ty, tx) = e2; S3(p3, ty, tx) = e3; }
    __syncthreads();
    if (own) {
      if (i <= g.ie && j <= g.je && (k <= k1 || wtop)) {          // x3-face of plane k (mhd_ct.cpp:67-77)
        const double b0v = ldu(b03, oc);
        const double b1v = copy_b1 ? b0v : ldu(b13, oc);
        double b = gam0*b0v + gam1*b1v;
        b -= DIVX(beta_dt*(S2(pp2, ty, tx + 1) - e2), dx1);
        b += DIVX(beta_dt*(S1(pp2, ty + 1, tx) - e1), dx2);
        rk_store_u(b03, b13, copy_b1, oc, b0v, b);
      }
      if (k > k0) {
        const int q3 = (t + 2) % 3;                                // the e3 buffer of plane k-1
        if (j <= g.je) {                                           // x1-face (:45-54)
          const double b0v = ldu(b01 - PS1, o1);
          const double b1v = copy_b1 ? b0v : ldu(b11 - PS1, o1);
          double b = gam0*b0v + gam1*b1v;
          b -= DIVX(beta_dt*(S3(q3, ty + 1, tx) - e3p), dx2);
          b += DIVX(beta_dt*(e2 - e2p), dx3);
          rk_store_u(b01 - PS1, b11 - PS1, copy_b1, o1, b0v, b);
        }
        if (i <= g.ie) {                                           // x2-face (:56-65)
          const double b0v = ldu(b02 - PS2, o2);
          const double b1v = copy_b1 ? b0v : ldu(b12 - PS2, o2);
          double b = gam0*b0v + gam1*b1v;
          b += DIVX(beta_dt*(S3(q3, ty, tx + 1) - e3p), dx1);
          b -= DIVX(beta_dt*(e1 - e1p), dx3);
          rk_store_u(b02 - PS2, b12 - PS2, copy_b1, o2, b0v, b);
        }
      }
    }
    e1p = e1; e2p = e2; e3p = e3;
    oc += (unsigned)PS*8u; o1 += (unsigned)PS1*8u; o2 += (unsigned)PS2*8u;
  }
}
#undef DIVX

#ifndef AKMI_CT_WAVES
#define AKMI_CT_WAVES 6         // waves per SIMD the register allocation aims at: 66 VGPRs, three workgroups per CU
#endif                          // (two at the 114 VGPRs the compiler takes when left alone: 640-665 us against 621)
__global__ void __launch_bounds__(CT_THREADS, AKMI_CT_WAVES)
k_corner_ct(Geo g, const double *__restrict__ e3x1, const double *__restrict__ e2x1,
            const double *__restrict__ e1x2, const double *__restrict__ e3x2,
            const double *__restrict__ e2x3, const double *__restrict__ e1x3,
            const double *__restrict__ c1, const double *__restrict__ c2,
            const double *__restrict__ c3, const double *__restrict__ flx1,
            const double *__restrict__ flx2, const double *__restrict__ flx3, double gam0,
            double gam1, double beta_dt, double *__restrict__ b0x1f, double *__restrict__ b0x2f,
            double *__restrict__ b0x3f, double *__restrict__ b1x1f, double *__restrict__ b1x2f,
            double *__restrict__ b1x3f, int copy_b1, int kA, int kB, int top, int nchunk,
            int ckl, int tw, int th, const double *dtp) {
  corner_ct_body<AKMI_POW2DX != 0>(g, e3x1, e2x1, e1x2, e3x2, e2x3, e1x3, c1, c2, c3, flx1, flx2, flx3, gam0,
                                   gam1, beta_dt, b0x1f, b0x2f, b0x3f, b1x1f, b1x2f, b1x3f, copy_b1, kA, kB, top,
                                   nchunk, ckl, tw, th, dtp);
}

// CT (mhd_ct.cpp:23-80) with CopyCons for B folded in at stage 1 (b1 <- b0 old)
__global__ void __launch_bounds__(SX*SY)
k_ct_copy(Geo g, double gam0, double gam1, double beta_dt, const double *__restrict__ e1,
          const double *__restrict__ e2, const double *__restrict__ e3, double *__restrict__ b0x1f,
          double *__restrict__ b0x2f, double *__restrict__ b0x3f, double *__restrict__ b1x1f,
          double *__restrict__ b1x2f, double *__restrict__ b1x3f, int copy_b1, int k0, int nk,
          int kb, const double *dtp) {
  beta_dt = beta_dt_of(beta_dt, dtp);
  // planes k in [k0, k0+nk-1]; x1f/x2f faces are updated for k <= kb (cells of this slab),
  // x3f faces for every k of the launch (the last slab also owns face ke+1)
  const long p = ((long)blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;   // rows [js,je+1] x N1
  const int jj = (int)(p/g.N1);
  const int i = (int)(p - (long)jj*g.N1);
  const int j = g.js + jj;
  const int m = blockIdx.z/nk;
  const int k = k0 + (blockIdx.z - m*nk);
  if (i < g.is || i > g.ie + 1 || j > g.je + 1) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
#define E1(k, j, i) e1[ix4(g.N3 + 1, g.N2 + 1, g.N1, m, k, j, i)]
#define E2(k, j, i) e2[ix4(g.N3 + 1, g.N2, g.N1 + 1, m, k, j, i)]
#define E3(k, j, i) e3[ix4(g.N3, g.N2 + 1, g.N1 + 1, m, k, j, i)]
  if (j <= g.je && k <= kb) {
    size_t c = ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i);
    const double b0v = b0x1f[c];
    const double b1v = copy_b1 ? b0v : b1x1f[c];
    if (g.multi_d) {
      double b = gam0*b0v + gam1*b1v;
      b -= beta_dt*(E3(k, j + 1, i) - E3(k, j, i))/dx2;
      if (g.three_d) b += beta_dt*(E2(k + 1, j, i) - E2(k, j, i))/dx3;
      rk_store(b0x1f, b1x1f, copy_b1, c, b0v, b);
    } else if (copy_b1) {
      b1x1f[c] = b0v;                  // 1-D: bx is constant; either way the register receives it
    }
  }
  if (i <= g.ie && k <= kb) {
    size_t c = ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i);
    const double b0v = b0x2f[c];
    const double b1v = copy_b1 ? b0v : b1x2f[c];
    double b = gam0*b0v + gam1*b1v;
    b += beta_dt*(E3(k, j, i + 1) - E3(k, j, i))/dx1;
    if (g.three_d) b -= beta_dt*(E1(k + 1, j, i) - E1(k, j, i))/dx3;
    rk_store(b0x2f, b1x2f, copy_b1, c, b0v, b);
  }
  if (i <= g.ie && j <= g.je) {
    size_t c = ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i);
    const double b0v = b0x3f[c];
    const double b1v = copy_b1 ? b0v : b1x3f[c];
    double b = gam0*b0v + gam1*b1v;
    b -= beta_dt*(E2(k, j, i + 1) - E2(k, j, i))/dx1;
    if (g.multi_d) b += beta_dt*(E1(k, j + 1, i) - E1(k, j, i))/dx2;
    rk_store(b0x3f, b1x3f, copy_b1, c, b0v, b);
  }
#undef E1
#undef E2
#undef E3
}

// ---------------------------------------------------------------------------------------
// Pass B: c2p over all cells (+ CFL scan over active cells on the last stage)
__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

// SingleC2P_IsothermalHyd / _IsothermalMHD (isothermal_hyd.cpp:30-45, isothermal_mhd.cpp:32-47; density
// floor fmax(dfloor, b^2/sigma_max) :104-106): no energy variable, scalars from variable 4 without a floor
template <bool MHD>
__device__ __forceinline__ void c2p_iso_cell(const Geo &g, const Eos &eos, double *__restrict__ u0,
                                             double *__restrict__ w0, size_t c, size_t cs, double ubx,
                                             double uby, double ubz, int *__restrict__ counters, double &wd,
                                             double &wvx, double &wvy, double &wvz) {
  double ud = u0[c];
  double dfloor_ = eos.dfloor;
  if constexpr (MHD) dfloor_ = fmax(eos.dfloor, (sqr(ubx) + sqr(uby) + sqr(ubz))/eos.sigma_max);
  if (ud < dfloor_) { ud = dfloor_; u0[c] = ud; atomicAdd(&counters[0], 1); }
  const double di = 1.0/ud;
  wd = ud; wvx = di*u0[c + cs]; wvy = di*u0[c + 2*cs]; wvz = di*u0[c + 3*cs];
  w0[c] = wd; w0[c + cs] = wvx; w0[c + 2*cs] = wvy; w0[c + 3*cs] = wvz;
  for (int n = 4; n < g.nvar; ++n) w0[c + n*cs] = u0[c + n*cs]/ud;
}

template <bool MHD>
__global__ void __launch_bounds__(SX*SY)
k_c2p_newdt(Geo g, Eos eos, double *__restrict__ u0, const double *__restrict__ bx1f,
            const double *__restrict__ bx2f, const double *__restrict__ bx3f,
            double *__restrict__ w0, double *__restrict__ bcc0, int do_newdt,
            int *__restrict__ counters, double *__restrict__ dt3, int il, int iu, int jl, int ju,
            int k0, int nk) {
  // cells [il,iu] x [jl,ju] x [k0,k0+nk-1]; lanes over the flattened rows [jl,ju] x N1 of one plane (flat_cells, akmi_common.hpp;
  // with the planes flattened too this kernel is slower on small MeshBlocks: 49 -> 60 us on 120 blocks of 16^3)
  __shared__ double sm[3][SY];
  const Cell3 q = flat_cells(0, g.N1, il, iu, jl, ju - jl + 1, k0, nk);
  const int i = q.i, j = q.j, k = q.k, m = q.m;
  double mv1 = 0.0, mv2 = 0.0, mv3 = 0.0;
  if (q.in) {
    const size_t cs = (size_t)g.N3*g.N2*g.N1;
    const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
    double wd, wvx, wvy, wvz, we = 0.0, ubx = 0, uby = 0, ubz = 0;
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
    const bool scan = do_newdt && i >= g.is && i <= g.ie && j >= g.js && j <= g.je && k >= g.ks && k <= g.ke;
    if (!eos.is_ideal) {
      c2p_iso_cell<MHD>(g, eos, u0, w0, c, cs, ubx, uby, ubz, counters, wd, wvx, wvy, wvz);
      if (scan) {                             // hydro_newdt.cpp:109-111, mhd_newdt.cpp:137-144
        if constexpr (MHD) {
          mv1 = fabs(wvx) + fast_speed_iso(eos.iso_cs, wd, ubx, uby, ubz);
          mv2 = fabs(wvy) + fast_speed_iso(eos.iso_cs, wd, uby, ubz, ubx);
          mv3 = fabs(wvz) + fast_speed_iso(eos.iso_cs, wd, ubz, ubx, uby);
        } else {
          mv1 = fabs(wvx) + eos.iso_cs; mv2 = fabs(wvy) + eos.iso_cs; mv3 = fabs(wvz) + eos.iso_cs;
        }
      }
    } else {
    double ud = u0[c], umx = u0[c + cs], umy = u0[c + 2*cs], umz = u0[c + 3*cs], ue = u0[c + 4*cs];
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
    if (scan) {
      // hydro_newdt.cpp:97-118 / mhd_newdt.cpp:123-136
      const double pr = (eos.gamma - 1.0)*we;
      if constexpr (MHD) {
        mv1 = fabs(wvx) + fast_speed(eos.gamma, wd, pr, ubx, uby, ubz);
        mv2 = fabs(wvy) + fast_speed(eos.gamma, wd, pr, uby, ubz, ubx);
        mv3 = fabs(wvz) + fast_speed(eos.gamma, wd, pr, ubz, ubx, uby);
      } else {
        const double cs_ = sqrt(eos.gamma*pr/wd);
        mv1 = fabs(wvx) + cs_; mv2 = fabs(wvy) + cs_; mv3 = fabs(wvz) + cs_;
      }
    }
    }
  }
  if (!do_newdt) return;        // uniform across the grid
  mv1 = wave_max(mv1); mv2 = wave_max(mv2); mv3 = wave_max(mv3);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.y] = mv1; sm[1][threadIdx.y] = mv2; sm[2][threadIdx.y] = mv3; }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < 3) {
    double v = sm[threadIdx.x][0];
    for (int q = 1; q < SY; ++q) v = fmax(v, sm[threadIdx.x][q]);
    if (v > 0.0) {
      // min over cells of fl(dx/a) == fl(dx/max a): correctly rounded division is monotone
      const double d = g.dx[3*m + threadIdx.x]/v;
      // one word saturates at ~88 atomics/us on this chip: only workgroups that can lower the
      // running minimum issue the atomic (the plain read is a filter, the atomic decides)
      if (d < __hip_atomic_load(&dt3[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(reinterpret_cast<unsigned long long *>(&dt3[threadIdx.x]),
                  (unsigned long long)__double_as_longlong(d));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Pass B, two cells per thread (ideal gas, no passive scalars, rows of an even number of cells): every access of
// the cell-centred arrays is one 16-byte non-temporal access -- the streaming form that copies at 6.8 instead of
// 6.2 TB/s on this chip (tools/micro/copy_bw.hip, profiles/r05_copy_bw.txt); the conversion reads each conserved
// value once and its results are not read again before the cache has turned over.  x1 faces have rows of N1 + 1
// (odd) elements: three 8-byte loads, shared with the neighbouring lanes through L1.
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2_t ld2nt(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p)); }
__device__ __forceinline__ d2_t ld2(const double *p) { return *reinterpret_cast<const d2_t *>(p); }
__device__ __forceinline__ void st2nt(double *p, double a, double b) {
  d2_t v; v.x = a; v.y = b;
  __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(p));
}

// (MHD: 86 VGPRs, five waves per SIMD; held to 80 / 64 registers it spills and runs at 383 / 605 us against 364-389,
//  profiles/r05_c2p_pairs.txt)
template <bool MHD>
__global__ void __launch_bounds__(SX*SY)
k_c2p_newdt2(Geo g, Eos eos, double *__restrict__ u0, const double *__restrict__ bx1f,
             const double *__restrict__ bx2f, const double *__restrict__ bx3f,
             double *__restrict__ w0, double *__restrict__ bcc0, int do_newdt,
             int *__restrict__ counters, double *__restrict__ dt3, int il, int iu, int jl, int ju,
             int k0, int nk) {
  // cells [il,iu] x [jl,ju] x [k0,k0+nk-1], il even and iu odd; lanes run over the cell PAIRS of the flattened rows
  __shared__ double sm[3][SY];
  const int hw = g.N1 >> 1;
  const long p = ((long)blockIdx.x*SY + threadIdx.y)*SX + threadIdx.x;
  const int jj = (int)(p/hw);
  const int j = jl + jj;
  const int i = 2*(int)(p - (long)jj*hw);
  const int m = blockIdx.z/nk;
  const int k = k0 + (blockIdx.z - m*nk);
  double mv1 = 0.0, mv2 = 0.0, mv3 = 0.0;
  if (j <= ju && i >= il && i <= iu) {
    const size_t cs = (size_t)g.N3*g.N2*g.N1;
    const size_t c = ix5(5, g.N3, g.N2, g.N1, m, 0, k, j, i);
    double ubx[2] = {0.0, 0.0}, uby[2] = {0.0, 0.0}, ubz[2] = {0.0, 0.0};
    if constexpr (MHD) {
      const double *b1 = bx1f + ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i);
      const double f0 = b1[0], f1 = b1[1], f2 = b1[2];
      ubx[0] = 0.5*(f0 + f1); ubx[1] = 0.5*(f1 + f2);
      const d2_t y0 = ld2(bx2f + ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i)), y1 = ld2(bx2f + ix4(g.N3, g.N2 + 1, g.N1, m, k, j + 1, i));
      uby[0] = 0.5*(y0.x + y1.x); uby[1] = 0.5*(y0.y + y1.y);
      const d2_t z0 = ld2(bx3f + ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i)), z1 = ld2(bx3f + ix4(g.N3 + 1, g.N2, g.N1, m, k + 1, j, i));
      ubz[0] = 0.5*(z0.x + z1.x); ubz[1] = 0.5*(z0.y + z1.y);
      const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
      st2nt(bcc0 + b, ubx[0], ubx[1]); st2nt(bcc0 + b + cs, uby[0], uby[1]); st2nt(bcc0 + b + 2*cs, ubz[0], ubz[1]);
    }
    const d2_t vd = ld2nt(u0 + c), vx = ld2nt(u0 + c + cs), vy = ld2nt(u0 + c + 2*cs), vz = ld2nt(u0 + c + 3*cs),
               ve = ld2nt(u0 + c + 4*cs);
    double ud[2] = {vd.x, vd.y}, ue[2] = {ve.x, ve.y};
    const double umx[2] = {vx.x, vx.y}, umy[2] = {vy.x, vy.y}, umz[2] = {vz.x, vz.y};
    double wd[2], wvx[2], wvy[2], wvz[2], we[2];
    const bool act = do_newdt && j >= g.js && j <= g.je && k >= g.ks && k <= g.ke;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bool dfl = false, efl = false, tfl = false;
      if constexpr (MHD) {
        c2p_mhd(eos, ud[q], umx[q], umy[q], umz[q], ue[q], ubx[q], uby[q], ubz[q], wd[q], wvx[q], wvy[q], wvz[q], we[q],
                dfl, efl, tfl);
      } else {
        c2p_hyd(eos, ud[q], umx[q], umy[q], umz[q], ue[q], wd[q], wvx[q], wvy[q], wvz[q], we[q], dfl, efl, tfl);
      }
      if (dfl) { u0[c + q] = ud[q]; atomicAdd(&counters[0], 1); }
      if (efl) { u0[c + q + 4*cs] = ue[q]; atomicAdd(&counters[1], 1); }
      if (tfl) { u0[c + q + 4*cs] = ue[q]; atomicAdd(&counters[2], 1); }
      if (act && i + q >= g.is && i + q <= g.ie) {
        // hydro_newdt.cpp:97-118 / mhd_newdt.cpp:123-136
        const double pr = (eos.gamma - 1.0)*we[q];
        if constexpr (MHD) {
          mv1 = fmax(mv1, fabs(wvx[q]) + fast_speed(eos.gamma, wd[q], pr, ubx[q], uby[q], ubz[q]));
          mv2 = fmax(mv2, fabs(wvy[q]) + fast_speed(eos.gamma, wd[q], pr, uby[q], ubz[q], ubx[q]));
          mv3 = fmax(mv3, fabs(wvz[q]) + fast_speed(eos.gamma, wd[q], pr, ubz[q], ubx[q], uby[q]));
        } else {
          const double cs_ = sqrt(eos.gamma*pr/wd[q]);
          mv1 = fmax(mv1, fabs(wvx[q]) + cs_); mv2 = fmax(mv2, fabs(wvy[q]) + cs_); mv3 = fmax(mv3, fabs(wvz[q]) + cs_);
        }
      }
    }
    st2nt(w0 + c, wd[0], wd[1]); st2nt(w0 + c + cs, wvx[0], wvx[1]); st2nt(w0 + c + 2*cs, wvy[0], wvy[1]);
    st2nt(w0 + c + 3*cs, wvz[0], wvz[1]); st2nt(w0 + c + 4*cs, we[0], we[1]);
  }
  if (!do_newdt) return;        // uniform across the grid
  mv1 = wave_max(mv1); mv2 = wave_max(mv2); mv3 = wave_max(mv3);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.y] = mv1; sm[1][threadIdx.y] = mv2; sm[2][threadIdx.y] = mv3; }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < 3) {
    double v = sm[threadIdx.x][0];
    for (int q = 1; q < SY; ++q) v = fmax(v, sm[threadIdx.x][q]);
    if (v > 0.0) {
      const double d = g.dx[3*m + threadIdx.x]/v;
      if (d < __hip_atomic_load(&dt3[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(reinterpret_cast<unsigned long long *>(&dt3[threadIdx.x]),
                  (unsigned long long)__double_as_longlong(d));
    }
  }
}


// ---------------------------------------------------------------------------------------
// Hydro, 3-D, DC/PLM: the three sweeps and the RK update of a stage in ONE kernel.
// A workgroup owns a tile of (tw-1) x (th-1) cell columns (lanes flattened over the tw x th
// positions; the last column / row of positions only provides the face on its low side) and
// marches along k.  Per step k (two barriers):
//   (A) plane k-1 of the primitives sits in LDS (2-low / 1-high halo in i and j).  Every cell of it is
//       reconstructed ONCE per in-plane direction: its position forms the limited slope from the two LDS
//       neighbours, keeps the value at the cell's low face (right state of its own face) and hands the value at
//       the high face to the position above through LDS (that position's left state).  The cells just outside
//       the tile's low sides (one column, one row) are reconstructed by the first th + tw threads.
//   (B) every position solves its low x1 and x2 face; the flux replaces the left state in the same LDS slot.
//       The x3 face below cell k comes from registers (own cells k-1, k, k+1, pending left state).  Plane k
//       replaces plane k-1 in LDS.
//   (C) cell k-1 is finished: divf = dF1/dx1; divf += dF2/dx2; divf += dF3/dx3 (hydro_update.cpp:55-80
//       order, the same rounding sequence as the three-kernel path).
// HBM traffic per cell: w0 once (+halo, L2 hits: neighbouring tiles share an XCD, xcd_order), u0 (+u1) once --
// no flux or partial divergence arrays.  The CT-extended ranges of MHD do not apply (faces is..ie+1 only).
struct HydTile { int tw, th, n1, n2, threads; };
constexpr int HS_THREADS = 512;
#ifndef AKMI_HS_WAVES
#define AKMI_HS_WAVES 3                 // waves per SIMD the register allocation aims at (168 VGPRs, 28 B of scratch: three
                                        // doubles of the x3 march saved and restored once per step; 2: 178 VGPRs, 4: spills 172 B)
#endif

// LDS of a workgroup, in doubles: the plane with its halo and two face-shaped arrays per direction-pair
constexpr int HS_ES = 5;                   // doubles per LDS entry (a padded 48-byte stride measured slower: profiles/r05_hydro_ab.txt)
static size_t hyd_lds_doubles(int tw, int th, int planes = 1) { return HS_ES*(planes*(size_t)(tw + 3)*(th + 3) + 2*(size_t)tw*th); }

static HydTile hyd_tile(int c1, int c2, int planes = 1) {
  // The kernel is bound by the issue of dependent fp64 chains (1 330 VALU instructions per cell, three waves per SIMD), so
  // what a shape costs is the lanes it launches per owned column, not the length of its rows:
  //   cost = lanes launched (tiles x padded workgroup)
  //   x (1 + 0.3 x plane-with-halo / owners)    the LDS plane every step loads: (tw+3)(th+3) entries for (tw-1)(th-1) owners
  //   x (1 + 4/tw)                              short rows (tw = 9: +15 % measured)
  //   / occupancy                               workgroups per CU by LDS x waves per workgroup, against the 12 waves the
  //                                             168 VGPRs allow
  // fitted to the scans in profiles/r04_hydro_tiles.txt and r05_hydro_ab.txt (256^3, this kernel: 23 x 11 960 us, 28 x 9 964,
  // 18 x 14 990, 21 x 12 990; 384-thread tiles 27 x 14 / 24 x 16: 1 380-1 480 -- six-wave workgroups leave wave slots idle).
  static const int maxt = getenv("AKMI_HS_MAXT") ? atoi(getenv("AKMI_HS_MAXT")) : 256;   // three four-wave workgroups per CU
  static const int maxlds = getenv("AKMI_HS_LDS") ? atoi(getenv("AKMI_HS_LDS")) : 53*1024;
  static int f_tw = -1, f_th = 0;                        // AKMI_HS_TILE=tw,th pins the shape (experiments)
  if (f_tw < 0) {
    const char *e = getenv("AKMI_HS_TILE");                 // "tw,th" or "twxth"
    f_tw = 0;
    if (e && sscanf(e, "%d%*[,x]%d", &f_tw, &f_th) != 2) f_tw = 0;
    if (f_tw < 4 || f_th < 3 || f_tw*f_th > HS_THREADS) f_tw = 0;
  }
  if (f_tw > 0 && 3*(f_tw + 3) + 3*f_th <= f_tw*f_th &&
      hyd_lds_doubles(f_tw, f_th, planes)*sizeof(double) <= 150*1024)
    return HydTile{f_tw, f_th, (c1 + f_tw - 2)/(f_tw - 1), (c2 + f_th - 2)/(f_th - 1), (f_tw*f_th + 63)/64*64};
  HydTile best{0, 0, 0, 0, 0};
  double best_cost = -1.0;
  for (int n1 = 1; n1 <= c1; ++n1) {
    const int tw = (c1 + n1 - 1)/n1 + 1;
    if (tw > 130) continue;
    if (tw < 8 && n1 > 1) break;
    for (int th = 3; th <= 40; ++th) {
      if (tw*th > maxt) break;
      if (3*(tw + 3) + 3*th > tw*th) continue;         // one halo entry per thread at most
      const size_t lds = hyd_lds_doubles(tw, th, planes)*sizeof(double);
      if (lds > (size_t)maxlds) continue;
      const int n2 = (c2 + th - 2)/(th - 1);
      const int threads = (tw*th + 63)/64*64;
      const int waves_cu = (int)(160*1024/lds)*(threads/64);
      const double occ = (waves_cu < 12 ? waves_cu : 12)/12.0;
      const double halo = (double)(tw + 3)*(th + 3)/((double)(tw - 1)*(th - 1));
      const double cost = (double)n1*n2*threads*(1.0 + 0.3*halo)*(1.0 + 4.0/tw)/occ;
      if (best_cost < 0 || cost < best_cost) { best = HydTile{tw, th, n1, n2, threads}; best_cost = cost; }
    }
  }
  return best;
}

// MASS: passive scalars ride along -- leave the three mass fluxes behind for k_scalar_update
struct Mass3 { double *m1, *m2, *m3; };
template <int RECON, int RS, bool MASS = false>
__global__ void __launch_bounds__(HS_THREADS, AKMI_HS_WAVES)
k_hydro_stage3d(Geo g, FaceEos eos, const double *__restrict__ w0, UpdArgs u, int kA, int kB,
                int nchunk, int ckl, int tw, int th, Mass3 ms) {
  static_assert(RECON <= 1, "one-kernel hydro stage: DC and PLM");
  constexpr bool ISO = rs_iso<RS>();        // isothermal: variable 4 (energy) does not exist; its slots stay unused
#define ISOSKIP if (ISO && n == 4) continue
  extern __shared__ double hs_lds[];
  const int pw = tw + 3, ph = th + 3;        // plane with halo: cols i0-2..i0+tw, rows j0-2..j0+th
  const int qn = ph*pw, fn = th*tw;
  constexpr int ES = HS_ES;                 // doubles per LDS entry (five used)
  // SQ: primitives of the plane being worked on.  SX1/SX2 (one entry per position, x1 / x2 direction) hold FIRST the
  // left state of the position's low face (written by the cell on the low side: every cell is reconstructed once per
  // direction), THEN, after the solve, the flux of that face (written by the position itself, read by its low-side
  // neighbour for the flux difference).  Entry (r,t) changes hands only between threads (r,t-1) / (r-1,t) and (r,t)
  // across a barrier, so one array serves both and nothing is double-buffered.
  // Layout: position-major, the five variables of an entry adjacent (stride 5 doubles: conflict-free for 64-bit
  // accesses), so that one address register per thread and array serves every variable and the x-neighbours through
  // the instruction's immediate offset; the y-neighbours cost one more register per array.
  const int tid = threadIdx.x;
  const int r = tid/tw, t = tid - r*tw;
  const bool in_tile = r < th;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {                       // tiles of a k-chunk of a block side by side on one XCD (x fastest, then y, then chunk / block)
    const unsigned lin = xcd_order(bx + gridDim.x*(by + gridDim.y*bz), gridDim.x*gridDim.y*gridDim.z);
    const unsigned row = lin/gridDim.x;
    bx = lin - row*gridDim.x; bz = row/gridDim.y; by = row - bz*gridDim.y;
  }
  const int i0 = g.is + (int)bx*(tw - 1), j0 = g.js + (int)by*(th - 1);
  const int i = i0 + t, j = j0 + r;
  const int m = (int)bz/nchunk;
  const int ch = (int)bz - m*nchunk;
  const int k0 = kA + ch*ckl;
  const int k1 = (k0 + ckl - 1 < kB) ? k0 + ckl - 1 : kB;
  const bool cell_ok = in_tile && i < g.N1 && j < g.N2;                // the column exists in memory
  const bool own = in_tile && t < tw - 1 && r < th - 1 && i <= g.ie && j <= g.je;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  // power-of-two cell sizes: x/dx == ldexp(x, n) bit for bit (pow2_shift); beta*dt once, in scalar registers
  const bool p2 = AKMI_POW2DX && is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);      // wave-uniform
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
  const double bdt = to_sgpr(beta_dt_of(u.beta_dt, u.dtp));
  const size_t cs = (size_t)g.N3*g.N2*g.N1, ps = (size_t)g.N2*g.N1;
  const double *wb = w0 + (size_t)m*g.nvar*cs;
  // halo entry of this thread: rows 0,1 and ph-1 in full, columns 0,1 and pw-1 of the tile rows
  int hy = -1, hx = 0;
  {
    const int nh = 3*pw + 3*th;
    if (tid < nh) {
      if (tid < 3*pw) { const int q = tid/pw; hy = q < 2 ? q : ph - 1; hx = tid - q*pw; }
      else { const int q = tid - 3*pw; const int rr = q/3, cc = q - rr*3; hy = 2 + rr; hx = cc < 2 ? cc : pw - 1; }
    }
  }
  const bool hload = hy >= 0 && j0 - 2 + hy < g.N2 && i0 - 2 + hx < g.N1;      // the halo cell exists in memory
  // the cells just outside the tile's low sides have no position of their own: column 1 of the plane (rows of the
  // tile) and row 1 (columns of the tile) are reconstructed by the first th + tw threads, one cell each, in the one
  // direction in which a face of the tile needs them
  int ha = -1, hb = 0, hc = 0, hd = 0;        // LDS offsets (per variable plane: + n*qn / + n*fn) of below, here, above, destination
  if (tid < th) { ha = (tid + 2)*pw*ES; hb = ha + ES; hc = ha + 2*ES; hd = ES*qn + tid*tw*ES; }
  else if (tid < th + tw) { const int c = tid - th; ha = (c + 2)*ES; hb = ha + ES*pw; hc = hb + ES*pw; hd = ES*qn + ES*fn + c*ES; }
  // own entries: cell (r+2, t+2) of the plane, position (r, t) of the two face arrays
  const int qo = in_tile ? ((r + 2)*pw + t + 2)*ES : 0, qy = ES*pw;
  const int xo = in_tile ? ES*qn + (r*tw + t)*ES : ES*qn, x2o = xo + ES*fn, xy = ES*tw;
  const int hq = hy >= 0 ? (hy*pw + hx)*ES : 0;
#define SX1o(n) hs_lds[xo + (n)]
#define SX2o(n) hs_lds[x2o + (n)]
  // (scalar base of the plane + 32-bit byte offset of the lane within it)
  const unsigned hcol = hload ? ((unsigned)(j0 - 2 + hy)*(unsigned)g.N1 + (unsigned)(i0 - 2 + hx))*8u : 0u;
  const unsigned col = cell_ok ? ((unsigned)j*(unsigned)g.N1 + (unsigned)i)*8u : 0u;
  const size_t mb = (size_t)m*g.nvar*cs;
  double W0[5], W1[5], PL[5], F3p[5], hv[5];
#pragma unroll
  for (int n = 0; n < 5; ++n) {
      ISOSKIP;
    const double *q = wb + n*cs;
    W0[n] = cell_ok ? ldu(q + (size_t)(k0 - 1)*ps, col) : 1.0;
    W1[n] = cell_ok ? ldu(q + (size_t)k0*ps, col) : 1.0;
    if constexpr (RECON == 1) {
      const double qa = cell_ok ? ldu(q + (size_t)(k0 - 2)*ps, col) : 1.0;
      double dummy;
      plm(qa, W0[n], W1[n], PL[n], dummy);
    } else {
      PL[n] = W0[n];
    }
    F3p[n] = 0.0;
    hv[n] = hload ? ldu(wb + n*cs + (size_t)k0*ps, hcol) : 1.0;
  }
  // step k: x3 face k (below cell k) from registers; for k > k0 also the x1/x2 faces of plane k-1
  // (in LDS since the previous step), which finishes cell k-1; then plane k replaces it
  for (int k = k0; k <= k1 + 1; ++k) {
    const bool plane = k > k0;                        // workgroup-uniform
    double wp[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) { ISOSKIP; wp[n] = cell_ok ? ldu(wb + n*cs + (size_t)(k + 1)*ps, col) : 1.0; }
    double pu0[5], pu1[5];
    if (plane && own) {                               // operands of the update, used after the solves
      const size_t c = mb + (size_t)(k - 1)*ps;
#pragma unroll
      for (int n = 0; n < 5; ++n) {
      ISOSKIP;
        pu0[n] = ldu(u.u0 + c + n*cs, col);
        pu1[n] = u.copy_u1 ? 0.0 : ldu(u.u1 + c + n*cs, col);
      }
    }
    double f1[5] = {0, 0, 0, 0, 0}, f2[5] = {0, 0, 0, 0, 0};      // this position's own in-plane fluxes
    if (plane) {
      // (A) every cell of plane k-1 once per direction: the value at its upper face goes to the position above (its
      // left state), the value at its lower face stays here (the right state of this position's own low face)
      double R1[5] = {0, 0, 0, 0, 0}, R2[5] = {0, 0, 0, 0, 0};
      if (in_tile) {
        double U1[5] = {0, 0, 0, 0, 0}, U2[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < 5; ++n) {
      ISOSKIP;
          if constexpr (RECON == 1) {
            plm(hs_lds[qo - ES + n], W0[n], hs_lds[qo + ES + n], U1[n], R1[n]);
            plm(hs_lds[qo - qy + n], W0[n], hs_lds[qo + qy + n], U2[n], R2[n]);
          } else {
            R1[n] = W0[n]; R2[n] = W0[n]; U1[n] = W0[n]; U2[n] = W0[n];
          }
        }
        if (t + 1 < tw) {
#pragma unroll
          for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[xo + ES + n] = U1[n]; }
        }
        if (r + 1 < th) {
#pragma unroll
          for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[x2o + xy + n] = U2[n]; }
        }
      }
      if (ha >= 0) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
      ISOSKIP;
          if constexpr (RECON == 1) {
            double up, dummy;
            plm(hs_lds[ha + n], hs_lds[hb + n], hs_lds[hc + n], up, dummy);
            hs_lds[hd + n] = up;
          } else {
            hs_lds[hd + n] = hs_lds[hb + n];
          }
        }
      }
      __syncthreads();
      // (B) the two in-plane faces of this position; the flux takes the place of the left state
      if (in_tile) {
        {
          double fd, fx, fy, fz, fe;
          riemann_hyd_e<RS, true>(eos, SX1o(0), SX1o(1), SX1o(2), SX1o(3), ISO ? 0.0 : SX1o(4),
                            R1[0], R1[1], R1[2], R1[3], R1[4], fd, fx, fy, fz, fe);
          SX1o(0) = fd; SX1o(1) = fx; SX1o(2) = fy; SX1o(3) = fz;
          if constexpr (!ISO) SX1o(4) = fe;
          f1[0] = fd; f1[1] = fx; f1[2] = fy; f1[3] = fz; f1[4] = fe;
          if constexpr (MASS) {
            if (i <= g.ie + 1 && j <= g.je) ms.m1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, 0, k - 1, j, i)] = fd;
          }
        }
        {  // sweep-aligned order (d, vy, vz, vx, e)
          double fd, fx, fy, fz, fe;
          riemann_hyd_e<RS, true>(eos, SX2o(0), SX2o(2), SX2o(3), SX2o(1), ISO ? 0.0 : SX2o(4),
                            R2[0], R2[2], R2[3], R2[1], R2[4], fd, fx, fy, fz, fe);
          SX2o(0) = fd; SX2o(2) = fx; SX2o(3) = fy; SX2o(1) = fz;
          if constexpr (!ISO) SX2o(4) = fe;
          f2[0] = fd; f2[2] = fx; f2[3] = fy; f2[1] = fz; f2[4] = fe;
          if constexpr (MASS) {
            if (i <= g.ie && j <= g.je + 1) ms.m2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, 0, k - 1, j, i)] = fd;
          }
        }
      }
    }
    // x3 face below cell k: sweep-aligned order (d, vz, vx, vy, e)
    double f3[5];
    {
      double L[5] = {0, 0, 0, 0, 0}, R[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < 5; ++n) {
      ISOSKIP;
        if constexpr (RECON == 1) {
          double qln;
          L[n] = PL[n];
          plm(W0[n], W1[n], wp[n], qln, R[n]);
          PL[n] = qln;
        } else {
          L[n] = W0[n]; R[n] = W1[n];
        }
      }
      double fd, fx, fy, fz, fe;
      riemann_hyd_e<RS, true>(eos, L[0], L[3], L[1], L[2], L[4], R[0], R[3], R[1], R[2], R[4], fd, fx, fy,
                      fz, fe);
      f3[0] = fd; f3[3] = fx; f3[1] = fy; f3[2] = fz; f3[4] = fe;
      if constexpr (MASS) {
        if (own) ms.m3[ix5(g.nvar, g.N3 + 1, g.N2, g.N1, m, 0, k, j, i)] = fd;
      }
    }
    if (k <= k1) {                                     // plane k for the next step (every reader of plane k-1 is past (A))
      if (in_tile) {
#pragma unroll
        for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[qo + n] = W1[n]; }
      }
      if (hy >= 0) {
#pragma unroll
        for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[hq + n] = hv[n]; }
        if (hload && k + 1 <= k1) {
#pragma unroll
          for (int n = 0; n < 5; ++n) { ISOSKIP; hv[n] = ldu(wb + n*cs + (size_t)(k + 1)*ps, hcol); }
        }
      }
    }
    __syncthreads();
    if (plane && own) {                                // (C) finish cell k-1
      const size_t c = mb + (size_t)(k - 1)*ps;
      double divf[5];
      if (p2) {                                        // one wave-uniform branch for the fifteen quotients
#pragma unroll
        for (int n = 0; n < 5; ++n) {
      ISOSKIP;
          divf[n] = ldexp(hs_lds[xo + ES + n] - f1[n], n1);
          divf[n] += ldexp(hs_lds[x2o + xy + n] - f2[n], n2);
          divf[n] += ldexp(f3[n] - F3p[n], n3);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
      ISOSKIP;
          divf[n] = (hs_lds[xo + ES + n] - f1[n])/dx1;
          divf[n] += (hs_lds[x2o + xy + n] - f2[n])/dx2;
          divf[n] += (f3[n] - F3p[n])/dx3;
        }
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
      ISOSKIP;
        const double u0v = pu0[n];
        const double u1v = u.copy_u1 ? u0v : pu1[n];
        rk_store_u(u.u0 + c + n*cs, u.u1 + c + n*cs, u.copy_u1, col, u0v,
                   u.gam0*u0v + u.gam1*u1v - bdt*divf[n]);
      }
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      ISOSKIP; F3p[n] = f3[n]; W0[n] = W1[n]; W1[n] = wp[n]; }
  }
#undef SQ
#undef SX1o
#undef SX2o
#undef ISOSKIP
}


#include "akmi_hydro_stage3d2.hpp"
#include "akmi_mhd_stage3d.hpp"

__global__ void k_init_dt3(double *dt3) {
  if (threadIdx.x < 3) dt3[threadIdx.x] = (double)FLT_MAX;
}

template <int DIR, bool MHD, bool ECC>
static int launch_sweep(const Geo &g, const Scheme &sc, const SweepArgs &a, hipStream_t st) {
  int nk = a.ku - a.kl + 1;
  dim3 block(SX, SY);
  int rc = dispatch_scheme_eos<MHD>(sc, [&](auto R, auto S) {
    // faces per wave: 63 when the lanes share their slopes (lane 0 of a wave only provides); those kernels run over
    // the columns il-1 .. iu of a row (sweep_x1_shared), the plain one over il (- 1 with ECC) .. iu
    constexpr bool share = x1_share<DIR, decltype(R)::value>();
    const long np = (long)nk*(a.ju - a.jl + 1)*(share ? a.iu - a.il + 2 : a.iu - a.il + 1 + (ECC ? 1 : 0));   // of one MeshBlock
    const long per_wg = (long)(share ? SX - 1 : SX)*SY;
    dim3 grid((unsigned)((np + 1 + per_wg - 1)/per_wg), 1, g.nmb);
    k_sweep<DIR, decltype(R)::value, MHD, ECC, decltype(S)::value><<<grid, block, 0, st>>>(
        g, sc.eos, a, nk);
    return AKMI_COMPLETE;
  });
  if (rc != AKMI_COMPLETE) return rc;
  AKMI_CHECK_LAUNCH("sweep");
  return AKMI_COMPLETE;
}

template <int DIR, bool MHD, int MODE = 0, bool USEACC = false>
static int launch_sweep_update(const Geo &g, const Scheme &sc, const SweepArgs &a,
                               const UpdArgs &u, hipStream_t st) {
  int rc;
  if constexpr (DIR == 0) {
    dim3 grid(cdiv(a.iu - a.il + 1, TX - 1), 1, g.nmb), block(TX, 1);
    rc = dispatch_scheme_eos<MHD>(sc, [&](auto R, auto S) {
      k_sweep_update_1d<decltype(R)::value, MHD, decltype(S)::value><<<grid, block, 0, st>>>(
          g, sc.eos, a, u);
      return AKMI_COMPLETE;
    });
  } else {
    dim3 grid, block(SX, SY);
    int ml;
    if (DIR == 2) {
      long np = (long)(a.ju - a.jl + 1)*(MODE == 2 ? a.iu - a.il + 1 : g.N1);          // flattened rows (sweep_update_body)
      const unsigned nb = (unsigned)((np + SX*SY - 1)/(SX*SY));
      const int nc = a.ku - a.kl > 0 ? a.ku - a.kl : 1;
      // resident workgroups per CU: the PLM march is compiled for AKMI_X3_WAVES waves per SIMD, the
      // five-point schemes for two
      const int res3 = (sc.recon >= AKMI_RECON_PPM4) ? 2 : AKMI_X3_WAVES;
      ml = march_len(nb, nc, g.nmb, ML, res3);
      grid = dim3(nb, cdiv(nc, ml), g.nmb);
    } else {
      long np = (long)(a.ku - a.kl + 1)*(MODE == 2 ? a.iu - a.il + 1 : g.N1);          // flattened (k,i)
      const unsigned nb = (unsigned)((np + SX*SY - 1)/(SX*SY));
      const int nc = a.ju - a.jl > 0 ? a.ju - a.jl : 1;
      ml = march_len(nb, nc, g.nmb, ML, (sc.recon >= AKMI_RECON_PPM4) ? 2 : AKMI_X2_WAVES);
      grid = dim3(nb, cdiv(nc, ml), g.nmb);
    }
    constexpr int D = (DIR == 0) ? 1 : DIR;
    rc = dispatch_scheme_eos<MHD>(sc, [&](auto R, auto S) {
      k_sweep_update<D, decltype(R)::value, MHD, MODE, USEACC, decltype(S)::value>
          <<<grid, block, 0, st>>>(g, sc.eos, a, u, ml);
      return AKMI_COMPLETE;
    });
  }
  if (rc != AKMI_COMPLETE) return rc;
  AKMI_CHECK_LAUNCH("sweep_update");
  return AKMI_COMPLETE;
}


// x1 sweep + x2 march of an MHD pack in one kernel (3-D)
// ---------------------------------------------------------------------------------------
// k_sweep12s: the x1 sweep inside the x2 march (MHD, 3-D, PLM).  While a thread walks along j it also solves
// the x1 face on the low side of its cell in row s-1, the row whose cells the step finishes; the x1 flux
// difference of the cell goes straight into acc = dF1/dx1 + dF2/dx2 (rounding sequence of
// hydro_update.cpp:55-80 unchanged), so the 5-component x1 flux array is never written.  The cells of
// row s-1 are the cells the march loaded one step earlier: they sit in its window (LDS), so the x1
// part issues TWO loads per step (the cell-centred By, which the x2 march does not carry, and the face
// field); the i-1 / i+1 neighbours of the slopes, the left state of the face and the flux of
// face i+1 come from the neighbouring lanes (wave shuffles).  Waves overlap by four lanes: lanes 0 and 63
// only provide cells, lane 1 a left state, lane 62 a face flux; lanes 2..61 own cells.
#ifndef AKMI_X12S_WAVES
#define AKMI_X12S_WAVES 3
#endif
#ifndef AKMI_X12S_EO1
#define AKMI_X12S_EO1 1
#endif
#ifndef AKMI_X12S_EO2
#define AKMI_X12S_EO2 1
#endif
template <int RS>
__global__ void __launch_bounds__(SX*SY, AKMI_X12S_WAVES)
k_sweep12s(Geo g, FaceEos eos, SweepArgs a1, SweepArgs a2, UpdArgs u, int ml) {
  constexpr int NV = 7, NW = 2, NT = SX*SY;
  constexpr int CPW = SX - 4;                              // cells per wave
  __shared__ double sm[(NV*NW + NV + 5)*SX*SY];
  const int lane = threadIdx.x;
  const long wv = (long)blockIdx.x*SY + threadIdx.y;       // wave index over the (k,i) rows
  const long pe = wv*CPW + lane - 2;
  const long np = (long)(a2.ku - a2.kl + 1)*g.N1;
  if (wv*CPW - 2 >= np) return;                            // whole wave beyond the planes
  const long p = pe < 0 ? 0 : (pe > np - 1 ? np - 1 : pe); // clamped lanes compute, never store
  const int kk = (int)(p/g.N1);
  const int i = (int)(p - (long)kk*g.N1);
  const int k = a2.kl + kk;
  const int m = blockIdx.z;
  const int s0 = a2.jl + blockIdx.y*ml;
  const int shi = a2.ju;                                   // last x2 face
  const bool exact = pe == p;
  const bool inner = exact && lane >= 2 && lane <= SX - 3;
  const bool x2_ok = inner && i >= a2.il && i <= a2.iu;    // owns the x2 faces of this column
  const bool x1_ok = inner && i >= a1.il && i <= a1.iu;    // owns the x1 faces of this column
  const bool col_active = inner && i >= g.is && i <= g.ie && k >= g.ks && k <= g.ke;
  double *my = sm + threadIdx.y*SX + threadIdx.x;
#define W_(n, c) my[((n)*NW + (c))*NT]
#define PL_(n) my[(NV*NW + (n))*NT]
#define FP_(n) my[(NV*NW + NV + (n))*NT]
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1];
  // blocks whose cell sizes are powers of two (the usual case: unit-length domains with 2^n cells, every level
  // of a refined mesh): 1/dx is exact and x/dx == x*(1/dx) bit for bit (both are the correctly rounded value of
  // the same real number), so the ten divisions of a step become products; wave-uniform branch per step
  const bool p2 = AKMI_POW2DX && is_pow2(dx1) && is_pow2(dx2);
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2);
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const long st = (long)g.N1;
  const double *wb = a2.w0 + (size_t)m*g.nvar*cs;
  const double *bb = a2.bcc0 + (size_t)m*3*cs;
  // x2-aligned order of the march: d, vy, vz, vx, e, bz, bx
  auto base2 = [&](int n) -> const double * {
    return n == 0 ? wb : n == 1 ? wb + 2*cs : n == 2 ? wb + 3*cs : n == 3 ? wb + cs
         : n == 4 ? wb + 4*cs : n == 5 ? bb + 2*cs : bb;
  };
  unsigned off = (((unsigned)k*(unsigned)g.N2 + (unsigned)s0)*(unsigned)g.N1 + (unsigned)i)*8u;    // cell (k, s, i)
  const unsigned st8 = (unsigned)g.N1*8u;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const double *q = base2(n);
    double pl, dummy;
    const double qa = ldu(q - 2*st, off), qb = ldu(q - st, off), qc = ldu(q, off);
    plm(qa, qb, qc, pl, dummy);
    W_(n, 0) = qb; W_(n, 1) = qc;
    PL_(n) = pl;
  }
  // face-shaped arrays: x2 faces (f3, f2, f1) = (N3, N2+1, N1), x1 faces (N3, N2, N1+1)
  const size_t fs2 = (size_t)a2.f3*a2.f2*a2.f1, fs1 = (size_t)a1.f3*a1.f2*a1.f1;
  unsigned foff2 = (((unsigned)k*(unsigned)a2.f2 + (unsigned)s0)*(unsigned)a2.f1 + (unsigned)i)*8u;   // x2 face s
  const unsigned fst28 = (unsigned)a2.f1*8u;
  unsigned foff1 = (((unsigned)k*(unsigned)a1.f2 + (unsigned)(s0 - 1))*(unsigned)a1.f1 + (unsigned)i)*8u;   // x1 face (k, s-1, i)
  const unsigned fst18 = (unsigned)a1.f1*8u;
  const double *bx2m = a2.bxf + (size_t)m*fs2, *bx1m = a1.bxf + (size_t)m*fs1;
  double *mf2 = a2.flx + (size_t)m*g.nvar*fs2, *mf1 = a1.flx + (size_t)m*g.nvar*fs1;
  const double *bym = bb + cs;                                                                   // cell-centred By
  const size_t mb = (size_t)m*g.nvar*cs;
  double by_n = ldu(bym, off - st8), bx1_n = ldu(bx1m, foff1);
  for (int t = 0;; ++t) {
    const int s = s0 + t;
    if (s > shi + 1) break;                   // the last chunk ends with the x1 faces of row ju(x1)
    if (t > ml && s <= shi) break;            // the others end with the face they share
    const bool do_x2 = s <= shi;
    const int jr = s - 1;                     // row of the x1 faces of this step
    const bool do_x1 = (t >= 1 || s0 == a2.jl) && jr >= a1.jl && jr <= a1.ju;
    const unsigned orow = off - st8;          // cell (k, jr, i)
    // Every load of the step is unconditional and issued in one group.  (With `do_x2 ? ldu(..) : w1` each of
    // the seven cell loads sat in its own scalar branch followed by s_waitcnt vmcnt(0): eight serialised
    // memory round trips per step, profiles/r03_isa_audit.txt.)  Beyond the last face the address is the
    // row the window already holds, i.e. the value `w1` the old form substituted.
    const double by_c = by_n, bx1 = bx1_n;
    by_n = ldu(bym, off);
    bx1_n = ldu(bx1m + a1.f1, foff1);
    double qn[NV];
    {
      const long stq = do_x2 ? st : 0;
#pragma unroll
      for (int n = 0; n < NV; ++n) qn[n] = ldu(base2(n) + stq, off);
    }
    const double bx2 = ldu(bx2m, do_x2 ? foff2 : foff2 - fst28);
    // ---- x1 face on the low side of cell (k, jr, i): the cell is W_(.,0) of the march
    double f1d, f1x, f1y, f1z, f1e, f1by, f1bz;
    double dF1[5];
    {
      // x1-aligned order d, vx, vy, vz, e, by, bz from the x2-aligned window d, vy, vz, vx, e, bz, bx
      double q0[NV];
      q0[0] = W_(0, 0); q0[1] = W_(3, 0); q0[2] = W_(1, 0); q0[3] = W_(2, 0); q0[4] = W_(4, 0);
      q0[5] = by_c; q0[6] = W_(5, 0);
      const double bxc = W_(6, 0);
      double qln[NV], qr[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        const double qm = AKMI_LANE_BELOW(q0[n]), qp = AKMI_LANE_ABOVE(q0[n]);
        plm(qm, q0[n], qp, qln[n], qr[n]);
      }
      if (do_x1 && inner && i >= a1.il - 1 && i <= a1.iu) {          // cell-centred E = -(v x B)
        stu(a1.ecc1 + (size_t)m*cs, orow, q0[3]*q0[5] - q0[2]*q0[6]);
        stu(a1.ecc2 + (size_t)m*cs, orow, q0[1]*q0[6] - q0[3]*bxc);
        stu(a1.ecc3 + (size_t)m*cs, orow, q0[2]*bxc - q0[1]*q0[5]);
      }
      double L1[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) L1[n] = AKMI_LANE_BELOW(qln[n]);
      Cons1D f1 = riemann_mhd_e<RS, AKMI_X12S_EO1 != 0, AKMI_X12S_FM != 0>(eos, L1[0], L1[1], L1[2], L1[3], L1[4], L1[5], L1[6], qr[0],
                                    qr[1], qr[2], qr[3], qr[4], qr[5], qr[6], bx1);
      f1d = f1.d; f1x = f1.mx; f1y = f1.my; f1z = f1.mz; f1e = f1.e; f1by = f1.by; f1bz = f1.bz;
      dF1[0] = AKMI_LANE_ABOVE(f1d) - f1d;
      dF1[1] = AKMI_LANE_ABOVE(f1x) - f1x;
      dF1[2] = AKMI_LANE_ABOVE(f1y) - f1y;
      dF1[3] = AKMI_LANE_ABOVE(f1z) - f1z;
      dF1[4] = AKMI_LANE_ABOVE(f1e) - f1e;
      if (do_x1 && x1_ok) {
        stu(mf1, foff1, f1d);
        stu(a1.ey + (size_t)m*cs, orow, -f1by);
        stu(a1.ez + (size_t)m*cs, orow, f1bz);
      }
    }
    // ---- x2 face s
    double L[NV], R[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      double qln2;
      L[n] = PL_(n);
      const double w0 = W_(n, 0), w1 = W_(n, 1);
      const double qp = qn[n];
      plm(w0, w1, qp, qln2, R[n]);
      W_(n, 0) = w1; W_(n, 1) = qp;
      PL_(n) = qln2;
    }
    Cons1D f2 = riemann_mhd_e<RS, AKMI_X12S_EO2 != 0, AKMI_X12S_FM != 0>(eos, L[0], L[1], L[2], L[3], L[4], L[5], L[6], R[0], R[1],
                                  R[2], R[3], R[4], R[5], R[6], bx2);
    if (do_x2 && x2_ok && (t < ml || s == shi)) {
      stu(mf2, foff2, f2.d);
      stu(a2.ey + (size_t)m*cs, off, -f2.by);
      stu(a2.ez + (size_t)m*cs, off, f2.bz);
    }
    // x2 flux in natural component order: d, m1, m2, m3, E  (ivx = 2, ivy = 3, ivz = 1)
    const double fv[5] = {f2.d, f2.mz, f2.mx, f2.my, f2.e};
    const int sc = s - 1;
    if (do_x2 && t > 0 && col_active && sc >= g.js && sc <= g.je) {
      if (p2) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          double divf = ldexp(dF1[n], n1);
          divf += ldexp(fv[n] - FP_(n), n2);
          stu(u.acc + mb + n*cs, orow, divf);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          double divf = dF1[n]/dx1;
          divf += (fv[n] - FP_(n))/dx2;
          stu(u.acc + mb + n*cs, orow, divf);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) FP_(n) = fv[n];
    off += st8; foff2 += fst28; foff1 += fst18;
  }
#undef W_
#undef PL_
#undef FP_
}

static int launch_sweep12s(const Geo &g, const Scheme &sc, const SweepArgs &a1, const SweepArgs &a2,
                           const UpdArgs &u, hipStream_t st) {
  const long np = (long)(a2.ku - a2.kl + 1)*g.N1;          // flattened (k,i) rows
  const long nwaves = (np + (SX - 4) - 1)/(SX - 4);
  const unsigned nb = (unsigned)((nwaves + SY - 1)/SY);
  const int nc = a2.ju - a2.jl > 0 ? a2.ju - a2.jl : 1;
  const int ml = march_len(nb, nc, g.nmb, ML, 3);
  dim3 grid(nb, cdiv(nc, ml), g.nmb), block(SX, SY);
  const int rs = sc.rsolver;
  if (sc.iso || sc.recon != 1 || rs != AKMI_RS_HLLD) { set_error("sweep12s: PLM + HLLD, ideal gas"); return AKMI_FAIL; }
  k_sweep12s<3><<<grid, block, 0, st>>>(g, sc.eos, a1, a2, u, ml);
  AKMI_CHECK_LAUNCH("sweep12s");
  return AKMI_COMPLETE;
}


// dynamic LDS beyond the 64 KB a kernel gets by default: granted once per kernel function and size
static int ensure_lds(const void *kern, size_t lds, const char *who) {
  static std::map<const void *, size_t> granted;
  size_t &gr = granted.emplace(kern, (size_t)64*1024).first->second;
  if (lds <= gr) return AKMI_COMPLETE;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    set_error("%s: %zu bytes of LDS refused", who, lds);
    return AKMI_FAIL;
  }
  gr = lds;
  return AKMI_COMPLETE;
}

static bool hyd_two_planes() {      // AKMI_HS2=0: the one-plane kernel of round 5 (A/B runs)
  static const bool two = !(getenv("AKMI_HS2") && atoi(getenv("AKMI_HS2")) == 0);
  return two;
}
// may the stage kernel convert the cells it finishes (k_hydro_stage3d2<.., C2P>)?  3-D, DC / PLM, ideal gas, no passive
// scalars, the two-plane kernel; AKMI_FUSE_C2P=0 switches it off (A/B runs)
static bool hyd_c2p_inside(const Geo &g, const Scheme &sc) {
  static const bool on = !(getenv("AKMI_FUSE_C2P") && atoi(getenv("AKMI_FUSE_C2P")) == 0);
  return on && hyd_two_planes() && g.three_d && sc.recon <= 1 && !sc.iso && g.nvar == 5 && hyd_tile(g.nx1, g.nx2, 2).tw > 0;
}

// cpa: ConsToPrim of the finished cells inside the kernel (new primitives to cpa->w_out), nullptr: update only
template <bool MASS>
static int launch_hydro_stage3d(const Geo &g, const Scheme &sc, const double *w0, const UpdArgs &u,
                                int kA, int kB, hipStream_t st, Mass3 ms, const HydC2P *cpa = nullptr) {
  const bool two = hyd_two_planes();
  const HydTile tl = hyd_tile(g.nx1, g.nx2, two ? 2 : 1);
  if (tl.tw == 0) { set_error("hydro_stage3d: no tile shape"); return AKMI_FAIL; }
  int ckl = march_len((long)tl.n1*tl.n2, kB - kA + 1, g.nmb, ML);
  static const int ckl_env = getenv("AKMI_HS_CKL") ? atoi(getenv("AKMI_HS_CKL")) : 0;     // experiments: pin the chunk length
  if (ckl_env > 0) ckl = ckl_env;                   // (128^3: the rule's 4 sits on the flat bottom, profiles/r05_hydro_host_ab.txt)
  const int nchunk = cdiv(kB - kA + 1, ckl);
  const size_t lds = hyd_lds_doubles(tl.tw, tl.th, two ? 2 : 1)*sizeof(double);
  dim3 grid(tl.n1, tl.n2, nchunk*g.nmb), block(tl.threads);
  int rc = dispatch_scheme_eos<false>(sc, [&](auto R, auto S) {
    constexpr int RV = decltype(R)::value, SV = decltype(S)::value;
    if constexpr (RV <= 1) {
      if constexpr (!MASS && SV < 10) {
        if (cpa) {
          auto kern = k_hydro_stage3d2<RV, SV, false, true>;
          if (ensure_lds((const void *)kern, lds, "hydro_stage3d") != AKMI_COMPLETE) return (int)AKMI_FAIL;
          kern<<<grid, block, lds, st>>>(g, sc.eos, w0, u, kA, kB, nchunk, ckl, tl.tw, tl.th, ms, *cpa);
          return (int)AKMI_COMPLETE;
        }
      }
      if (cpa) { set_error("hydro_stage3d: ConsToPrim inside the kernel is for the ideal gas without passive scalars"); return (int)AKMI_FAIL; }
      if (two) {
        auto kern = k_hydro_stage3d2<RV, SV, MASS, false>;
        if (ensure_lds((const void *)kern, lds, "hydro_stage3d") != AKMI_COMPLETE) return (int)AKMI_FAIL;
        kern<<<grid, block, lds, st>>>(g, sc.eos, w0, u, kA, kB, nchunk, ckl, tl.tw, tl.th, ms, HydC2P{});
      } else {
        auto kern = k_hydro_stage3d<RV, SV, MASS>;
        if (ensure_lds((const void *)kern, lds, "hydro_stage3d") != AKMI_COMPLETE) return (int)AKMI_FAIL;
        kern<<<grid, block, lds, st>>>(g, sc.eos, w0, u, kA, kB, nchunk, ckl, tl.tw, tl.th, ms);
      }
      return (int)AKMI_COMPLETE;
    } else {
      set_error("hydro_stage3d: DC and PLM only");
      return (int)AKMI_FAIL;
    }
  });
  if (rc != AKMI_COMPLETE) return rc;
  AKMI_CHECK_LAUNCH("hydro_stage3d");
  return AKMI_COMPLETE;
}

// ---------------------------------------------------------------------------------------
// c2p on the ghost SHELL only (after the halo exchange): the interior was converted inside
// the slab pipeline.  MODE 0: k-ghost planes (all j,i); 1: j-ghost rows of active planes;
// 2: i-ghost columns of active rows.  Flattened 1-D over the slab so that the 2*ng-wide
// i-slabs do not waste 60 of 64 lanes.
template <bool MHD>
__global__ void __launch_bounds__(256)
k_c2p_shell(Geo g, Eos eos, double *__restrict__ u0, const double *__restrict__ bx1f,
            const double *__restrict__ bx2f, const double *__restrict__ bx3f,
            double *__restrict__ w0, double *__restrict__ bcc0, int *__restrict__ counters) {
  const int MODE = blockIdx.y;                 // slab of the shell, one launch for all three
  const int ng3 = g.three_d ? g.ng : 0, ng2 = g.multi_d ? g.ng : 0;
  int e1, e2, e3;
  if (MODE == 0) { e1 = g.N1; e2 = g.N2; e3 = 2*ng3; }
  else if (MODE == 1) { e1 = g.N1; e2 = 2*ng2; e3 = g.nx3; }
  else { e1 = 2*g.ng; e2 = g.nx2; e3 = g.nx3; }
  const long long per = (long long)e1*e2*e3;
  const long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= per*g.nmb) return;
  const int m = (int)(t/per);
  long long r = t - (long long)m*per;
  const int kk = (int)(r/((long long)e1*e2));
  r -= (long long)kk*e1*e2;
  const int jj = (int)(r/e1);
  const int ii = (int)(r - (long long)jj*e1);
  int i, j, k;
  if (MODE == 0) { i = ii; j = jj; k = kk < ng3 ? kk : g.ke + 1 + (kk - ng3); }
  else if (MODE == 1) { i = ii; j = jj < ng2 ? jj : g.je + 1 + (jj - ng2); k = g.ks + kk; }
  else { i = ii < g.ng ? ii : g.ie + 1 + (ii - g.ng); j = g.js + jj; k = g.ks + kk; }
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const size_t c = ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i);
  double wd, wvx, wvy, wvz, we;
  double ubx = 0.0, uby = 0.0, ubz = 0.0;
  if constexpr (MHD) {
    ubx = 0.5*(bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i)] + bx1f[ix4(g.N3, g.N2, g.N1 + 1, m, k, j, i + 1)]);
    uby = 0.5*(bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j, i)] + bx2f[ix4(g.N3, g.N2 + 1, g.N1, m, k, j + 1, i)]);
    ubz = 0.5*(bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k, j, i)] + bx3f[ix4(g.N3 + 1, g.N2, g.N1, m, k + 1, j, i)]);
    const size_t b = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
    bcc0[b] = ubx; bcc0[b + cs] = uby; bcc0[b + 2*cs] = ubz;
  }
  if (!eos.is_ideal) {
    c2p_iso_cell<MHD>(g, eos, u0, w0, c, cs, ubx, uby, ubz, counters, wd, wvx, wvy, wvz);
    return;
  }
  double ud = u0[c], umx = u0[c + cs], umy = u0[c + 2*cs], umz = u0[c + 3*cs], ue = u0[c + 4*cs];
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
  for (int n = 5; n < g.nvar; ++n) {
    double us = u0[c + n*cs];
    if (us < 0.0) { us = 0.0; u0[c + n*cs] = 0.0; }
    w0[c + n*cs] = us/ud;
  }
}

template <bool MHD>
static int c2p_shell(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                     const double *bx3f, double *w0, double *bcc0, int *counters, hipStream_t st) {
  Geo g = make_geo(p);
  Eos eos = make_eos(p);
  const long long n0 = g.three_d ? (long long)g.N1*g.N2*2*g.ng*g.nmb : 0;
  const long long n1 = g.multi_d ? (long long)g.N1*2*g.ng*g.nx3*g.nmb : 0;
  const long long n2 = (long long)2*g.ng*g.nx2*g.nx3*g.nmb;
  const long long nmax = n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2);
  dim3 grid((unsigned)((nmax + 255)/256), 3);
  k_c2p_shell<MHD><<<grid, 256, 0, st>>>(g, eos, u0, bx1f, bx2f, bx3f, w0, bcc0, counters);
  AKMI_CHECK_LAUNCH("c2p_shell");
  return AKMI_COMPLETE;
}

// passive scalars of the planes [k0, k0+nk-1] (after the sweeps that left the mass fluxes of these cells)
static int launch_scalars(const Geo &g, const Scheme &sc, const double *w0, const double *m1,
                          const double *m2, const double *m3, const UpdArgs &u, int k0, int nk,
                          hipStream_t st) {
  const int nf = sc.iso ? 4 : 5;
  dim3 grid((unsigned)(((long)(g.je - g.js + 1)*g.N1 + SX*SY - 1)/(SX*SY)), 1, nk*g.nmb), block(SX, SY);
  return dispatch_recon(sc.recon, [&](auto R) {
    k_scalar_update<decltype(R)::value><<<grid, block, 0, st>>>(g, sc.eos, w0, m1, m2, m3, u, k0, nk, nf);
    AKMI_CHECK_LAUNCH("scalar_update");
    return AKMI_COMPLETE;
  });
}

template <bool MHD>
static int launch_c2p(const Geo &g, const Eos &eos, double *u0, const double *bx1f,
                      const double *bx2f, const double *bx3f, double *w0, double *bcc0,
                      int do_newdt, int *counters, double *dt3, int il, int iu, int jl, int ju,
                      int k0, int nk, hipStream_t st) {
  dim3 grid((unsigned)(((long)(ju - jl + 1)*g.N1 + SX*SY - 1)/(SX*SY)), 1, nk*g.nmb), block(SX, SY);
#if AKMI_C2P_PAIRS
  static const bool pairs_off = getenv("AKMI_C2P_PAIRS") && atoi(getenv("AKMI_C2P_PAIRS")) == 0;       // A/B switch
  // (from ~4 M cells per launch: at 128^3 = 2.3 M the launch is latency-bound and half the threads lose 5 % of a cycle,
  //  at 192^3 = 7.5 M the pairs gain 2.4 %, at 256^3 3.5 %; profiles/r05_c2p_pairs.txt)
  const long ncell = (long)g.nmb*nk*(ju - jl + 1)*g.N1;
  if (!pairs_off && ncell >= AKMI_C2P_PAIRS_MIN && eos.is_ideal && g.nvar == 5 && !(g.N1 & 1) && !(il & 1) && (iu & 1)) {
    dim3 grid2((unsigned)(((long)(ju - jl + 1)*(g.N1/2) + SX*SY - 1)/(SX*SY)), 1, nk*g.nmb);
    k_c2p_newdt2<MHD><<<grid2, block, 0, st>>>(g, eos, u0, bx1f, bx2f, bx3f, w0, bcc0, do_newdt, counters,
                                               dt3, il, iu, jl, ju, k0, nk);
    AKMI_CHECK_LAUNCH("c2p pairs");
    return AKMI_COMPLETE;
  }
#endif
  k_c2p_newdt<MHD><<<grid, block, 0, st>>>(g, eos, u0, bx1f, bx2f, bx3f, w0, bcc0, do_newdt, counters,
                                           dt3, il, iu, jl, ju, k0, nk);
  AKMI_CHECK_LAUNCH("c2p");
  return AKMI_COMPLETE;
}

// ---------------------------------------------------------------------------------------
// The flux kernels of the task-granular entry points (akmi_hydro_fluxes / akmi_mhd_fluxes: the path of
// refined meshes and of everything the fused stage does not cover) by the sweeps of the fused stage:
// the x1 sweep with the reconstruction of a cell shared between its two faces, the x2/x3 marches in
// MODE 2.  Same outputs (all five flux components + the two face EMFs per direction, same ranges:
// hydro_fluxes.cpp:95-104, mhd_fluxes.cpp:117-248), same arithmetic per face.
template <bool MHD>
static int sweeps_store_fluxes_t(const akmi_pack *p, int recon, int rsolver, const double *w0,
                                 const double *bcc0, const double *bx1f, const double *bx2f, const double *bx3f,
                                 double *flx1, double *flx2, double *flx3, int fsh, double *e3x1, double *e2x1,
                                 double *e1x2, double *e3x2, double *e2x3, double *e1x3, hipStream_t st) {
  Geo g = make_geo(p);
  if ((size_t)(g.N3 + 1)*(g.N2 + 1)*(g.N1 + 1)*sizeof(double) >= ((size_t)1 << 32)) return -1;   // caller falls back
  const Scheme sc{recon, rsolver, make_face_eos(p), !p->is_ideal};
  SweepArgs a1{w0, bcc0, bx1f, flx1, e3x1, e2x1, nullptr, nullptr, nullptr,
               g.is, g.ie + 1, g.js, g.je, g.ks, g.ke, g.N3, g.N2, g.N1 + fsh};
  SweepArgs a2{w0, bcc0, bx2f, flx2, e1x2, e3x2, nullptr, nullptr, nullptr,
               g.is, g.ie, g.js, g.je + 1, g.ks, g.ke, g.N3, g.N2 + fsh, g.N1};
  SweepArgs a3{w0, bcc0, bx3f, flx3, e2x3, e1x3, nullptr, nullptr, nullptr,
               g.is, g.ie, g.js, g.je, g.ks, g.ke + 1, g.N3 + fsh, g.N2, g.N1};
  if (MHD) {
    if (g.multi_d) { a1.jl = g.js - 1; a1.ju = g.je + 1; }
    if (g.three_d) { a1.kl = g.ks - 1; a1.ku = g.ke + 1; }
    a2.il = g.is - 1; a2.iu = g.ie + 1;
    if (g.three_d) { a2.kl = g.ks - 1; a2.ku = g.ke + 1; }
    a3.il = g.is - 1; a3.iu = g.ie + 1; a3.jl = g.js - 1; a3.ju = g.je + 1;
  }
  const UpdArgs u{};
  int rc = launch_sweep<0, MHD, false>(g, sc, a1, st);
#if AKMI_SMALL_FACE_SWEEPS
  // small packs (the deck-size mesh of BASELINE config 5): a march is a chain of dependent steps per thread and a
  // few hundred workgroups; one thread per face has no chain at all
  static const int tf_env = getenv("AKMI_FACE_SWEEPS") ? atoi(getenv("AKMI_FACE_SWEEPS")) : -1;
  const long ncell = (long)g.nmb*g.nx1*g.nx2*g.nx3;
  // ... and, since the face sweeps run their lanes over the flattened planes (x3: over (j; k; i)), packs of MeshBlocks of up
  // to 96 cells per side at any size: x2 / x3 of 960 blocks of 32^3 (PPM4) 1 712 / 1 641 -> 1 563 / 1 622 us, 64 blocks of
  // 64^3 (PLM) 752 / 720 -> 659 / 655, 8 blocks of 96^3 322 / 314 -> 306 / 285; 8 blocks of 128^3 even (642 / 673 against
  // 664 / 668), one block of 256^3 the marches by 13 % (602 / 583 against 679 / 702) -- profiles/r06_lane_mapping.txt
  const int nmax = g.nx1 > g.nx2 ? (g.nx1 > g.nx3 ? g.nx1 : g.nx3) : (g.nx2 > g.nx3 ? g.nx2 : g.nx3);
  if (g.three_d && (tf_env == 1 || (tf_env != 0 && (ncell <= (long)AKMI_SMALL_FACE_SWEEPS || nmax <= 96)))) {
    if (rc == AKMI_COMPLETE) rc = launch_sweep<1, MHD, false>(g, sc, a2, st);
    if (rc == AKMI_COMPLETE) rc = launch_sweep<2, MHD, false>(g, sc, a3, st);
    return rc;
  }
#endif
  if (rc == AKMI_COMPLETE && g.multi_d) rc = launch_sweep_update<1, MHD, 2, false>(g, sc, a2, u, st);
  if (rc == AKMI_COMPLETE && g.three_d) rc = launch_sweep_update<2, MHD, 2, false>(g, sc, a3, u, st);
  return rc;
}
int sweeps_store_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0, const double *bcc0,
                        const double *bx1f, const double *bx2f, const double *bx3f, double *flx1, double *flx2,
                        double *flx3, int face_shaped, double *e3x1, double *e2x1, double *e1x2, double *e3x2,
                        double *e2x3, double *e1x3, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (bcc0)
    return sweeps_store_fluxes_t<true>(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, 1,
                                       e3x1, e2x1, e1x2, e3x2, e2x3, e1x3, st);
  return sweeps_store_fluxes_t<false>(p, recon, rsolver, w0, nullptr, nullptr, nullptr, nullptr, flx1, flx2,
                                      flx3, face_shaped ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      nullptr, st);
}

struct C2PArgs {          // ConsToPrim of the active cells (+ CFL scan) at the end of the stage call
  int enable, do_newdt;
  int *counters;
  double *dt3;
};

// Pass A (+ optionally the interior part of pass B) of one stage, in stream order: the Riemann sweeps with the RK update,
// CornerE + CT, ConsToPrim of the active cells.  `phases` selects the parts, so that a rank with off-rank neighbours can
// post its halo messages in between.
template <bool MHD>
static int stage_update(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                        double beta_dt,
                        int copy_u1, const double *w0, const double *bcc0, double *u0, double *u1,
                        double *b0x1f, double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f,
                        double *b1x3f, void *ws, const C2PArgs &cp_in, hipStream_t st,
                        int phases = AKMI_PHASE_ALL, const double *dt_dev = nullptr, double *w_out = nullptr,
                        int *wrote_new = nullptr) {
  if (check_scheme(p, recon, "stage") != AKMI_COMPLETE) return AKMI_FAIL;
  if (p->nvar < (p->is_ideal ? 5 : 4)) {
    set_error("stage: nvar = %d is smaller than the fluid variable set of the EOS", p->nvar);
    return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  Eos eos = make_eos(p);
  // the sweeps address one variable of one MeshBlock with a 32-bit byte offset per lane
  if ((size_t)(g.N3 + 1)*(g.N2 + 1)*(g.N1 + 1)*sizeof(double) >= ((size_t)1 << 32)) {
    set_error("stage: a MeshBlock of %d x %d x %d cells (with ghosts) exceeds 4 GB per variable; use smaller "
              "MeshBlocks", g.N1, g.N2, g.N3);
    return AKMI_FAIL;
  }
  const Scheme sc{recon, rsolver, make_face_eos(p), !p->is_ideal};
  StageWs w = carve(g, MHD ? 1 : 0, ws);
  // phases: a caller that exchanges halos between the parts of a stage (multi-rank runs) asks
  // for them one at a time; the parts communicate through u0/b0 and the workspace only
  const bool do_sweeps = (phases & AKMI_PHASE_SWEEPS) != 0;
  const bool do_emf = MHD && (phases & AKMI_PHASE_EMF_CT) != 0;
  C2PArgs cp = cp_in;
  if (!(phases & AKMI_PHASE_C2P)) cp.enable = 0;
  const int ndim = g.three_d ? 3 : (g.multi_d ? 2 : 1);
  // dt_dev: beta_dt is the RK weight beta, the kernels multiply it with *dt_dev (akmi_*_stage_fused_dt)
  UpdArgs u{gam0, gam1, beta_dt, u0, u1, w.flx1, w.flx2, copy_u1, w.acc, dt_dev};
  // copy_u1 == 2 (out-of-place first stage): the new state lands in u1 / b1, which is what the c2p
  // of the active cells has to read
  double *un = copy_u1 == 2 ? u1 : u0;
  const double *n1f = copy_u1 == 2 ? b1x1f : b0x1f, *n2f = copy_u1 == 2 ? b1x2f : b0x2f,
               *n3f = copy_u1 == 2 ? b1x3f : b0x3f;
  int rc = AKMI_COMPLETE;
  // sweep ranges: hydro_fluxes.cpp:95-104 (no FOFC) / mhd_fluxes.cpp:117-248 (CT-extended)
  SweepArgs a1{w0, bcc0, b0x1f, w.flx1, w.efc[0], w.efc[1], w.ecc[0], w.ecc[1], w.ecc[2],
               g.is, g.ie + 1, g.js, g.je, g.ks, g.ke, g.N3, g.N2, g.N1 + 1};
  SweepArgs a2{w0, bcc0, b0x2f, w.flx2, w.efc[2], w.efc[3], nullptr, nullptr, nullptr,
               g.is, g.ie, g.js, g.je + 1, g.ks, g.ke, g.N3, g.N2 + 1, g.N1};
  SweepArgs a3{w0, bcc0, b0x3f, w.flx3, w.efc[4], w.efc[5], nullptr, nullptr, nullptr,
               g.is, g.ie, g.js, g.je, g.ks, g.ke + 1, g.N3 + 1, g.N2, g.N1};
  if (MHD) {
    if (g.multi_d) { a1.jl = g.js - 1; a1.ju = g.je + 1; }
    if (g.three_d) { a1.kl = g.ks - 1; a1.ku = g.ke + 1; }
    a2.il = g.is - 1; a2.iu = g.ie + 1;
    if (g.three_d) { a2.kl = g.ks - 1; a2.ku = g.ke + 1; }
    a3.il = g.is - 1; a3.iu = g.ie + 1; a3.jl = g.js - 1; a3.ju = g.je + 1;
  }
  if (cp.enable && cp.do_newdt == 1) k_init_dt3<<<1, 64, 0, st>>>(cp.dt3);       // 2: the caller has reset the minima
  bool c2p_done = false;              // ConsToPrim of the active cells happened inside the sweeps' kernel

  if (ndim < 3) {
    // 1-D / 2-D: small problems, plain sequence on the caller's stream
    if (!do_sweeps) {
    } else if (ndim == 1) {
      rc = launch_sweep_update<0, MHD>(g, sc, a1, u, st);
    } else {
      rc = MHD ? launch_sweep<0, MHD, MHD>(g, sc, a1, st)
               : launch_sweep<0, MHD, false>(g, sc, a1, st);
      if (rc == AKMI_COMPLETE) rc = launch_sweep_update<1, MHD>(g, sc, a2, u, st);
    }
    if (rc == AKMI_COMPLETE && do_sweeps && g.nvar > (sc.iso ? 4 : 5))
      rc = launch_scalars(g, sc, w0, w.flx1, w.flx2, w.flx3, u, g.ks, g.ke - g.ks + 1, st);
    if (rc != AKMI_COMPLETE) return rc;
    if (do_emf) {
      rc = akmi_mhd_corner_e(p, w0, bcc0, w.efc[0], w.efc[1], w.efc[2], w.efc[3], w.efc[4], w.efc[5],
                             w.flx1, w.flx2, w.flx3, w.e1, w.e2, w.e3, st);
      if (rc != AKMI_COMPLETE) return rc;
      const int nkc = g.ke - g.ks + 2;
      long npc = (long)(g.je - g.js + 2)*g.N1;
      dim3 grid((unsigned)((npc + SX*SY - 1)/(SX*SY)), 1, nkc*g.nmb), block(SX, SY);
      k_ct_copy<<<grid, block, 0, st>>>(g, gam0, gam1, beta_dt, w.e1, w.e2, w.e3, b0x1f, b0x2f, b0x3f,
                                        b1x1f, b1x2f, b1x3f, copy_u1, g.ks, nkc, g.ke, dt_dev);
      AKMI_CHECK_LAUNCH("ct");
    }
    if (cp.enable)
      rc = launch_c2p<MHD>(g, eos, un, n1f, n2f, n3f, const_cast<double *>(w0),
                           const_cast<double *>(bcc0), cp.do_newdt, cp.counters, cp.dt3, g.is, g.ie,
                           g.js, g.je, g.ks, g.ke - g.ks + 1, st);
    return rc;
  }

  // ---- 3-D: the VALU-bound sweeps, then the HBM-bound CornerE + CT and ConsToPrim, in stream order.
  // (Cutting the block into k-slabs and running the two groups on two streams one slab apart was built in round 1 and
  //  measured slower at every slab thickness in rounds 1 and 3 -- profiles/r03_slab_ab.txt; it left the source in round 5.)
  // AKMI_HYDRO_ONE_KERNEL=0: the three-kernel sweep/march sequence also for hydro DC/PLM (A/B runs)
  static const bool hyd_one = !(getenv("AKMI_HYDRO_ONE_KERNEL") && atoi(getenv("AKMI_HYDRO_ONE_KERNEL")) == 0);
  // AKMI_MHD_ONE_KERNEL=1: k_mhd_stage3d instead of k_sweep12s + x3 march (A/B runs)
  static const bool mhd_one = getenv("AKMI_MHD_ONE_KERNEL") && atoi(getenv("AKMI_MHD_ONE_KERNEL")) != 0;   // opt-in until it wins
  const int kA = g.ks, kB = g.ke;
  SweepArgs b1 = a1, b2 = a2, b3 = a3;
  b1.kl = kA - (MHD ? 1 : 0); b1.ku = kB + (MHD ? 1 : 0);
  b2.kl = b1.kl; b2.ku = b1.ku;
  b3.kl = kA; b3.ku = kB + 1;
  if (do_sweeps && !MHD && hyd_one && sc.recon <= 1 && hyd_tile(g.nx1, g.nx2).tw > 0) {
    // hydro DC/PLM: sweeps + update in one kernel
    if constexpr (!MHD) {
      if (w_out && cp.enable && hyd_c2p_inside(g, sc)) {
        // (the floor flags of the cells, one byte each, at the start of the workspace: akmi_hydro_ghost_uw reads them there)
        const HydC2P hc{reinterpret_cast<unsigned char *>(ws), w_out, eos, cp.do_newdt, cp.counters, cp.dt3};
        rc = launch_hydro_stage3d<false>(g, sc, w0, u, kA, kB, st, Mass3{nullptr, nullptr, nullptr}, &hc);
        c2p_done = true;
        if (wrote_new) *wrote_new = 1;
      } else {
        rc = g.nvar > (sc.iso ? 4 : 5) ? launch_hydro_stage3d<true>(g, sc, w0, u, kA, kB, st, Mass3{w.flx1, w.flx2, w.flx3})
                        : launch_hydro_stage3d<false>(g, sc, w0, u, kA, kB, st, Mass3{nullptr, nullptr, nullptr});
      }
    }
  } else if (do_sweeps && MHD && mhd_one && sc.recon == 1 && !sc.iso && sc.rsolver == AKMI_RS_HLLD && g.nvar == 5 &&
             mhd_tile(g.nx1 + 2, g.nx2 + 2).tw > 0) {
    // MHD PLM + HLLD: the three sweeps + update in one kernel (k_mhd_stage3d)
    if constexpr (MHD) {
      const MhdStageArgs ma{w0, bcc0, b0x1f, b0x2f, b0x3f, w.flx1, w.flx2, w.flx3, w.efc[0], w.efc[1], w.efc[2], w.efc[3],
                            w.efc[4], w.efc[5], w.ecc[0], w.ecc[1], w.ecc[2]};
      rc = launch_mhd_stage3d(g, sc, ma, u, st);
    }
  } else if (do_sweeps && MHD && sc.recon == 1 && !sc.iso && sc.rsolver == AKMI_RS_HLLD && g.nvar == 5) {
    // x1 sweep inside the x2 march, cells from the march's window (k_sweep12s); x3 march consumes acc
    // (the other order -- x3 march first, leaving dF3/dx3, k_sweep12s finishing the update -- was built and measured in
    //  round 4: x3 march 891 -> 602 us, k_sweep12s 1130 -> 1374 us, +1 % on the bench; profiles/r04_x3first.txt)
    if constexpr (MHD) rc = launch_sweep12s(g, sc, b1, b2, u, st);
    if (rc == AKMI_COMPLETE) rc = launch_sweep_update<2, MHD, 0, true>(g, sc, b3, u, st);
  } else if (do_sweeps) {
    rc = MHD ? launch_sweep<0, MHD, MHD>(g, sc, b1, st)
             : launch_sweep<0, MHD, false>(g, sc, b1, st);
    // x2 sweep as a march along j that leaves acc = dF1/dx1 + dF2/dx2; x3 march consumes it
    if (rc == AKMI_COMPLETE) rc = launch_sweep_update<1, MHD, 1, false>(g, sc, b2, u, st);
    if (rc == AKMI_COMPLETE) rc = launch_sweep_update<2, MHD, 0, true>(g, sc, b3, u, st);
  }
  if (rc == AKMI_COMPLETE && do_sweeps && g.nvar > (sc.iso ? 4 : 5))
    rc = launch_scalars(g, sc, w0, w.flx1, w.flx2, w.flx3, u, kA, kB - kA + 1, st);
  if (rc != AKMI_COMPLETE) return rc;
  if (MHD && do_emf) {
    // CornerE + CT in one kernel (corner EMFs of the planes [ks, ke+1] stay on chip)
    const CtTile tl = ct_tile(g.nx1 + 1, g.nx2 + 1);
    const int ckl = march_len((long)tl.n1*tl.n2, kB - kA + 1, g.nmb, CKL);
    const int nchunk = cdiv(kB - kA + 1, ckl);
    dim3 grid(tl.n1, tl.n2, nchunk*g.nmb), block(tl.threads);
    k_corner_ct<<<grid, block, 7*tl.tw*tl.th*sizeof(double), st>>>(
        g, w.efc[0], w.efc[1], w.efc[2], w.efc[3], w.efc[4], w.efc[5], w.ecc[0], w.ecc[1], w.ecc[2],
        w.flx1, w.flx2, w.flx3, gam0, gam1, beta_dt, b0x1f, b0x2f, b0x3f, b1x1f, b1x2f, b1x3f,
        copy_u1, kA, kB, 1, nchunk, ckl, tl.tw, tl.th, dt_dev);
    AKMI_CHECK_LAUNCH("corner_ct");
  }
  if (cp.enable && !c2p_done)
    rc = launch_c2p<MHD>(g, eos, un, n1f, n2f, n3f, const_cast<double *>(w0),
                         const_cast<double *>(bcc0), cp.do_newdt, cp.counters, cp.dt3, g.is, g.ie,
                         g.js, g.je, kA, kB - kA + 1, st);
  return rc;
}

}  // namespace akmi

extern "C" {

const char *akmi_build_flags(void) {
#ifdef AKMI_EXPERIMENTS
  return "experiments";
#else
  return "production";
#endif
}

long long akmi_stage_workspace_bytes(const akmi_pack *p, int is_mhd) {
  Geo g = make_geo(p);
  return (long long)carve(g, is_mhd, nullptr).total;
}

int akmi_hydro_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                            double beta_dt, int copy_u1, const double *w0, double *u0, double *u1,
                            void *ws, void *stream) {
  C2PArgs cp{0, 0, nullptr, nullptr};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, nullptr, u0, u1, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, ws, cp, (hipStream_t)stream);
}

int akmi_mhd_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                          double beta_dt, int copy_u1, const double *w0, const double *bcc0,
                          double *u0, double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                          double *b1x1f, double *b1x2f, double *b1x3f, void *ws, void *stream) {
  C2PArgs cp{0, 0, nullptr, nullptr};
  return stage_update<true>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, bcc0, u0, u1, b0x1f, b0x2f,
                            b0x3f, b1x1f, b1x2f, b1x3f, ws, cp, (hipStream_t)stream);
}

int akmi_hydro_c2p_newdt(const akmi_pack *p, double *u0, double *w0, int do_newdt, int *counters,
                         double *dt3, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  if (do_newdt == 1) k_init_dt3<<<1, 64, 0, st>>>(dt3);
  return launch_c2p<false>(g, make_eos(p), u0, nullptr, nullptr, nullptr, w0, nullptr, do_newdt,
                           counters, dt3, 0, g.N1 - 1, 0, g.N2 - 1, 0, g.N3, st);
}

int akmi_mhd_c2p_newdt(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                       const double *bx3f, double *w0, double *bcc0, int do_newdt, int *counters,
                       double *dt3, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  if (do_newdt == 1) k_init_dt3<<<1, 64, 0, st>>>(dt3);
  return launch_c2p<true>(g, make_eos(p), u0, bx1f, bx2f, bx3f, w0, bcc0, do_newdt, counters, dt3,
                          0, g.N1 - 1, 0, g.N2 - 1, 0, g.N3, st);
}

int akmi_hydro_stage_fused(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                           double beta_dt, int copy_u1, double *w0, double *u0, double *u1,
                           int do_newdt, int *counters, double *dt3, void *ws, void *stream) {
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, nullptr, u0, u1, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, ws, cp, (hipStream_t)stream);
}

int akmi_hydro_stage_fused_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                              double beta, const double *dt_dev, int copy_u1, double *w0, double *u0,
                              double *u1, int do_newdt, int *counters, double *dt3, void *ws, void *stream) {
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta, copy_u1, w0, nullptr, u0, u1, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, ws, cp, (hipStream_t)stream,
                             AKMI_PHASE_ALL, dt_dev);
}

int akmi_mhd_stage_fused_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                            double beta, const double *dt_dev, int copy_u1, double *w0, double *bcc0,
                            double *u0, double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                            double *b1x1f, double *b1x2f, double *b1x3f, int do_newdt, int *counters,
                            double *dt3, void *ws, void *stream) {
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<true>(p, recon, rsolver, gam0, gam1, beta, copy_u1, w0, bcc0, u0, u1, b0x1f, b0x2f,
                            b0x3f, b1x1f, b1x2f, b1x3f, ws, cp, (hipStream_t)stream, AKMI_PHASE_ALL, dt_dev);
}

int akmi_mhd_stage_fused(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                         double beta_dt, int copy_u1, double *w0, double *bcc0, double *u0, double *u1,
                         double *b0x1f, double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f,
                         double *b1x3f, int do_newdt, int *counters, double *dt3, void *ws,
                         void *stream) {
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<true>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, bcc0, u0, u1, b0x1f, b0x2f,
                            b0x3f, b1x1f, b1x2f, b1x3f, ws, cp, (hipStream_t)stream);
}

int akmi_hydro_stage_phase(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                           double beta_dt, int copy_u1, double *w0, double *u0, double *u1,
                           int do_newdt, int *counters, double *dt3, int phases, void *ws,
                           void *stream) {
  if (phases <= 0 || phases > AKMI_PHASE_ALL) { set_error("stage_phase: bad phase mask"); return AKMI_FAIL; }
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, nullptr, u0, u1, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, ws, cp,
                             (hipStream_t)stream, phases);
}

int akmi_mhd_stage_phase(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                         double beta_dt, int copy_u1, double *w0, double *bcc0, double *u0, double *u1,
                         double *b0x1f, double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f,
                         double *b1x3f, int do_newdt, int *counters, double *dt3, int phases,
                         void *ws, void *stream) {
  if (phases <= 0 || phases > AKMI_PHASE_ALL) { set_error("stage_phase: bad phase mask"); return AKMI_FAIL; }
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<true>(p, recon, rsolver, gam0, gam1, beta_dt, copy_u1, w0, bcc0, u0, u1, b0x1f, b0x2f,
                            b0x3f, b1x1f, b1x2f, b1x3f, ws, cp, (hipStream_t)stream, phases);
}

int akmi_hydro_stage_phase_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                              const double *dt_dev, int copy_u1, double *w0, double *u0, double *u1, int do_newdt,
                              int *counters, double *dt3, int phases, void *ws, void *stream) {
  if (phases <= 0 || phases > AKMI_PHASE_ALL) { set_error("stage_phase: bad phase mask"); return AKMI_FAIL; }
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta, copy_u1, w0, nullptr, u0, u1, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, ws, cp, (hipStream_t)stream, phases, dt_dev);
}

int akmi_mhd_stage_phase_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                            const double *dt_dev, int copy_u1, double *w0, double *bcc0, double *u0, double *u1,
                            double *b0x1f, double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f, double *b1x3f,
                            int do_newdt, int *counters, double *dt3, int phases, void *ws, void *stream) {
  if (phases <= 0 || phases > AKMI_PHASE_ALL) { set_error("stage_phase: bad phase mask"); return AKMI_FAIL; }
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<true>(p, recon, rsolver, gam0, gam1, beta, copy_u1, w0, bcc0, u0, u1, b0x1f, b0x2f,
                            b0x3f, b1x1f, b1x2f, b1x3f, ws, cp, (hipStream_t)stream, phases, dt_dev);
}

int akmi_hydro_stage_w_eligible(const akmi_pack *p, int recon, int rsolver) {
  Geo g = make_geo(p);
  const Scheme sc{recon, rsolver, make_face_eos(p), !p->is_ideal};
  return (rsolver == AKMI_RS_LLF || rsolver == AKMI_RS_HLLE || rsolver == AKMI_RS_HLLC || rsolver == AKMI_RS_ROE) &&
         hyd_c2p_inside(g, sc) ? 1 : 0;
}

int akmi_hydro_stage_w(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                       const double *dt_dev, int copy_u1, double *w0, double *w0_new, double *u0, double *u1,
                       int do_newdt, int *counters, double *dt3, void *ws, void *stream, int *wrote_new) {
  if (wrote_new) *wrote_new = 0;
  C2PArgs cp{1, do_newdt, counters, dt3};
  return stage_update<false>(p, recon, rsolver, gam0, gam1, beta, copy_u1, w0, nullptr, u0, u1, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, ws, cp, (hipStream_t)stream, AKMI_PHASE_ALL, dt_dev, w0_new,
                             wrote_new);
}

int akmi_hydro_c2p_shell(const akmi_pack *p, double *u0, double *w0, int *counters, void *stream) {
  return c2p_shell<false>(p, u0, nullptr, nullptr, nullptr, w0, nullptr, counters, (hipStream_t)stream);
}

int akmi_mhd_c2p_shell(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                       const double *bx3f, double *w0, double *bcc0, int *counters, void *stream) {
  return c2p_shell<true>(p, u0, bx1f, bx2f, bx3f, w0, bcc0, counters, (hipStream_t)stream);
}

}  // extern "C"
