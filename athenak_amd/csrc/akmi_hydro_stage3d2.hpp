// akmi_hydro_stage3d2.hpp -- k_hydro_stage3d with TWO planes of the primitives in LDS (round 6; included by akmi_stage.hip
// behind k_hydro_stage3d, whose tile chooser, LDS entry layout and launch parameters it shares).
//
// Same decomposition and the same arithmetic per value as k_hydro_stage3d (hydro_fluxes.cpp:44-170, hydro_update.cpp:24-84):
// a tile of (tw-1) x (th-1) cell columns marching along k, every cell reconstructed once per in-plane direction, x1 / x2
// faces from the plane in LDS, the flux taking the place of the left state in the same LDS slot, x3 face per position,
// update in the kernel.  What changes is where the marching state lives and when memory is touched -- the findings of
// k_mhd_stage3d (profiles/r06_mhd_stage3d.txt) applied to the kernel that has the registers for three waves per SIMD:
//   * planes k-1 AND k sit in LDS (plane k+1, loaded during the step, replaces plane k-1 once its readers are past the
//     in-plane reconstruction): the cells k-1 and k of the position's own column are read from there for the x3 face, so
//     the ten doubles W0 / W1 leave the register file -- no scratch (the one-plane kernel saves and restores three doubles
//     per step through scratch, each reload an `s_waitcnt vmcnt(0)`);
//   * every global load is unconditional (lanes without a cell read element 0 of the plane, planes are clamped into the
//     array): a load inside a branch turns the wait for it into a wait for everything at the join;
//   * the loads of plane k+1 are issued after the x1 solve and consumed after the x2 solve, the operands of the update
//     after the x2 solve and consumed after the x3 solve; the stores of a step (u0, mass fluxes) all come after its last
//     wait for a load (with loads and stores both outstanding the compiler's wait for a load is `vmcnt(0)`).
#ifndef AKMI_HS2_WAVES
#define AKMI_HS2_WAVES 3
#endif
static size_t hyd2_lds_doubles(int tw, int th) { return HS_ES*(2*(size_t)(tw + 3)*(th + 3) + 2*(size_t)tw*th); }

// C2P: ConsToPrim of the cell the step finishes (+ the CFL scan on the last stage) in the same kernel -- the new conserved
// state is in registers there, so pass B of the active cells costs five stores instead of a kernel that reads u0 back
// (SingleC2P_IdealHyd, src/eos/ideal_c2p_hyd.hpp:22-66 with its floors and counters; hydro_newdt.cpp:97-118).  The new
// primitives go to ANOTHER array of w0's shape (w_out): neighbouring tiles and chunks still read w0 while this one finishes,
// in place is not possible; the caller swaps the two afterwards and converts the ghost shell after the ghost fill
// (akmi_hydro_c2p_shell).  Ideal gas, no passive scalars.
struct HydC2P {
  unsigned char *flags;       // per cell (m, k, j, i): bit 0 / 1 / 2 = the density / energy / temperature floor acted -- the ghost
                              // fill that follows counts the ghost images of such cells (the reference's counters include them)
  double *w_out;
  Eos eos;
  int do_newdt;
  int *counters;
  double *dt3;
};
template <int RECON, int RS, bool MASS = false, bool C2P = false>
__global__ void __launch_bounds__(HS_THREADS, AKMI_HS2_WAVES)
k_hydro_stage3d2(Geo g, FaceEos eos, const double *__restrict__ w0, UpdArgs u, int kA, int kB,
                 int nchunk, int ckl, int tw, int th, Mass3 ms, HydC2P cp) {
  static_assert(!C2P || (!MASS && RS < 10), "ConsToPrim inside the stage kernel: ideal gas, no passive scalars");
  static_assert(RECON <= 1, "one-kernel hydro stage: DC and PLM");
  constexpr bool ISO = rs_iso<RS>();        // isothermal: variable 4 (energy) does not exist; its slots stay unused
#define ISOSKIP if (ISO && n == 4) continue
  extern __shared__ double hs_lds[];
  const int pw = tw + 3, ph = th + 3;        // plane with halo: cols i0-2..i0+tw, rows j0-2..j0+th
  const int qn = ph*pw, fn = th*tw;
  constexpr int ES = HS_ES;
  const int PLSZ = ES*qn;                    // doubles per plane: two planes, then the two face arrays
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
  const int hj = j0 - 2 + hy, hi = i0 - 2 + hx;
  const bool hload = hy >= 0 && hj < g.N2 && hi < g.N1;      // the halo cell exists in memory (i0 - 2, j0 - 2 >= 0: ng >= 2)
  // the cells just outside the tile's low sides: column 1 of the plane (rows of the tile) and row 1 (columns of the tile),
  // reconstructed by the first th + tw threads, one cell each, in the one direction in which a face of the tile needs them
  int ha = -1, hs = 0, hd = 0;                 // entry below the cell, stride to the cell / the entry above, destination
  if (tid < th) { ha = (tid + 2)*pw*ES; hs = ES; hd = 2*PLSZ + tid*tw*ES; }
  else if (tid < th + tw) { const int c = tid - th; ha = (c + 2)*ES; hs = ES*pw; hd = 2*PLSZ + ES*fn + c*ES; }
  // own entries: cell (r+2, t+2) of a plane, position (r, t) of the two face arrays
  const int qo = in_tile ? ((r + 2)*pw + t + 2)*ES : 0, qy = ES*pw;
  const int xo = in_tile ? 2*PLSZ + (r*tw + t)*ES : 2*PLSZ, x2o = xo + ES*fn, xy = ES*tw;
  const int hq = hy >= 0 ? (hy*pw + hx)*ES : -1;
  // scalar base of (block, variable) + a 32-bit byte offset per lane, advanced by one plane per step
  unsigned oc = (cell_ok ? ((unsigned)j*(unsigned)g.N1 + (unsigned)i)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps*8u;      // plane k-1
  unsigned oh = (hload ? ((unsigned)hj*(unsigned)g.N1 + (unsigned)hi)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps*8u;
  const size_t mb = (size_t)m*g.nvar*cs;
  double *u0m = u.u0 + mb, *u1m = u.u1 + mb;
  const double *u1s = u.copy_u1 ? u0m : u1m;             // first stage: the second register is not read (the load repeats u0's)
  // Plane kk lives in slot (kk - k0 + 1) & 1.
  double PL[5], F3p[5];
  {
    double qa[5], qb[5], qc[5], qh[5];
    const long sa = k0 >= 2 ? -(long)ps : 0;             // plane k0-2 (k0-1 where it does not exist: the value is not used)
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      ISOSKIP;
      const double *q = wb + n*cs;
      qa[n] = ldu(q + sa, oc); qb[n] = ldu(q, oc); qc[n] = ldu(q + ps, oc); qh[n] = ldu(q + ps, oh);
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      ISOSKIP;
      if constexpr (RECON == 1) { double dummy; plm(qa[n], qb[n], qc[n], PL[n], dummy); }
      else PL[n] = qb[n];
      if (in_tile) { hs_lds[qo + n] = qb[n]; hs_lds[PLSZ + qo + n] = qc[n]; }      // own entries of planes k0-1 and k0
      if (hq >= 0) hs_lds[PLSZ + hq + n] = qh[n];
    }
  }
#pragma unroll
  for (int n = 0; n < 5; ++n) F3p[n] = 0.0;
  double mv1 = 0.0, mv2 = 0.0, mv3 = 0.0;           // C2P, last stage: running maxima of |v| + c_s over the thread's cells
  double *wom = C2P ? cp.w_out + (size_t)m*g.nvar*cs : nullptr;
  // step k: x3 face k (below cell k); for k > k0 also the x1/x2 faces of plane k-1, which finishes cell k-1;
  // plane k+1 (loaded during the step) replaces plane k-1
  for (int k = k0; k <= k1 + 1; ++k) {
    const bool plane = k > k0;                        // workgroup-uniform
    const long s2 = (k + 1 < g.N3) ? 2*(long)ps : (long)ps;            // plane k+1 relative to plane k-1 (clamped into the array)
    const bool upd = plane && own;
    const int pP = ((k - k0) & 1) ? PLSZ : 0, pC = PLSZ - pP;          // slots of plane k-1 and of plane k
    const int qP = qo + pP, qC = qo + pC;
    double f1[5] = {0, 0, 0, 0, 0}, f2[5] = {0, 0, 0, 0, 0};      // this position's own in-plane fluxes
    double wp[5] = {0, 0, 0, 0, 0}, hv[5] = {0, 0, 0, 0, 0};
    if (!plane) {
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; wp[n] = ldu(wb + n*cs + s2, oc); hv[n] = ldu(wb + n*cs + s2, oh); }
    } else {
      // (A) every cell of plane k-1 once per direction
      double R1[5] = {0, 0, 0, 0, 0}, R2[5] = {0, 0, 0, 0, 0};
      if (in_tile) {
        double W0[5], Wl[5], Wr[5], Wd[5], Wu[5], U1[5] = {0, 0, 0, 0, 0}, U2[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          ISOSKIP;
          W0[n] = hs_lds[qP + n];
          if constexpr (RECON == 1) {
            Wl[n] = hs_lds[qP - ES + n]; Wr[n] = hs_lds[qP + ES + n];
            Wd[n] = hs_lds[qP - qy + n]; Wu[n] = hs_lds[qP + qy + n];
          }
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          ISOSKIP;
          if constexpr (RECON == 1) {
            plm(Wl[n], W0[n], Wr[n], U1[n], R1[n]);
            plm(Wd[n], W0[n], Wu[n], U2[n], R2[n]);
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
        const int hA = ha + pP;
        double ea[5], eb[5], ec[5];              // all three cells requested before the first limiter
#pragma unroll
        for (int n = 0; n < 5; ++n) { ISOSKIP; ea[n] = hs_lds[hA + n]; eb[n] = hs_lds[hA + hs + n]; ec[n] = hs_lds[hA + 2*hs + n]; }
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          ISOSKIP;
          if constexpr (RECON == 1) {
            double up, dummy;
            plm(ea[n], eb[n], ec[n], up, dummy);
            hs_lds[hd + n] = up;
          } else {
            hs_lds[hd + n] = eb[n];
          }
        }
      }
      __syncthreads();
      // (B) the two in-plane faces of this position; the flux takes the place of the left state
      if (in_tile) {
        double fd, fx, fy, fz, fe;
        riemann_hyd_e<RS, true>(eos, hs_lds[xo], hs_lds[xo + 1], hs_lds[xo + 2], hs_lds[xo + 3], ISO ? 0.0 : hs_lds[xo + 4],
                                R1[0], R1[1], R1[2], R1[3], R1[4], fd, fx, fy, fz, fe);
        hs_lds[xo] = fd; hs_lds[xo + 1] = fx; hs_lds[xo + 2] = fy; hs_lds[xo + 3] = fz;
        if constexpr (!ISO) hs_lds[xo + 4] = fe;
        f1[0] = fd; f1[1] = fx; f1[2] = fy; f1[3] = fz; f1[4] = fe;
      }
      __builtin_amdgcn_sched_barrier(0);
      // cell k+1 of the column and of the thread's halo entry: in flight during the x2 solve
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; wp[n] = ldu(wb + n*cs + s2, oc); hv[n] = ldu(wb + n*cs + s2, oh); }
      __builtin_amdgcn_sched_barrier(0);
      if (in_tile) {  // sweep-aligned order (d, vy, vz, vx, e)
        double fd, fx, fy, fz, fe;
        riemann_hyd_e<RS, true>(eos, hs_lds[x2o], hs_lds[x2o + 2], hs_lds[x2o + 3], hs_lds[x2o + 1], ISO ? 0.0 : hs_lds[x2o + 4],
                                R2[0], R2[2], R2[3], R2[1], R2[4], fd, fx, fy, fz, fe);
        hs_lds[x2o] = fd; hs_lds[x2o + 2] = fx; hs_lds[x2o + 3] = fy; hs_lds[x2o + 1] = fz;
        if constexpr (!ISO) hs_lds[x2o + 4] = fe;
        f2[0] = fd; f2[2] = fx; f2[3] = fy; f2[1] = fz; f2[4] = fe;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // x3 face below cell k: sweep-aligned order (d, vz, vx, vy, e).  Cells k-1 and k are the position's entries of the two
    // planes; plane k+1 then takes the place of plane k-1 (every reader of it is past (A))
    double L[5] = {0, 0, 0, 0, 0}, R[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      ISOSKIP;
      if constexpr (RECON == 1) {
        double qln;
        L[n] = PL[n];
        plm(hs_lds[qP + n], hs_lds[qC + n], wp[n], qln, R[n]);
        PL[n] = qln;
      } else {
        L[n] = hs_lds[qP + n]; R[n] = hs_lds[qC + n];
      }
    }
    if (in_tile) {
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[qP + n] = wp[n]; }
    }
    if (hq >= 0) {
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; hs_lds[hq + pP + n] = hv[n]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    double pu0[5] = {0, 0, 0, 0, 0}, pu1[5] = {0, 0, 0, 0, 0};
    // operands of the update: in flight during the x3 solve
#pragma unroll
    for (int n = 0; n < 5; ++n) { ISOSKIP; pu0[n] = ldu(u0m + n*cs, oc); pu1[n] = ldu(u1s + n*cs, oc); }
    __builtin_amdgcn_sched_barrier(0);
    double f3[5];
    {
      double fd, fx, fy, fz, fe;
      riemann_hyd_e<RS, true>(eos, L[0], L[3], L[1], L[2], L[4], R[0], R[3], R[1], R[2], R[4], fd, fx, fy, fz, fe);
      f3[0] = fd; f3[3] = fx; f3[1] = fy; f3[2] = fz; f3[4] = fe;
    }
    __syncthreads();
    if (upd) {                                         // (C) finish cell k-1
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
      double un[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; un[n] = u.gam0*pu0[n] + u.gam1*pu1[n] - bdt*divf[n]; }
      if constexpr (C2P) {
        double wd, wvx, wvy, wvz, we;
        bool dfl = false, efl = false, tfl = false;
        c2p_hyd<true>(cp.eos, un[0], un[1], un[2], un[3], un[4], wd, wvx, wvy, wvz, we, dfl, efl, tfl);   // floors rewrite un[0] / un[4]
        if (dfl) atomicAdd(&cp.counters[0], 1);
        if (efl) atomicAdd(&cp.counters[1], 1);
        if (tfl) atomicAdd(&cp.counters[2], 1);
        cp.flags[(size_t)m*cs + (oc >> 3)] = (unsigned char)((dfl ? 1 : 0) | (efl ? 2 : 0) | (tfl ? 4 : 0));
        stu(wom, oc, wd); stu(wom + cs, oc, wvx); stu(wom + 2*cs, oc, wvy); stu(wom + 3*cs, oc, wvz); stu(wom + 4*cs, oc, we);
        if (cp.do_newdt) {                             // hydro_newdt.cpp:97-118
          const double pr = (cp.eos.gamma - 1.0)*we;
          const double cs_ = sqrt(cp.eos.gamma*pr/wd);
          mv1 = fmax(mv1, fabs(wvx) + cs_); mv2 = fmax(mv2, fabs(wvy) + cs_); mv3 = fmax(mv3, fabs(wvz) + cs_);
        }
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) { ISOSKIP; rk_store_u(u0m + n*cs, u1m + n*cs, u.copy_u1, oc, pu0[n], un[n]); }
    }
    if constexpr (MASS) {                              // passive scalars ride on the mass fluxes (k_scalar_update)
      if (plane && in_tile) {
        if (i <= g.ie + 1 && j <= g.je) ms.m1[ix5(g.nvar, g.N3, g.N2, g.N1 + 1, m, 0, k - 1, j, i)] = f1[0];
        if (i <= g.ie && j <= g.je + 1) ms.m2[ix5(g.nvar, g.N3, g.N2 + 1, g.N1, m, 0, k - 1, j, i)] = f2[0];
      }
      if (own) ms.m3[ix5(g.nvar, g.N3 + 1, g.N2, g.N1, m, 0, k, j, i)] = f3[0];
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) { ISOSKIP; F3p[n] = f3[n]; }
    oc += (unsigned)ps*8u; oh += (unsigned)ps*8u;
  }
  if constexpr (C2P) {
    if (cp.do_newdt) {         // uniform across the grid: wave maxima, workgroup maxima through LDS, one division + filtered
      __syncthreads();         // atomicMin per direction (min over cells of fl(dx/a) == fl(dx/max a): division is monotone)
      mv1 = wave_max(mv1); mv2 = wave_max(mv2); mv3 = wave_max(mv3);
      const int wv = tid >> 6, nwv = (int)(blockDim.x >> 6);
      if ((tid & 63) == 0) { hs_lds[wv] = mv1; hs_lds[8 + wv] = mv2; hs_lds[16 + wv] = mv3; }
      __syncthreads();
      if (tid < 3) {
        double v = hs_lds[8*tid];
        for (int q = 1; q < nwv; ++q) v = fmax(v, hs_lds[8*tid + q]);
        if (v > 0.0) {
          const double d = g.dx[3*m + tid]/v;
          if (d < __hip_atomic_load(&cp.dt3[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(reinterpret_cast<unsigned long long *>(&cp.dt3[tid]), (unsigned long long)__double_as_longlong(d));
        }
      }
    }
  }
#undef ISOSKIP
}
