// akmi_numerics.hpp -- per-cell / per-face device arithmetic of the MeshBlock update.
//
// Written for gfx950 (wave64, fp64 VALU).  Everything here is register-only: a face's L/R
// states are reconstructed from the cell stencil in registers and handed straight to the
// Riemann solver, so the reference's global L/R buffers (src/hydro/hydro.hpp:102-113,
// src/mhd/mhd.hpp:134-137) never exist.  Operation order and parenthesisation follow the
// cited reference lines so a -ffp-contract=off build is bit-comparable with the CPU path.
#ifndef AKMI_NUMERICS_HPP_
#define AKMI_NUMERICS_HPP_
#include <hip/hip_runtime.h>
#include <cfloat>

namespace akmi {

#define AKMI_DEV __device__ __forceinline__

AKMI_DEV double sqr(double x) { return x*x; }

// ---------------------------------------------------------------------------------------
// Correctly rounded fp64 sqrt(x) and 1/x without the range handling of the compiler's expansions.
//
// hipcc expands `sqrt(x)` to 18 VALU instructions: v_rsq_f64, one Goldschmidt step, two residual
// corrections (10 instructions) wrapped in a 2^256 pre-/post-scaling for x < 2^-767 (v_cmp, 2 v_cndmask,
// 2 v_ldexp) and a zero/inf pass-through (v_cmp_class, 2 v_cndmask).  `1.0/x` becomes 11: two
// v_div_scale, v_rcp_f64, two Newton steps, the quotient, its residual, v_div_fmas, v_div_fixup.
// (profiles/r03_isa_audit.txt.)  For operands well inside the normal range the scaling is the
// identity, v_div_fmas is a plain fma and the fix-ups pass the value through, so the same
// iterations WITHOUT them return the same bits.  sqrt_x/rcp_x run that core when every lane of the
// wave holds an operand with 2^-700 <= |x| < 2^700 (one integer add and one compare on the high
// word) and take the compiler's full expansion otherwise (wave-uniform branch): zero, subnormal,
// huge, infinite, NaN and negative operands never reach the short form.  Bit equality with `sqrt`
// and `/` is asserted on >= 1e9 random and edge operands by tests/test_gpu_fastmath.py.
// ---------------------------------------------------------------------------------------
#ifndef AKMI_FAST_SQRT
#define AKMI_FAST_SQRT 1
#endif
#ifndef AKMI_FAST_RCP
#define AKMI_FAST_RCP 1
#endif
#ifndef AKMI_HLLD_FAST_RCP
#define AKMI_HLLD_FAST_RCP 0   // rcp_x in hlld<.., FM>: measured neutral in k_sweep12s (1099 vs 1109 us) at 8 B of scratch: off
#endif
AKMI_DEV unsigned hi_word(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }
// true when 2^-700 <= |x| < 2^700: biased exponent in [323, 1723), sign shifted out
AKMI_DEV bool in_core_range(double x) {
  return ((hi_word(x) << 1) - (323u << 21)) < (1400u << 21);
}
AKMI_DEV double sqrt_core(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x*y;
  double h = y*0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
AKMI_DEV double rcp_core(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(e, r, r);
}
template <bool FM = true>
AKMI_DEV double sqrt_x(double x) {
  if constexpr (!FM) return sqrt(x);
#if AKMI_FAST_SQRT
  // positive only: a set sign bit puts the high word outside the window as well
  const bool ok = (hi_word(x) - (323u << 20)) < (1400u << 20);
  if (__builtin_expect(__any(!ok), 0)) return sqrt(x);
  return sqrt_core(x);
#else
  return sqrt(x);
#endif
}
template <bool FM = true>
AKMI_DEV double rcp_x(double x) {
  if constexpr (!FM) return 1.0/x;
#if AKMI_FAST_RCP
  if (__builtin_expect(__any(!in_core_range(x)), 0)) return 1.0/x;
  return rcp_core(x);
#else
  return 1.0/x;
#endif
}

// PLM, src/reconstruct/plm.hpp:20-37 (van-Leer/harmonic slope on primitives)
AKMI_DEV void plm(double qm, double q, double qp, double &ql_ip1, double &qr_i) {
  double dql = (q - qm);
  double dqr = (qp - q);
  double dq2 = dql*dqr;
  double dqm = dq2/(dql + dqr);
  if (dq2 <= 0.0) dqm = 0.0;
  ql_ip1 = q + dqm;
  qr_i = q - dqm;
}

// PPM4, src/reconstruct/ppm.hpp:44-77 (Colella-Woodward limiters)
AKMI_DEV void ppm4(double qm2, double qm1, double q, double qp1, double qp2, double &ql_ip1,
                   double &qr_i) {
  double qlv = (7.*(q + qm1) - (qm2 + qp1))/12.0;
  double qrv = (7.*(q + qp1) - (qm1 + qp2))/12.0;
  qlv = fmax(qlv, fmin(q, qm1));
  qlv = fmin(qlv, fmax(q, qm1));
  qrv = fmax(qrv, fmin(q, qp1));
  qrv = fmin(qrv, fmax(q, qp1));
  double qc = qrv - q;
  double qd = qlv - q;
  if ((qc*qd) >= 0.0) {
    qlv = q;
    qrv = q;
  } else {
    if (fabs(qc) >= 2.0*fabs(qd)) qrv = q - 2.0*qd;
    if (fabs(qd) >= 2.0*fabs(qc)) qlv = q - 2.0*qc;
  }
  ql_ip1 = qrv;
  qr_i = qlv;
}

AKMI_DEV double sgn(double x) { return (x < 0.0) ? -1.0 : 1.0; }   // SIGN, src/athena.hpp:52

// PPMX, src/reconstruct/ppm.hpp:84-181 (Colella & Sekora limiters, PH = Peterson & Hammett)
AKMI_DEV void ppmx(double qm2, double qm1, double q, double qp1, double qp2, double &ql_ip1,
                   double &qr_i) {
  double qlv = (7.*(q + qm1) - (qm2 + qp1))/12.0;
  double qrv = (7.*(q + qp1) - (qm1 + qp2))/12.0;
  double d2qc = 3.0*((qm1 + q) - 2.0*qlv);
  double d2ql = (qm2 + q) - 2.0*qm1;
  double d2qr = (qm1 + qp1) - 2.0*q;
  double d2qlim = 0.0;
  double lim_slope = fmin(fabs(d2ql), fabs(d2qr));
  if (d2qc > 0.0 && d2ql > 0.0 && d2qr > 0.0) d2qlim = sgn(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (d2qc < 0.0 && d2ql < 0.0 && d2qr < 0.0) d2qlim = sgn(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (((qm1 - qlv)*(q - qlv)) > 0.0) qlv = 0.5*(q + qm1) - d2qlim/6.0;
  d2qc = 3.0*((q + qp1) - 2.0*qrv);
  d2ql = d2qr;
  d2qr = (q + qp2) - 2.0*qp1;
  d2qlim = 0.0;
  lim_slope = fmin(fabs(d2ql), fabs(d2qr));
  if (d2qc > 0.0 && d2ql > 0.0 && d2qr > 0.0) d2qlim = sgn(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (d2qc < 0.0 && d2ql < 0.0 && d2qr < 0.0) d2qlim = sgn(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (((q - qrv)*(qp1 - qrv)) > 0.0) qrv = 0.5*(q + qp1) - d2qlim/6.0;
  double qa = (qrv - q)*(q - qlv);
  double qb = (qm1 - q)*(q - qp1);
  if (qa <= 0.0 || qb <= 0.0) {
    double d2q = 6.0*(qlv + qrv - 2.0*q);
    double e2qc = (qm1 + qp1) - 2.0*q;
    double e2ql = (qm2 + q) - 2.0*qm1;
    double e2qr = (q + qp2) - 2.0*qp1;
    d2qlim = 0.0;
    lim_slope = fmin(fabs(e2ql), fabs(e2qr));
    lim_slope = fmin(fabs(e2qc), lim_slope);
    if (e2qc > 0.0 && e2ql > 0.0 && e2qr > 0.0 && d2q > 0.0)
      d2qlim = sgn(d2q)*fmin(1.25*lim_slope, fabs(d2q));
    if (e2qc < 0.0 && e2ql < 0.0 && e2qr < 0.0 && d2q < 0.0)
      d2qlim = sgn(d2q)*fmin(1.25*lim_slope, fabs(d2q));
    double rho = 0.0;
    if (fabs(d2q) > (1.0e-12)*fmax(fabs(qm1), fmax(fabs(q), fabs(qp1)))) rho = d2qlim/d2q;
    qlv = q + (qlv - q)*rho;
    qrv = q + (qrv - q)*rho;
  } else {
    double qc = qrv - q;
    double qd = qlv - q;
    if (fabs(qc) >= 2.0*fabs(qd)) qrv = q - 2.0*qd;
    if (fabs(qd) >= 2.0*fabs(qc)) qlv = q - 2.0*qc;
  }
  ql_ip1 = qrv;
  qr_i = qlv;
}

// Jiang-Shu smoothness indicators (wenoz.hpp:32-43 == teno.hpp:33-44) and the two 5th-order
// face values for un-normalised weights (wenoz.hpp:58-81 == teno.hpp:64-86)
AKMI_DEV void js_beta(double qm2, double qm1, double q, double qp1, double qp2, double &b0,
                      double &b1, double &b2) {
  const double c0 = 13./12., c1 = 0.25;
  b0 = c0*sqr(qm2 + q - 2.0*qm1) + c1*sqr(qm2 + 3.0*q - 4.0*qm1);
  b1 = c0*sqr(qm1 + qp1 - 2.0*q) + c1*sqr(qm1 - qp1);
  b2 = c0*sqr(qp2 + q - 2.0*qp1) + c1*sqr(qp2 + 3.0*q - 4.0*qp1);
}
AKMI_DEV void weno_faces(double qm2, double qm1, double q, double qp1, double qp2, double wa,
                         double wb, double wc, double va, double vc, double &ql_ip1,
                         double &qr_i) {
  double f0 = (2.0*qm2 - 7.0*qm1 + 11.0*q);
  double f1 = (-1.0*qm1 + 5.0*q + 2.0*qp1);
  double f2 = (2.0*q + 5.0*qp1 - qp2);
  double asum = 6.0*(wa + wb + wc);
  ql_ip1 = (f0*wa + f1*wb + f2*wc)/asum;
  f0 = (2.0*qp2 - 7.0*qp1 + 11.0*q);
  f1 = (-1.0*qp1 + 5.0*q + 2.0*qm1);
  f2 = (2.0*q + 5.0*qm1 - qm2);
  asum = 6.0*(va + wb + vc);
  qr_i = (f0*va + f1*wb + f2*vc)/asum;
}

// WENO-Z, src/reconstruct/wenoz.hpp:29-84
AKMI_DEV void wenoz(double qm2, double qm1, double q, double qp1, double qp2, double &ql_ip1,
                    double &qr_i) {
  double b0, b1, b2;
  js_beta(qm2, qm1, q, qp1, qp2, b0, b1, b2);
  const double epsL = 1.0e-42;
  const double tau_5 = fabs(b0 - b2);
  double ind0 = sqr(tau_5/(b0 + epsL));
  double ind1 = sqr(tau_5/(b1 + epsL));
  double ind2 = sqr(tau_5/(b2 + epsL));
  weno_faces(qm2, qm1, q, qp1, qp2, 0.1*(1.0 + ind0), 0.6*(1.0 + ind1), 0.3*(1.0 + ind2),
             0.1*(1.0 + ind2), 0.3*(1.0 + ind0), ql_ip1, qr_i);
}

// TENO, src/reconstruct/teno.hpp:30-89
AKMI_DEV double cube(double x) { return x*x*x; }
AKMI_DEV void teno(double qm2, double qm1, double q, double qp1, double qp2, double &ql_ip1,
                   double &qr_i) {
  double b0, b1, b2;
  js_beta(qm2, qm1, q, qp1, qp2, b0, b1, b2);
  const double epsT = 1.0e-40, cT = 1.0e-6;
  double a0 = 1.0/sqr(cube(b0 + epsT));
  double a1 = 1.0/sqr(cube(b1 + epsT));
  double a2 = 1.0/sqr(cube(b2 + epsT));
  double asum = a0 + a1 + a2;
  double ind0 = (a0 < cT*asum ? 0.0 : 1.0);
  double ind1 = (a1 < cT*asum ? 0.0 : 1.0);
  double ind2 = (a2 < cT*asum ? 0.0 : 1.0);
  weno_faces(qm2, qm1, q, qp1, qp2, 0.1*ind0, 0.6*ind1, 0.3*ind2, 0.1*ind2, 0.3*ind0, ql_ip1,
             qr_i);
}

// what the flux kernels need of EOS_Data: gamma and the floors of the L/R states
// (recon.hpp:52-53: dfloor, efloor = pfloor/(gamma-1))
struct FaceEos { double gamma, dfloor, efloor, iso_cs; };

// five-point reconstructions behind one name.  RECON: 2 ppm4, 3 ppmx, 4 wenoz, 5 teno
template <int RECON>
AKMI_DEV void recon5(double qm2, double qm1, double q, double qp1, double qp2, double &ql_ip1,
                     double &qr_i) {
  if constexpr (RECON == 2) ppm4(qm2, qm1, q, qp1, qp2, ql_ip1, qr_i);
  else if constexpr (RECON == 3) ppmx(qm2, qm1, q, qp1, qp2, ql_ip1, qr_i);
  else if constexpr (RECON == 4) wenoz(qm2, qm1, q, qp1, qp2, ql_ip1, qr_i);
  else teno(qm2, qm1, q, qp1, qp2, ql_ip1, qr_i);
}

// floors of ReconCellT (recon.hpp:72-103): only in the ppmx/wenoz/teno branches, only for the
// fluid density (FL == 1) and internal energy (FL == 2); FL == 0: velocities, B
template <int RECON, int FL>
AKMI_DEV void floor_lr(const FaceEos &eos, double &a, double &b) {
  if constexpr (RECON >= 3 && FL == 1) { a = fmax(a, eos.dfloor); b = fmax(b, eos.dfloor); }
  if constexpr (RECON >= 3 && FL == 2) { a = fmax(a, eos.efloor); b = fmax(b, eos.efloor); }
}

// L/R states of the face between cells (c-1) and c along a direction, for one variable.
// q points at cell c; s is the element stride along the direction.
// RECON: 0 dc, 1 plm, 2 ppm4, 3 ppmx, 4 wenoz, 5 teno (ReconCellT,
// src/reconstruct/recon.hpp:40-118: cell c-1 writes wl(c), cell c writes wr(c)).
template <int RECON, int FL = 0>
AKMI_DEV void face_states(const double *__restrict__ q, long s, const FaceEos &eos, double &ql,
                          double &qr) {
  double dummy;
  if constexpr (RECON == 1) {
    double qm2 = q[-2*s], qm1 = q[-s], q0 = q[0], qp1 = q[s];
    plm(qm2, qm1, q0, ql, dummy);
    plm(qm1, q0, qp1, dummy, qr);
  } else if constexpr (RECON >= 2) {
    double qm3 = q[-3*s], qm2 = q[-2*s], qm1 = q[-s], q0 = q[0], qp1 = q[s], qp2 = q[2*s];
    recon5<RECON>(qm3, qm2, qm1, q0, qp1, ql, dummy);
    recon5<RECON>(qm2, qm1, q0, qp1, qp2, dummy, qr);
    floor_lr<RECON, FL>(eos, ql, qr);
  } else {
    ql = q[-s];
    qr = q[0];
  }
}

// Same, addressed as (wave-uniform base pointer) + (32-bit per-lane element offset): the
// stencil neighbours differ only in the uniform part, so the compiler keeps ONE offset VGPR per
// lane and forms the neighbour addresses on the scalar unit (global_load ... v_off, s[base]).
template <int RECON, int FL = 0>
AKMI_DEV void face_states_u(const double *__restrict__ base, unsigned ob, long s,
                            const FaceEos &eos, double &ql, double &qr) {
  // ob = BYTE offset of the lane (address = scalar base + zero-extended 32-bit VGPR)
  auto at = [&](const double *b) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(b) + ob); };
  double dummy;
  if constexpr (RECON == 1) {
    double qm2 = at(base - 2*s), qm1 = at(base - s), q0 = at(base), qp1 = at(base + s);
    plm(qm2, qm1, q0, ql, dummy);
    plm(qm1, q0, qp1, dummy, qr);
  } else if constexpr (RECON >= 2) {
    double qm3 = at(base - 3*s), qm2 = at(base - 2*s), qm1 = at(base - s), q0 = at(base),
           qp1 = at(base + s), qp2 = at(base + 2*s);
    recon5<RECON>(qm3, qm2, qm1, q0, qp1, ql, dummy);
    recon5<RECON>(qm2, qm1, q0, qp1, qp2, dummy, qr);
    floor_lr<RECON, FL>(eos, ql, qr);
  } else {
    ql = at(base - s);
    qr = at(base);
  }
}

// the same from a stencil that is already in registers: q[0..W-1] = cells f-LO .. f+W-LO-1 of face f
// (W, LO = 2, 1 donor cell; 4, 2 PLM; 6, 3 the five-point schemes).  Same calls, same order as face_states_u.
template <int RECON> constexpr int stencil_w() { return RECON == 0 ? 2 : (RECON == 1 ? 4 : 6); }
template <int RECON> constexpr int stencil_lo() { return RECON == 0 ? 1 : (RECON == 1 ? 2 : 3); }
template <int RECON, int FL = 0>
AKMI_DEV void face_states_v(const double *q, const FaceEos &eos, double &ql, double &qr) {
  double dummy;
  if constexpr (RECON == 1) {
    plm(q[0], q[1], q[2], ql, dummy);
    plm(q[1], q[2], q[3], dummy, qr);
  } else if constexpr (RECON >= 2) {
    recon5<RECON>(q[0], q[1], q[2], q[3], q[4], ql, dummy);
    recon5<RECON>(q[1], q[2], q[3], q[4], q[5], dummy, qr);
    floor_lr<RECON, FL>(eos, ql, qr);
  } else {
    ql = q[0];
    qr = q[1];
  }
}

// HLLC, src/hydro/rsolvers/hllc_hyd.hpp:20-115.  States are (d, vx, vy, vz, e_int) with
// vx along the sweep; flux is (d, mx, my, mz, E).
AKMI_DEV void hllc(double gamma, double wl_idn, double wl_ivx, double wl_ivy, double wl_ivz,
                   double wl_ien, double wr_idn, double wr_ivx, double wr_ivy, double wr_ivz,
                   double wr_ien, double &f_d, double &f_mx, double &f_my, double &f_mz,
                   double &f_e) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const double alpha = (gamma + 1.0)/(2.0*gamma);
  double wl_ipr = (gamma - 1.0)*wl_ien;
  double wr_ipr = (gamma - 1.0)*wr_ien;
  double qa, qb, qc, qd, qe, qf;
  qa = sqrt(gamma*wl_ipr/wl_idn);
  qb = sqrt(gamma*wr_ipr/wr_idn);
  double el = wl_ipr*igm1 + 0.5*wl_idn*(sqr(wl_ivx) + sqr(wl_ivy) + sqr(wl_ivz));
  double er = wr_ipr*igm1 + 0.5*wr_idn*(sqr(wr_ivx) + sqr(wr_ivy) + sqr(wr_ivz));
  qc = 0.25*(wl_idn + wr_idn)*(qa + qb);
  qd = 0.5*(wl_ipr + wr_ipr + (wl_ivx - wr_ivx)*qc);
  qe = (qd <= wl_ipr) ? 1.0 : sqrt(1.0 + alpha*((qd/wl_ipr) - 1.0));
  qf = (qd <= wr_ipr) ? 1.0 : sqrt(1.0 + alpha*((qd/wr_ipr) - 1.0));
  qc = wl_ivx - qa*qe;
  qd = wr_ivx + qb*qf;
  qa = qd > 0.0 ? qd : 1.0e-20;
  qb = qc < 0.0 ? qc : -1.0e-20;
  qe = wl_ivx - qc;
  qf = wr_ivx - qd;
  qc = wl_ipr + qe*wl_idn*wl_ivx;
  qd = wr_ipr + qf*wr_idn*wr_ivx;
  double ml = wl_idn*qe;
  double mr = -(wr_idn*qf);
  double am = (qc - qd)/(ml + mr);
  double cp = (ml*qd + mr*qc)/(ml + mr);
  cp = cp > 0.0 ? cp : 0.0;
  qe = wl_idn*(wl_ivx - qb);
  qf = wr_idn*(wr_ivx - qa);
  double fl_d = qe, fr_d = qf;
  double fl_mx = qe*wl_ivx + wl_ipr, fr_mx = qf*wr_ivx + wr_ipr;
  double fl_my = qe*wl_ivy, fr_my = qf*wr_ivy;
  double fl_mz = qe*wl_ivz, fr_mz = qf*wr_ivz;
  double fl_e = el*(wl_ivx - qb) + wl_ipr*wl_ivx;
  double fr_e = er*(wr_ivx - qa) + wr_ipr*wr_ivx;
  if (am >= 0.0) {
    qc = am/(am - qb);
    qd = 0.0;
    qe = -qb/(am - qb);
  } else {
    qc = 0.0;
    qd = -am/(qa - am);
    qe = qa/(qa - am);
  }
  f_d = qc*fl_d + qd*fr_d;
  f_mx = qc*fl_mx + qd*fr_mx + qe*cp;
  f_my = qc*fl_my + qd*fr_my;
  f_mz = qc*fl_mz + qd*fr_mz;
  f_e = qc*fl_e + qd*fr_e + qe*cp*am;
}

// LLF, src/hydro/rsolvers/llf_hyd_singlestate.hpp:28-78 (ideal gas)
AKMI_DEV void llf_hyd(double gamma, double ld, double lx, double ly, double lz, double le,
                      double rd, double rx, double ry, double rz, double re, double &f_d,
                      double &f_mx, double &f_my, double &f_mz, double &f_e) {
  double qa = ld*lx;
  double qb = rd*rx;
  double s_d = qa + qb;
  double s_mx = qa*lx + qb*rx;
  double s_my = qa*ly + qb*ry;
  double s_mz = qa*lz + qb*rz;
  double pl = (gamma - 1.0)*le;
  double pr = (gamma - 1.0)*re;
  double el = le + 0.5*ld*(sqr(lx) + sqr(ly) + sqr(lz));
  double er = re + 0.5*rd*(sqr(rx) + sqr(ry) + sqr(rz));
  s_mx += (pl + pr);
  double s_e = (el + pl)*lx + (er + pr)*rx;
  qa = sqrt(gamma*pl/ld);
  qb = sqrt(gamma*pr/rd);
  double a = fmax((fabs(lx) + qa), (fabs(rx) + qb));
  double du_d = a*(rd - ld);
  double du_mx = a*(rd*rx - ld*lx);
  double du_my = a*(rd*ry - ld*ly);
  double du_mz = a*(rd*rz - ld*lz);
  double du_e = a*(er - el);
  f_d = 0.5*(s_d - du_d);
  f_mx = 0.5*(s_mx - du_mx);
  f_my = 0.5*(s_my - du_my);
  f_mz = 0.5*(s_mz - du_mz);
  f_e = 0.5*(s_e - du_e);
}

// HLLE, src/hydro/rsolvers/hlle_hyd.hpp:27-129 (ideal gas)
AKMI_DEV void hlle_hyd(double gamma, double dl, double ul, double vl, double zl, double eil,
                       double dr, double ur, double vr, double zr, double eir, double &f_d,
                       double &f_mx, double &f_my, double &f_mz, double &f_e) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  double pl = (gamma - 1.0)*eil, pr = (gamma - 1.0)*eir;
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_vy = (sqrtdl*vl + sqrtdr*vr)*isdlpdr;
  double roe_vz = (sqrtdl*zl + sqrtdr*zr)*isdlpdr;
  double el = pl*igm1 + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(zl));
  double er = pr*igm1 + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(zr));
  double hroe = ((el + pl)/sqrtdl + (er + pr)/sqrtdr)*isdlpdr;
  double qa = sqrt(gamma*pl/dl);
  double qb = sqrt(gamma*pr/dr);
  double a = hroe - 0.5*(sqr(roe_vx) + sqr(roe_vy) + sqr(roe_vz));
  a = (a < 0.0) ? 0.0 : sqrt(gm1*a);
  double al = fmin((roe_vx - a), (ul - qa));
  double ar = fmax((roe_vx + a), (ur + qb));
  double bp = (ar > 0.0) ? ar : 1.0e-20;
  double bm = (al < 0.0) ? al : -1.0e-20;
  qa = ul - bm;
  qb = ur - bp;
  double fl_d = dl*qa, fr_d = dr*qb;
  double fl_mx = dl*ul*qa, fr_mx = dr*ur*qb;
  double fl_my = dl*vl*qa, fr_my = dr*vr*qb;
  double fl_mz = dl*zl*qa, fr_mz = dr*zr*qb;
  fl_mx += pl;
  fr_mx += pr;
  double fl_e = el*qa + pl*ul;
  double fr_e = er*qb + pr*ur;
  qa = 0.0;
  if (bp != bm) qa = 0.5*(bp + bm)/(bp - bm);
  f_d = 0.5*(fl_d + fr_d) + qa*(fl_d - fr_d);
  f_mx = 0.5*(fl_mx + fr_mx) + qa*(fl_mx - fr_mx);
  f_my = 0.5*(fl_my + fr_my) + qa*(fl_my - fr_my);
  f_mz = 0.5*(fl_mz + fr_mz) + qa*(fl_mz - fr_mz);
  f_e = 0.5*(fl_e + fr_e) + qa*(fl_e - fr_e);
}

// Roe with LLF fallback, src/hydro/rsolvers/roe_hyd.hpp:40-268 (RoeFluxAdb :183-268)
AKMI_DEV void roe_hyd(double gamma, double ld, double lx, double ly, double lz, double lei,
                      double rd, double rx, double ry, double rz, double rei, double &f_d,
                      double &f_mx, double &f_my, double &f_mz, double &f_e) {
  const double gm1 = gamma - 1.0;
  double wli[5] = {ld, lx, ly, lz, (gamma - 1.0)*lei};
  double wri[5] = {rd, rx, ry, rz, (gamma - 1.0)*rei};
  double fl[5], fr[5], du[5], ev[5], f[5];
  double sqrtdl = sqrt(wli[0]);
  double sqrtdr = sqrt(wri[0]);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double v1 = (sqrtdl*wli[1] + sqrtdr*wri[1])*isdlpdr;
  double v2 = (sqrtdl*wli[2] + sqrtdr*wri[2])*isdlpdr;
  double v3 = (sqrtdl*wli[3] + sqrtdr*wri[3])*isdlpdr;
  double el = wli[4]/gm1 + 0.5*wli[0]*(sqr(wli[1]) + sqr(wli[2]) + sqr(wli[3]));
  double er = wri[4]/gm1 + 0.5*wri[0]*(sqr(wri[1]) + sqr(wri[2]) + sqr(wri[3]));
  double h = ((el + wli[4])/sqrtdl + (er + wri[4])/sqrtdr)*isdlpdr;
  double mxl = wli[0]*wli[1];
  double mxr = wri[0]*wri[1];
  fl[0] = mxl;            fr[0] = mxr;
  fl[1] = mxl*wli[1];     fr[1] = mxr*wri[1];
  fl[2] = mxl*wli[2];     fr[2] = mxr*wri[2];
  fl[3] = mxl*wli[3];     fr[3] = mxr*wri[3];
  fl[1] += wli[4];        fr[1] += wri[4];
  fl[4] = (el + wli[4])*wli[1];
  fr[4] = (er + wri[4])*wri[1];
  du[0] = wri[0] - wli[0];
  du[1] = wri[0]*wri[1] - wli[0]*wli[1];
  du[2] = wri[0]*wri[2] - wli[0]*wli[2];
  du[3] = wri[0]*wri[3] - wli[0]*wli[3];
  du[4] = er - el;
#pragma unroll
  for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  bool llf_flag = false;
  {
    double vsq = v1*v1 + v2*v2 + v3*v3;
    double q = h - 0.5*vsq;
    double cs_sq = (q < 0.0) ? (double)(FLT_MIN) : gm1*q;
    double cs = sqrt(cs_sq);
    ev[0] = v1 - cs; ev[1] = v1; ev[2] = v1; ev[3] = v1; ev[4] = v1 + cs;
    double a[5];
    double na = 0.5/cs_sq;
    a[0]  = du[0]*(0.5*gm1*vsq + v1*cs);
    a[0] -= du[1]*(gm1*v1 + cs);
    a[0] -= du[2]*gm1*v2;
    a[0] -= du[3]*gm1*v3;
    a[0] += du[4]*gm1;
    a[0] *= na;
    a[1]  = du[0]*(-v2);
    a[1] += du[2];
    a[2]  = du[0]*(-v3);
    a[2] += du[3];
    double qa = gm1/cs_sq;
    a[3]  = du[0]*(1.0 - na*gm1*vsq);
    a[3] += du[1]*qa*v1;
    a[3] += du[2]*qa*v2;
    a[3] += du[3]*qa*v3;
    a[3] -= du[4]*qa;
    a[4]  = du[0]*(0.5*gm1*vsq - v1*cs);
    a[4] -= du[1]*(gm1*v1 - cs);
    a[4] -= du[2]*gm1*v2;
    a[4] -= du[3]*gm1*v3;
    a[4] += du[4]*gm1;
    a[4] *= na;
    double co[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) co[n] = -0.5*fabs(ev[n])*a[n];
    double dens = wli[0] + a[0];
    if (dens < 0.0) llf_flag = true;
    dens += a[3];
    if (dens < 0.0) llf_flag = true;
    f[0] += co[0];
    f[0] += co[3];
    f[0] += co[4];
    f[1] += co[0]*(v1 - cs);
    f[1] += co[3]*v1;
    f[1] += co[4]*(v1 + cs);
    f[2] += co[0]*v2;
    f[2] += co[1];
    f[2] += co[3]*v2;
    f[2] += co[4]*v2;
    f[3] += co[0]*v3;
    f[3] += co[2];
    f[3] += co[3]*v3;
    f[3] += co[4]*v3;
    f[4] += co[0]*(h - v1*cs);
    f[4] += co[1]*v2;
    f[4] += co[2]*v3;
    f[4] += co[3]*0.5*vsq;
    f[4] += co[4]*(h + v1*cs);
  }
  if (ev[0] >= 0.0) {
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = fl[n];
  }
  if (ev[4] <= 0.0) {
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = fr[n];
  }
  if (llf_flag) {
    double cl = sqrt(gamma*wli[4]/wli[0]);
    double cr = sqrt(gamma*wri[4]/wri[0]);
    double a = 0.5*fmax((fabs(wli[1]) + cl), (fabs(wri[1]) + cr));
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]) - a*du[n];
  }
  f_d = f[0]; f_mx = f[1]; f_my = f[2]; f_mz = f[3]; f_e = f[4];
}

// Advect, src/hydro/rsolvers/advect_hyd.hpp:19-55: upwind flux by the sign of the left normal
// velocity (kinematic runs).  As in the reference the transverse components are velocity times
// velocity (no density factor) and the energy component is e_int*v.
AKMI_DEV void advect_hyd(double ld, double lx, double ly, double lz, double le, double rd, double rx,
                         double ry, double rz, double re, double &f_d, double &f_mx, double &f_my,
                         double &f_mz, double &f_e) {
  if (lx >= 0.0) {
    f_d = ld*lx; f_mx = ld*lx*lx; f_my = ly*lx; f_mz = lz*lx; f_e = le*lx;
  } else {
    f_d = rd*rx; f_mx = rd*rx*rx; f_my = ry*rx; f_mz = rz*rx; f_e = re*rx;
  }
}

// Hydro_RSolver selection at compile time: RS = AKMI_RS_LLF 0, HLLE 1, HLLC 2, ROE 4, ADVECT 5
template <int RS>
AKMI_DEV void riemann_hyd(double gamma, double ld, double lx, double ly, double lz, double le,
                          double rd, double rx, double ry, double rz, double re, double &f_d,
                          double &f_mx, double &f_my, double &f_mz, double &f_e) {
  if constexpr (RS == 0) llf_hyd(gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 1) hlle_hyd(gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 4) roe_hyd(gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 5) advect_hyd(ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
  else hllc(gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
}

// IdealMHDFastSpeed, src/eos/eos.hpp:49-57.  FM: the short square root (sqrt_x) -- chosen per call site,
// like the early-outs of hlld(): it pays in the issue-bound k_sweep12s (1137 -> 1099 us) and costs the
// memory-latency-bound marches registers and basic blocks (x3 march 984 -> 1020 us), profiles/r03_ab1.txt
template <bool FM = false>
AKMI_DEV double fast_speed(double gamma, double d, double p, double bx, double by, double bz) {
  double asq = gamma*p;
  double ct2 = by*by + bz*bz;
  double qsq = bx*bx + ct2 + asq;
  double tmp = bx*bx + ct2 - asq;
  return sqrt_x<FM>(0.5*(qsq + sqrt_x<FM>(tmp*tmp + 4.0*asq*ct2))/d);
}

struct Cons1D { double d, mx, my, mz, e, by, bz; };

// HLLD (ideal gas), src/mhd/rsolvers/hlld_mhd.hpp:41-347.  Returns the 7-component flux
// (d,mx,my,mz,E,by,bz); the caller forms ey=-F(by), ez=+F(bz) (:346-347).
// EO: skip intermediate states no lane of the wave needs (see below).  Thread-per-face kernels gain
// (x1 sweep 689 -> 637 us at 256^3); the marching kernels sit at their register limit and lose
// (x2 march 877 -> 1011 us with 28 B of scratch): profiles/r02_hlld_earlyout.txt.  Chosen per call site.
#ifndef AKMI_HLLD_EARLYOUT
#define AKMI_HLLD_EARLYOUT 1
#endif
template <bool EO = false, bool FM = false>
AKMI_DEV Cons1D hlld(double gamma, double wl_idn, double wl_ivx, double wl_ivy, double wl_ivz,
                     double wl_ien, double wl_iby, double wl_ibz, double wr_idn, double wr_ivx,
                     double wr_ivy, double wr_ivz, double wr_ien, double wr_iby, double wr_ibz,
                     double bxi) {
  constexpr double SMALL = 1.0e-4;  // HLLD_SMALL_NUMBER, hlld_mhd.hpp:18
  double gm1 = gamma - 1.0;
  double igm1 = 1.0/gm1;
  double wl_ipr = (gamma - 1.0)*wl_ien;
  double wr_ipr = (gamma - 1.0)*wr_ien;

  double bxsq = bxi*bxi;
  double pbl = 0.5*(bxsq + (sqr(wl_iby) + sqr(wl_ibz)));
  double pbr = 0.5*(bxsq + (sqr(wr_iby) + sqr(wr_ibz)));
  double kel = 0.5*wl_idn*(sqr(wl_ivx) + (sqr(wl_ivy) + sqr(wl_ivz)));
  double ker = 0.5*wr_idn*(sqr(wr_ivx) + (sqr(wr_ivy) + sqr(wr_ivz)));

  Cons1D ul, ur;
  ul.d = wl_idn; ul.mx = wl_ivx*ul.d; ul.my = wl_ivy*ul.d; ul.mz = wl_ivz*ul.d;
  ul.e = wl_ipr*igm1 + kel + pbl; ul.by = wl_iby; ul.bz = wl_ibz;
  ur.d = wr_idn; ur.mx = wr_ivx*ur.d; ur.my = wr_ivy*ur.d; ur.mz = wr_ivz*ur.d;
  ur.e = wr_ipr*igm1 + ker + pbr; ur.by = wr_iby; ur.bz = wr_ibz;

  double cfl = fast_speed<FM>(gamma, wl_idn, wl_ipr, bxi, wl_iby, wl_ibz);
  double cfr = fast_speed<FM>(gamma, wr_idn, wr_ipr, bxi, wr_iby, wr_ibz);
  double spd0 = fmin(wl_ivx - cfl, wr_ivx - cfr);
  double spd4 = fmax(wl_ivx + cfl, wr_ivx + cfr);

  double ptl = wl_ipr + pbl;
  double ptr = wr_ipr + pbr;

  Cons1D fl, fr, flxi;
  fl.d = ul.mx;
  fl.mx = ul.mx*wl_ivx + ptl - bxsq;
  fl.my = ul.my*wl_ivx - bxi*ul.by;
  fl.mz = ul.mz*wl_ivx - bxi*ul.bz;
  fl.e = wl_ivx*(ul.e + ptl - bxsq) - bxi*(wl_ivy*ul.by + wl_ivz*ul.bz);
  fl.by = ul.by*wl_ivx - bxi*wl_ivy;
  fl.bz = ul.bz*wl_ivx - bxi*wl_ivz;

  fr.d = ur.mx;
  fr.mx = ur.mx*wr_ivx + ptr - bxsq;
  fr.my = ur.my*wr_ivx - bxi*ur.by;
  fr.mz = ur.mz*wr_ivx - bxi*ur.bz;
  fr.e = wr_ivx*(ur.e + ptr - bxsq) - bxi*(wr_ivy*ur.by + wr_ivz*ur.bz);
  fr.by = ur.by*wr_ivx - bxi*wr_ivy;
  fr.bz = ur.bz*wr_ivx - bxi*wr_ivz;

  double sdl = spd0 - wl_ivx;
  double sdr = spd4 - wr_ivx;
  double spd2 = (sdr*ur.mx - sdl*ul.mx + (ptl - ptr))/(sdr*ur.d - sdl*ul.d);

  double sdml = spd0 - spd2;
  double sdmr = spd4 - spd2;
  double sdml_inv = rcp_x<FM && AKMI_HLLD_FAST_RCP>(sdml);
  double sdmr_inv = rcp_x<FM && AKMI_HLLD_FAST_RCP>(sdmr);

  Cons1D ulst, uldst, urdst, urst;
  ulst.d = ul.d*sdl*sdml_inv;
  urst.d = ur.d*sdr*sdmr_inv;
  double ulst_d_inv = rcp_x<FM && AKMI_HLLD_FAST_RCP>(ulst.d);
  double urst_d_inv = rcp_x<FM && AKMI_HLLD_FAST_RCP>(urst.d);
  double sqrtdl = sqrt_x<FM>(ulst.d);
  double sqrtdr = sqrt_x<FM>(urst.d);

  double spd1 = spd2 - fabs(bxi)/sqrtdl;
  double spd3 = spd2 + fabs(bxi)/sqrtdr;

  double ptstl = ptl + ul.d*sdl*(spd2 - wl_ivx);
  double ptstr = ptr + ur.d*sdr*(spd2 - wr_ivx);
  double ptst = 0.5*(ptstr + ptstl);

  // Which intermediate states the selected flux needs (the selection below is the reference's,
  // hlld_mhd.hpp:313-344): F_L and F_R none; F*_L only U*_L; F*_R only U*_R; the double-star fluxes
  // both star states and the double-star states.  A state no lane of the wave needs is not computed
  // (wave-uniform branch, no divergence): in super-Alfvenic smooth flow whole waves take F*_L or F*_R
  // and skip half of the solver.  A state that IS computed is computed exactly as before.
#if AKMI_HLLD_EARLYOUT
  bool need_l = true, need_r = true, need_ds = true;
  if constexpr (EO) {
  // the branch of the selection below, by the same chain of comparisons (NaNs fall through alike)
  const int sel = (spd0 >= 0.0) ? 0 : (spd4 <= 0.0) ? 1 : (spd1 >= 0.0) ? 2 : (spd2 >= 0.0) ? 3 : (spd3 > 0.0) ? 4 : 5;
  need_l = __any(sel >= 2 && sel <= 4);
  need_r = __any(sel >= 3);
  need_ds = __any(sel == 3 || sel == 4);
  }
#else
  constexpr bool need_l = true, need_r = true, need_ds = true;
#endif
  double vbstl = 0.0, vbstr = 0.0;
  if (need_l) {
    ulst.mx = ulst.d*spd2;
    if (fabs(ul.d*sdl*sdml - bxsq) < (SMALL)*ptst) {
      ulst.my = ulst.d*wl_ivy;
      ulst.mz = ulst.d*wl_ivz;
      ulst.by = ul.by;
      ulst.bz = ul.bz;
    } else {
      double tmp = bxi*(sdl - sdml)/(ul.d*sdl*sdml - bxsq);
      ulst.my = ulst.d*(wl_ivy - ul.by*tmp);
      ulst.mz = ulst.d*(wl_ivz - ul.bz*tmp);
      tmp = (ul.d*sqr(sdl) - bxsq)/(ul.d*sdl*sdml - bxsq);
      ulst.by = ul.by*tmp;
      ulst.bz = ul.bz*tmp;
    }
    vbstl = (ulst.mx*bxi + (ulst.my*ulst.by + ulst.mz*ulst.bz))*ulst_d_inv;
    ulst.e = (sdl*ul.e - ptl*wl_ivx + ptst*spd2 +
              bxi*(wl_ivx*bxi + (wl_ivy*ul.by + wl_ivz*ul.bz) - vbstl))*sdml_inv;

  }
  if (need_r) {
    urst.mx = urst.d*spd2;
    if (fabs(ur.d*sdr*sdmr - bxsq) < (SMALL)*ptst) {
      urst.my = urst.d*wr_ivy;
      urst.mz = urst.d*wr_ivz;
      urst.by = ur.by;
      urst.bz = ur.bz;
    } else {
      double tmp = bxi*(sdr - sdmr)/(ur.d*sdr*sdmr - bxsq);
      urst.my = urst.d*(wr_ivy - ur.by*tmp);
      urst.mz = urst.d*(wr_ivz - ur.bz*tmp);
      tmp = (ur.d*sqr(sdr) - bxsq)/(ur.d*sdr*sdmr - bxsq);
      urst.by = ur.by*tmp;
      urst.bz = ur.bz*tmp;
    }
    vbstr = (urst.mx*bxi + (urst.my*urst.by + urst.mz*urst.bz))*urst_d_inv;
    urst.e = (sdr*ur.e - ptr*wr_ivx + ptst*spd2 +
              bxi*(wr_ivx*bxi + (wr_ivy*ur.by + wr_ivz*ur.bz) - vbstr))*sdmr_inv;

  }
  if (need_ds) {
    if (0.5*bxsq < (SMALL)*ptst) {
      uldst = ulst;
      urdst = urst;
    } else {
      double invsumd = rcp_x<FM && AKMI_HLLD_FAST_RCP>(sqrtdl + sqrtdr);
      double bxsig = (bxi > 0.0 ? 1.0 : -1.0);
      uldst.d = ulst.d;
      urdst.d = urst.d;
      uldst.mx = ulst.mx;
      urdst.mx = urst.mx;
      double tmp = invsumd*(sqrtdl*(ulst.my*ulst_d_inv) + sqrtdr*(urst.my*urst_d_inv) +
                            bxsig*(urst.by - ulst.by));
      uldst.my = uldst.d*tmp;
      urdst.my = urdst.d*tmp;
      tmp = invsumd*(sqrtdl*(ulst.mz*ulst_d_inv) + sqrtdr*(urst.mz*urst_d_inv) +
                     bxsig*(urst.bz - ulst.bz));
      uldst.mz = uldst.d*tmp;
      urdst.mz = urdst.d*tmp;
      tmp = invsumd*(sqrtdl*urst.by + sqrtdr*ulst.by +
                     bxsig*sqrtdl*sqrtdr*((urst.my*urst_d_inv) - (ulst.my*ulst_d_inv)));
      uldst.by = urdst.by = tmp;
      tmp = invsumd*(sqrtdl*urst.bz + sqrtdr*ulst.bz +
                     bxsig*sqrtdl*sqrtdr*((urst.mz*urst_d_inv) - (ulst.mz*ulst_d_inv)));
      uldst.bz = urdst.bz = tmp;
      tmp = spd2*bxi + (uldst.my*uldst.by + uldst.mz*uldst.bz)/uldst.d;
      uldst.e = ulst.e - sqrtdl*bxsig*(vbstl - tmp);
      urdst.e = urst.e + sqrtdr*bxsig*(vbstr - tmp);
    }

    uldst.d = spd1*(uldst.d - ulst.d);
    uldst.mx = spd1*(uldst.mx - ulst.mx);
    uldst.my = spd1*(uldst.my - ulst.my);
    uldst.mz = spd1*(uldst.mz - ulst.mz);
    uldst.e = spd1*(uldst.e - ulst.e);
    uldst.by = spd1*(uldst.by - ulst.by);
    uldst.bz = spd1*(uldst.bz - ulst.bz);

  }
  if (need_l) {
    ulst.d = spd0*(ulst.d - ul.d);
    ulst.mx = spd0*(ulst.mx - ul.mx);
    ulst.my = spd0*(ulst.my - ul.my);
    ulst.mz = spd0*(ulst.mz - ul.mz);
    ulst.e = spd0*(ulst.e - ul.e);
    ulst.by = spd0*(ulst.by - ul.by);
    ulst.bz = spd0*(ulst.bz - ul.bz);

  }
  if (need_ds) {
    urdst.d = spd3*(urdst.d - urst.d);
    urdst.mx = spd3*(urdst.mx - urst.mx);
    urdst.my = spd3*(urdst.my - urst.my);
    urdst.mz = spd3*(urdst.mz - urst.mz);
    urdst.e = spd3*(urdst.e - urst.e);
    urdst.by = spd3*(urdst.by - urst.by);
    urdst.bz = spd3*(urdst.bz - urst.bz);

  }
  if (need_r) {
    urst.d = spd4*(urst.d - ur.d);
    urst.mx = spd4*(urst.mx - ur.mx);
    urst.my = spd4*(urst.my - ur.my);
    urst.mz = spd4*(urst.mz - ur.mz);
    urst.e = spd4*(urst.e - ur.e);
    urst.by = spd4*(urst.by - ur.by);
    urst.bz = spd4*(urst.bz - ur.bz);

  }
  if (spd0 >= 0.0) {
    flxi = fl;
  } else if (spd4 <= 0.0) {
    flxi = fr;
  } else if (spd1 >= 0.0) {
    flxi.d = fl.d + ulst.d;     flxi.mx = fl.mx + ulst.mx;
    flxi.my = fl.my + ulst.my;  flxi.mz = fl.mz + ulst.mz;
    flxi.e = fl.e + ulst.e;     flxi.by = fl.by + ulst.by;  flxi.bz = fl.bz + ulst.bz;
  } else if (spd2 >= 0.0) {
    flxi.d = fl.d + ulst.d + uldst.d;      flxi.mx = fl.mx + ulst.mx + uldst.mx;
    flxi.my = fl.my + ulst.my + uldst.my;  flxi.mz = fl.mz + ulst.mz + uldst.mz;
    flxi.e = fl.e + ulst.e + uldst.e;
    flxi.by = fl.by + ulst.by + uldst.by;  flxi.bz = fl.bz + ulst.bz + uldst.bz;
  } else if (spd3 > 0.0) {
    flxi.d = fr.d + urst.d + urdst.d;      flxi.mx = fr.mx + urst.mx + urdst.mx;
    flxi.my = fr.my + urst.my + urdst.my;  flxi.mz = fr.mz + urst.mz + urdst.mz;
    flxi.e = fr.e + urst.e + urdst.e;
    flxi.by = fr.by + urst.by + urdst.by;  flxi.bz = fr.bz + urst.bz + urdst.bz;
  } else {
    flxi.d = fr.d + urst.d;     flxi.mx = fr.mx + urst.mx;
    flxi.my = fr.my + urst.my;  flxi.mz = fr.mz + urst.mz;
    flxi.e = fr.e + urst.e;     flxi.by = fr.by + urst.by;  flxi.bz = fr.bz + urst.bz;
  }
  return flxi;
}

// LLF for MHD, src/mhd/rsolvers/llf_mhd_singlestate.hpp:28-89 (ideal gas).  by/bz of the
// result are F(by), F(bz) in the convention of hlld(): the caller stores ey=-by, ez=+bz.
AKMI_DEV Cons1D llf_mhd(double gamma, double ld, double lx, double ly, double lz, double le,
                        double lby, double lbz, double rd, double rx, double ry, double rz,
                        double re, double rby, double rbz, double bxi) {
  double qa = ld*lx;
  double qb = rd*rx;
  double qc = 0.5*(sqr(lby) + sqr(lbz) - sqr(bxi));
  double qd = 0.5*(sqr(rby) + sqr(rbz) - sqr(bxi));
  double s_d = qa + qb;
  double s_mx = qa*lx + qb*rx + qc + qd;
  double s_my = qa*ly + qb*ry - bxi*(lby + rby);
  double s_mz = qa*lz + qb*rz - bxi*(lbz + rbz);
  double s_by = lby*lx + rby*rx - bxi*(ly + ry);
  double s_bz = lbz*lx + rbz*rx - bxi*(lz + rz);
  double pl = (gamma - 1.0)*le;
  double pr = (gamma - 1.0)*re;
  double el = le + 0.5*ld*(sqr(lx) + sqr(ly) + sqr(lz)) + qc + sqr(bxi);
  double er = re + 0.5*rd*(sqr(rx) + sqr(ry) + sqr(rz)) + qd + sqr(bxi);
  s_mx += (pl + pr);
  double s_e = (el + pl + qc)*lx + (er + pr + qd)*rx;
  s_e -= bxi*(lby*ly + lbz*lz);
  s_e -= bxi*(rby*ry + rbz*rz);
  qa = fast_speed(gamma, ld, pl, bxi, lby, lbz);
  qb = fast_speed(gamma, rd, pr, bxi, rby, rbz);
  double a = fmax((fabs(lx) + qa), (fabs(rx) + qb));
  Cons1D f;
  f.d = 0.5*(s_d - a*(rd - ld));
  f.mx = 0.5*(s_mx - a*(rd*rx - ld*lx));
  f.my = 0.5*(s_my - a*(rd*ry - ld*ly));
  f.mz = 0.5*(s_mz - a*(rd*rz - ld*lz));
  f.e = 0.5*(s_e - a*(er - el));
  f.by = 0.5*(s_by - a*(rby - lby));
  f.bz = 0.5*(s_bz - a*(rbz - lbz));
  return f;
}

// Advect for MHD, src/mhd/rsolvers/advect_mhd.hpp:18-58 (kinematic runs).  .e is NOT a flux: the
// reference leaves the energy flux untouched, and so do the callers of this function.
AKMI_DEV Cons1D advect_mhd(double ld, double lx, double ly, double lz, double lby, double lbz,
                           double rd, double rx, double ry, double rz, double rby, double rbz,
                           double bxi) {
  Cons1D f;
  f.my = 0.0; f.mz = 0.0; f.e = 0.0;
  if (lx >= 0.0) {
    f.d = ld*lx; f.mx = ld*lx*lx;
    f.by = -(-lby*lx + bxi*ly);            // the caller stores ey = -f.by
    f.bz = lbz*lx - bxi*lz;
  } else {
    f.d = rd*rx; f.mx = rd*rx*rx;
    f.by = -(-rby*rx + bxi*ry);
    f.bz = rbz*rx - bxi*rz;
  }
  return f;
}

// HLLE for MHD, src/mhd/rsolvers/hlle_mhd.hpp:24-178 (ideal gas)
AKMI_DEV Cons1D hlle_mhd(double gamma, double dl, double ul, double vl, double zl, double eil,
                         double byl, double bzl, double dr, double ur, double vr, double zr,
                         double eir, double byr, double bzr, double bxi) {
  double gm1 = gamma - 1.0;
  double igm1 = 1.0/gm1;
  double pl = (gamma - 1.0)*eil, pr = (gamma - 1.0)*eir;
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_d = sqrtdl*sqrtdr;
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_vy = (sqrtdl*vl + sqrtdr*vr)*isdlpdr;
  double roe_vz = (sqrtdl*zl + sqrtdr*zr)*isdlpdr;
  double roe_by = (sqrtdr*byl + sqrtdl*byr)*isdlpdr;
  double roe_bz = (sqrtdr*bzl + sqrtdl*bzr)*isdlpdr;
  double x = 0.5*(sqr(byl - byr) + sqr(bzl - bzr))/(sqr(sqrtdl + sqrtdr));
  double y = 0.5*(dl + dr)/roe_d;
  double pbl = 0.5*(bxi*bxi + sqr(byl) + sqr(bzl));
  double pbr = 0.5*(bxi*bxi + sqr(byr) + sqr(bzr));
  double el = pl*igm1 + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(zl)) + pbl;
  double er = pr*igm1 + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(zr)) + pbr;
  double hroe = ((el + pl + pbl)/sqrtdl + (er + pr + pbr)/sqrtdr)*isdlpdr;
  double cl = fast_speed(gamma, dl, pl, bxi, byl, bzl);
  double cr = fast_speed(gamma, dr, pr, bxi, byr, bzr);
  double btsq = sqr(roe_by) + sqr(roe_bz);
  double vaxsq = bxi*bxi/roe_d;
  double bt_starsq = (gm1 - (gm1 - 1.0)*y)*btsq;
  double hp = hroe - (vaxsq + btsq/roe_d);
  double vsq = sqr(roe_vx) + sqr(roe_vy) + sqr(roe_vz);
  double twid_asq = fmax((gm1*(hp - 0.5*vsq) - (gm1 - 1.0)*x), 0.0);
  double ct2 = bt_starsq/roe_d;
  double tsum = vaxsq + ct2 + twid_asq;
  double tdif = vaxsq + ct2 - twid_asq;
  double cf2_cs2 = sqrt(tdif*tdif + 4.0*twid_asq*ct2);
  double cfsq = 0.5*(tsum + cf2_cs2);
  double a = sqrt(cfsq);
  double al = fmin((roe_vx - a), (ul - cl));
  double ar = fmax((roe_vx + a), (ur + cr));
  double bp = ar > 0.0 ? ar : 1.0e-20;
  double bm = al < 0.0 ? al : -1.0e-20;
  double vxl = ul - bm;
  double vxr = ur - bp;
  double fl_d = dl*vxl, fr_d = dr*vxr;
  double fl_mx = dl*ul*vxl + pbl - sqr(bxi);
  double fr_mx = dr*ur*vxr + pbr - sqr(bxi);
  double fl_my = dl*vl*vxl - bxi*byl;
  double fr_my = dr*vr*vxr - bxi*byr;
  double fl_mz = dl*zl*vxl - bxi*bzl;
  double fr_mz = dr*zr*vxr - bxi*bzr;
  fl_mx += pl;
  fr_mx += pr;
  double fl_e = el*vxl + ul*(pl + pbl - bxi*bxi);
  double fr_e = er*vxr + ur*(pr + pbr - bxi*bxi);
  fl_e -= bxi*(byl*vl + bzl*zl);
  fr_e -= bxi*(byr*vr + bzr*zr);
  double fl_by = byl*vxl - bxi*vl;
  double fr_by = byr*vxr - bxi*vr;
  double fl_bz = bzl*vxl - bxi*zl;
  double fr_bz = bzr*vxr - bxi*zr;
  double tmp = 0.0;
  if (bp != bm) tmp = 0.5*(bp + bm)/(bp - bm);
  Cons1D f;
  f.d = 0.5*(fl_d + fr_d) + (fl_d - fr_d)*tmp;
  f.mx = 0.5*(fl_mx + fr_mx) + (fl_mx - fr_mx)*tmp;
  f.my = 0.5*(fl_my + fr_my) + (fl_my - fr_my)*tmp;
  f.mz = 0.5*(fl_mz + fr_mz) + (fl_mz - fr_mz)*tmp;
  f.e = 0.5*(fl_e + fr_e) + (fl_e - fr_e)*tmp;
  f.by = 0.5*(fl_by + fr_by) + (fl_by - fr_by)*tmp;
  f.bz = 0.5*(fl_bz + fr_bz) + (fl_bz - fr_bz)*tmp;
  return f;
}

// MHD_RSolver selection at compile time: RS = AKMI_RS_LLF 0, HLLE 1, HLLD 3
template <int RS, bool EO = false, bool FM = false>
AKMI_DEV Cons1D riemann_mhd(double gamma, double ld, double lx, double ly, double lz, double le,
                            double lby, double lbz, double rd, double rx, double ry, double rz,
                            double re, double rby, double rbz, double bxi) {
  if constexpr (RS == 5) return advect_mhd(ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  else if constexpr (RS == 0) return llf_mhd(gamma, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
  else if constexpr (RS == 1) return hlle_mhd(gamma, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
  else return hlld<EO, FM>(gamma, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
}

// ---- isothermal EOS (EOS_Data::is_ideal == false): the same source lines as the ideal-gas
// functions above, with the branches the reference takes when there is no energy equation.
// States (d, vx, vy, vz[, by, bz]); the energy slots of the common signatures are ignored.
AKMI_DEV void llf_hyd_iso(double cs, double ld, double lx, double ly, double lz, double rd,
                          double rx, double ry, double rz, double &f_d, double &f_mx, double &f_my,
                          double &f_mz) {
  double qa = ld*lx;
  double qb = rd*rx;
  double s_d = qa + qb;
  double s_mx = qa*lx + qb*rx;
  double s_my = qa*ly + qb*ry;
  double s_mz = qa*lz + qb*rz;
  s_mx += sqr(cs)*(ld + rd);
  double a = fmax((fabs(lx) + cs), (fabs(rx) + cs));
  f_d = 0.5*(s_d - a*(rd - ld));
  f_mx = 0.5*(s_mx - a*(rd*rx - ld*lx));
  f_my = 0.5*(s_my - a*(rd*ry - ld*ly));
  f_mz = 0.5*(s_mz - a*(rd*rz - ld*lz));
}

AKMI_DEV void hlle_hyd_iso(double iso_cs, double dl, double ul, double vl, double zl, double dr,
                           double ur, double vr, double zr, double &f_d, double &f_mx,
                           double &f_my, double &f_mz) {
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double al = fmin((roe_vx - iso_cs), (ul - iso_cs));
  double ar = fmax((roe_vx + iso_cs), (ur + iso_cs));
  double bp = (ar > 0.0) ? ar : 1.0e-20;
  double bm = (al < 0.0) ? al : -1.0e-20;
  double qa = ul - bm;
  double qb = ur - bp;
  double fl_d = dl*qa, fr_d = dr*qb;
  double fl_mx = dl*ul*qa, fr_mx = dr*ur*qb;
  double fl_my = dl*vl*qa, fr_my = dr*vr*qb;
  double fl_mz = dl*zl*qa, fr_mz = dr*zr*qb;
  fl_mx += (iso_cs*iso_cs)*dl;
  fr_mx += (iso_cs*iso_cs)*dr;
  qa = 0.0;
  if (bp != bm) qa = 0.5*(bp + bm)/(bp - bm);
  f_d = 0.5*(fl_d + fr_d) + qa*(fl_d - fr_d);
  f_mx = 0.5*(fl_mx + fr_mx) + qa*(fl_mx - fr_mx);
  f_my = 0.5*(fl_my + fr_my) + qa*(fl_my - fr_my);
  f_mz = 0.5*(fl_mz + fr_mz) + qa*(fl_mz - fr_mz);
}

// roe_hyd.hpp:40-181 with RoeFluxIso (:275-346)
AKMI_DEV void roe_hyd_iso(double iso_cs, double ld, double lx, double ly, double lz, double rd,
                          double rx, double ry, double rz, double &f_d, double &f_mx, double &f_my,
                          double &f_mz) {
  double wl[4] = {ld, lx, ly, lz}, wr[4] = {rd, rx, ry, rz};
  double fl[4], fr[4], du[4], ev[4], f[4];
  double sqrtdl = sqrt(wl[0]);
  double sqrtdr = sqrt(wr[0]);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double v1 = (sqrtdl*wl[1] + sqrtdr*wr[1])*isdlpdr;
  double v2 = (sqrtdl*wl[2] + sqrtdr*wr[2])*isdlpdr;
  double v3 = (sqrtdl*wl[3] + sqrtdr*wr[3])*isdlpdr;
  double mxl = wl[0]*wl[1];
  double mxr = wr[0]*wr[1];
  fl[0] = mxl;           fr[0] = mxr;
  fl[1] = mxl*wl[1];     fr[1] = mxr*wr[1];
  fl[2] = mxl*wl[2];     fr[2] = mxr*wr[2];
  fl[3] = mxl*wl[3];     fr[3] = mxr*wr[3];
  fl[1] += (iso_cs*iso_cs)*wl[0];
  fr[1] += (iso_cs*iso_cs)*wr[0];
  du[0] = wr[0] - wl[0];
  du[1] = wr[0]*wr[1] - wl[0]*wl[1];
  du[2] = wr[0]*wr[2] - wl[0]*wl[2];
  du[3] = wr[0]*wr[3] - wl[0]*wl[3];
#pragma unroll
  for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  bool llf_flag = false;
  {
    ev[0] = v1 - iso_cs; ev[1] = v1; ev[2] = v1; ev[3] = v1 + iso_cs;
    double a[4];
    a[0]  = du[0]*(0.5 + 0.5*v1/iso_cs);
    a[0] -= du[1]*0.5/iso_cs;
    a[1]  = du[0]*(-v2);
    a[1] += du[2];
    a[2]  = du[0]*(-v3);
    a[2] += du[3];
    a[3]  = du[0]*(0.5 - 0.5*v1/iso_cs);
    a[3] += du[1]*0.5/iso_cs;
    double co[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) co[n] = -0.5*fabs(ev[n])*a[n];
    double dens = wl[0] + a[0];
    if (dens < 0.0) llf_flag = true;
    dens += a[3];
    if (dens < 0.0) llf_flag = true;
    f[0] += co[0];
    f[0] += co[3];
    f[1] += co[0]*(v1 - iso_cs);
    f[1] += co[3]*(v1 + iso_cs);
    f[2] += co[0]*v2;
    f[2] += co[1];
    f[2] += co[3]*v2;
    f[3] += co[0]*v3;
    f[3] += co[2];
    f[3] += co[3]*v3;
  }
  if (ev[0] >= 0.0) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = fl[n];
  }
  if (ev[3] <= 0.0) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = fr[n];
  }
  if (llf_flag) {
    double a = 0.5*fmax((fabs(wl[1]) + iso_cs), (fabs(wr[1]) + iso_cs));
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]) - a*du[n];
  }
  f_d = f[0]; f_mx = f[1]; f_my = f[2]; f_mz = f[3];
}

// RS as in riemann_hyd (hllc does not exist for the isothermal EOS)
template <int RS>
AKMI_DEV void riemann_hyd_iso(double cs, double ld, double lx, double ly, double lz, double rd,
                              double rx, double ry, double rz, double &f_d, double &f_mx,
                              double &f_my, double &f_mz) {
  if constexpr (RS == 0) llf_hyd_iso(cs, ld, lx, ly, lz, rd, rx, ry, rz, f_d, f_mx, f_my, f_mz);
  else if constexpr (RS == 1) hlle_hyd_iso(cs, ld, lx, ly, lz, rd, rx, ry, rz, f_d, f_mx, f_my, f_mz);
  else if constexpr (RS == 5) {
    double fe;
    advect_hyd(ld, lx, ly, lz, 0.0, rd, rx, ry, rz, 0.0, f_d, f_mx, f_my, f_mz, fe);
  } else roe_hyd_iso(cs, ld, lx, ly, lz, rd, rx, ry, rz, f_d, f_mx, f_my, f_mz);
}

// isothermal fast speed, src/eos/eos.hpp:60-68
AKMI_DEV double fast_speed_iso(double cs, double d, double bx, double by, double bz) {
  double asq = (cs*cs)*d;
  double ct2 = by*by + bz*bz;
  double qsq = bx*bx + ct2 + asq;
  double tmp = bx*bx + ct2 - asq;
  return sqrt(0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*ct2))/d);
}

AKMI_DEV Cons1D llf_mhd_iso(double cs, double ld, double lx, double ly, double lz, double lby,
                            double lbz, double rd, double rx, double ry, double rz, double rby,
                            double rbz, double bxi) {
  double qa = ld*lx;
  double qb = rd*rx;
  double qc = 0.5*(sqr(lby) + sqr(lbz) - sqr(bxi));
  double qd = 0.5*(sqr(rby) + sqr(rbz) - sqr(bxi));
  double s_d = qa + qb;
  double s_mx = qa*lx + qb*rx + qc + qd;
  double s_my = qa*ly + qb*ry - bxi*(lby + rby);
  double s_mz = qa*lz + qb*rz - bxi*(lbz + rbz);
  double s_by = lby*lx + rby*rx - bxi*(ly + ry);
  double s_bz = lbz*lx + rbz*rx - bxi*(lz + rz);
  s_mx += sqr(cs)*(ld + rd);
  qa = fast_speed_iso(cs, ld, bxi, lby, lbz);
  qb = fast_speed_iso(cs, rd, bxi, rby, rbz);
  double a = fmax((fabs(lx) + qa), (fabs(rx) + qb));
  Cons1D f;
  f.d = 0.5*(s_d - a*(rd - ld));
  f.mx = 0.5*(s_mx - a*(rd*rx - ld*lx));
  f.my = 0.5*(s_my - a*(rd*ry - ld*ly));
  f.mz = 0.5*(s_mz - a*(rd*rz - ld*lz));
  f.e = 0.0;
  f.by = 0.5*(s_by - a*(rby - lby));
  f.bz = 0.5*(s_bz - a*(rbz - lbz));
  return f;
}

AKMI_DEV Cons1D hlle_mhd_iso(double iso_cs, double dl, double ul, double vl, double zl,
                             double byl, double bzl, double dr, double ur, double vr, double zr,
                             double byr, double bzr, double bxi) {
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_d = sqrtdl*sqrtdr;
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_by = (sqrtdr*byl + sqrtdl*byr)*isdlpdr;
  double roe_bz = (sqrtdr*bzl + sqrtdl*bzr)*isdlpdr;
  double x = 0.5*(sqr(byl - byr) + sqr(bzl - bzr))/(sqr(sqrtdl + sqrtdr));
  double y = 0.5*(dl + dr)/roe_d;
  double pbl = 0.5*(bxi*bxi + sqr(byl) + sqr(bzl));
  double pbr = 0.5*(bxi*bxi + sqr(byr) + sqr(bzr));
  double cl = fast_speed_iso(iso_cs, dl, bxi, byl, bzl);
  double cr = fast_speed_iso(iso_cs, dr, bxi, byr, bzr);
  double btsq = sqr(roe_by) + sqr(roe_bz);
  double vaxsq = bxi*bxi/roe_d;
  double bt_starsq = btsq*y;
  double twid_asq = iso_cs*iso_cs + x;
  double ct2 = bt_starsq/roe_d;
  double tsum = vaxsq + ct2 + twid_asq;
  double tdif = vaxsq + ct2 - twid_asq;
  double cf2_cs2 = sqrt(tdif*tdif + 4.0*twid_asq*ct2);
  double cfsq = 0.5*(tsum + cf2_cs2);
  double a = sqrt(cfsq);
  double al = fmin((roe_vx - a), (ul - cl));
  double ar = fmax((roe_vx + a), (ur + cr));
  double bp = ar > 0.0 ? ar : 1.0e-20;
  double bm = al < 0.0 ? al : -1.0e-20;
  double vxl = ul - bm;
  double vxr = ur - bp;
  double fl_d = dl*vxl, fr_d = dr*vxr;
  double fl_mx = dl*ul*vxl + pbl - sqr(bxi);
  double fr_mx = dr*ur*vxr + pbr - sqr(bxi);
  double fl_my = dl*vl*vxl - bxi*byl;
  double fr_my = dr*vr*vxr - bxi*byr;
  double fl_mz = dl*zl*vxl - bxi*bzl;
  double fr_mz = dr*zr*vxr - bxi*bzr;
  fl_mx += (iso_cs*iso_cs)*dl;
  fr_mx += (iso_cs*iso_cs)*dr;
  double fl_by = byl*vxl - bxi*vl;
  double fr_by = byr*vxr - bxi*vr;
  double fl_bz = bzl*vxl - bxi*zl;
  double fr_bz = bzr*vxr - bxi*zr;
  double tmp = 0.0;
  if (bp != bm) tmp = 0.5*(bp + bm)/(bp - bm);
  Cons1D f;
  f.d = 0.5*(fl_d + fr_d) + (fl_d - fr_d)*tmp;
  f.mx = 0.5*(fl_mx + fr_mx) + (fl_mx - fr_mx)*tmp;
  f.my = 0.5*(fl_my + fr_my) + (fl_my - fr_my)*tmp;
  f.mz = 0.5*(fl_mz + fr_mz) + (fl_mz - fr_mz)*tmp;
  f.e = 0.0;
  f.by = 0.5*(fl_by + fr_by) + (fl_by - fr_by)*tmp;
  f.bz = 0.5*(fl_bz + fr_bz) + (fl_bz - fr_bz)*tmp;
  return f;
}

// isothermal HLLD (Mignone 2007), src/mhd/rsolvers/hlld_mhd.hpp:349-545
AKMI_DEV Cons1D hlld_iso(double iso_cs, double dfloor_, double wl_idn, double wl_ivx, double wl_ivy,
                         double wl_ivz, double wl_iby, double wl_ibz, double wr_idn, double wr_ivx,
                         double wr_ivy, double wr_ivz, double wr_iby, double wr_ibz, double bxi) {
  constexpr double SMALL = 1.0e-4;
  double ul_d = wl_idn, ul_mx = wl_ivx*ul_d, ul_my = wl_ivy*ul_d, ul_mz = wl_ivz*ul_d;
  double ul_by = wl_iby, ul_bz = wl_ibz;
  double ur_d = wr_idn, ur_mx = wr_ivx*ur_d, ur_my = wr_ivy*ur_d, ur_mz = wr_ivz*ur_d;
  double ur_by = wr_iby, ur_bz = wr_ibz;
  double cfl = fast_speed_iso(iso_cs, wl_idn, bxi, wl_iby, wl_ibz);
  double cfr = fast_speed_iso(iso_cs, wr_idn, bxi, wr_iby, wr_ibz);
  double spd0 = fmin(wl_ivx - cfl, wr_ivx - cfr);
  double spd4 = fmax(wl_ivx + cfl, wr_ivx + cfr);
  double bxsq = bxi*bxi;
  double ptl = sqr(iso_cs)*wl_idn + 0.5*(bxsq + sqr(wl_iby) + sqr(wl_ibz));
  double ptr = sqr(iso_cs)*wr_idn + 0.5*(bxsq + sqr(wr_iby) + sqr(wr_ibz));
  double fl_d = ul_mx;
  double fl_mx = ul_mx*wl_ivx + ptl - bxsq;
  double fl_my = ul_my*wl_ivx - bxi*ul_by;
  double fl_mz = ul_mz*wl_ivx - bxi*ul_bz;
  double fl_by = ul_by*wl_ivx - bxi*wl_ivy;
  double fl_bz = ul_bz*wl_ivx - bxi*wl_ivz;
  double fr_d = ur_mx;
  double fr_mx = ur_mx*wr_ivx + ptr - bxsq;
  double fr_my = ur_my*wr_ivx - bxi*ur_by;
  double fr_mz = ur_mz*wr_ivx - bxi*ur_bz;
  double fr_by = ur_by*wr_ivx - bxi*wr_ivy;
  double fr_bz = ur_bz*wr_ivx - bxi*wr_ivz;
  double idspd = 1.0/(spd4 - spd0);
  double dhll = (spd4*ur_d - spd0*ul_d - fr_d + fl_d)*idspd;
  dhll = fmax(dhll, dfloor_);
  double sqrtdhll = sqrt(dhll);
  double fdhll = (spd4*fl_d - spd0*fr_d + spd4*spd0*(ur_d - ul_d))*idspd;
  double fmxhll = (spd4*fl_mx - spd0*fr_mx + spd4*spd0*(ur_mx - ul_mx))*idspd;
  double ustar = fdhll/dhll;
  double mxhll = (spd4*ur_mx - spd0*ul_mx - fr_mx + fl_mx)*idspd;
  double spd1 = ustar - fabs(bxi)/sqrtdhll;
  double spd3 = ustar + fabs(bxi)/sqrtdhll;
  double ulst_my, ulst_mz, ulst_by, ulst_bz, urst_my, urst_mz, urst_by, urst_bz;
  double tmp = (spd0 - spd1)*(spd0 - spd3);
  if (fabs(spd0 - spd1) < (SMALL)*iso_cs) {
    ulst_my = ul_my; ulst_mz = ul_mz; ulst_by = ul_by; ulst_bz = ul_bz;
  } else {
    double mfact = bxi*(ustar - wl_ivx)/tmp;
    double bfact = (ul_d*sqr(spd0 - wl_ivx) - bxsq)/(dhll*tmp);
    ulst_my = dhll*wl_ivy - ul_by*mfact;
    ulst_mz = dhll*wl_ivz - ul_bz*mfact;
    ulst_by = ul_by*bfact;
    ulst_bz = ul_bz*bfact;
  }
  tmp = (spd4 - spd1)*(spd4 - spd3);
  if (fabs(spd4 - spd3) < (SMALL)*iso_cs) {
    urst_my = ur_my; urst_mz = ur_mz; urst_by = ur_by; urst_bz = ur_bz;
  } else {
    double mfact = bxi*(ustar - wr_ivx)/tmp;
    double bfact = (ur_d*sqr(spd4 - wr_ivx) - bxsq)/(dhll*tmp);
    urst_my = dhll*wr_ivy - ur_by*mfact;
    urst_mz = dhll*wr_ivz - ur_bz*mfact;
    urst_by = ur_by*bfact;
    urst_bz = ur_bz*bfact;
  }
  double x = sqrtdhll*(bxi > 0.0 ? 1.0 : -1.0);
  double ucst_d = dhll;
  double ucst_my = 0.5*(ulst_my + urst_my + (urst_by - ulst_by)*x);
  double ucst_mz = 0.5*(ulst_mz + urst_mz + (urst_bz - ulst_bz)*x);
  double ucst_by = 0.5*(ulst_by + urst_by + (urst_my - ulst_my)/x);
  double ucst_bz = 0.5*(ulst_bz + urst_bz + (urst_mz - ulst_mz)/x);
  Cons1D f;
  f.e = 0.0;
  if (spd0 >= 0.0) {
    f.d = fl_d; f.mx = fl_mx; f.my = fl_my; f.mz = fl_mz; f.by = fl_by; f.bz = fl_bz;
  } else if (spd4 <= 0.0) {
    f.d = fr_d; f.mx = fr_mx; f.my = fr_my; f.mz = fr_mz; f.by = fr_by; f.bz = fr_bz;
  } else if (spd1 >= 0.0) {
    f.d = fl_d + spd0*(dhll - ul_d);
    f.mx = fl_mx + spd0*(mxhll - ul_mx);
    f.my = fl_my + spd0*(ulst_my - ul_my);
    f.mz = fl_mz + spd0*(ulst_mz - ul_mz);
    f.by = fl_by + spd0*(ulst_by - ul_by);
    f.bz = fl_bz + spd0*(ulst_bz - ul_bz);
  } else if (spd3 <= 0.0) {
    f.d = fr_d + spd4*(dhll - ur_d);
    f.mx = fr_mx + spd4*(mxhll - ur_mx);
    f.my = fr_my + spd4*(urst_my - ur_my);
    f.mz = fr_mz + spd4*(urst_mz - ur_mz);
    f.by = fr_by + spd4*(urst_by - ur_by);
    f.bz = fr_bz + spd4*(urst_bz - ur_bz);
  } else {
    f.d = dhll*ustar;
    f.mx = fmxhll;
    f.my = ucst_my*ustar - bxi*ucst_by;
    f.mz = ucst_mz*ustar - bxi*ucst_bz;
    f.by = ucst_by*ustar - bxi*ucst_my/ucst_d;
    f.bz = ucst_bz*ustar - bxi*ucst_mz/ucst_d;
  }
  return f;
}

template <int RS>
AKMI_DEV Cons1D riemann_mhd_iso(const FaceEos &eos, double ld, double lx, double ly, double lz,
                                double lby, double lbz, double rd, double rx, double ry, double rz,
                                double rby, double rbz, double bxi) {
  if constexpr (RS == 5) return advect_mhd(ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  else if constexpr (RS == 0) return llf_mhd_iso(eos.iso_cs, ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  else if constexpr (RS == 1) return hlle_mhd_iso(eos.iso_cs, ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  else return hlld_iso(eos.iso_cs, eos.dfloor, ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
}

// The fused stage kernels carry the equation of state in their Riemann-solver template parameter:
// RS >= 10 is the isothermal solver RS - 10 (llf 10, hlle 11, hlld 13, roe 14).  Isothermal states
// have no energy variable: slot 4 of the kernels' variable arrays stays unused (rs_iso<RS>()).
template <int RS> constexpr bool rs_iso() { return RS >= 10; }
template <int RS, bool EO = false, bool FM = false>
AKMI_DEV Cons1D riemann_mhd_e(const FaceEos &eos, double ld, double lx, double ly, double lz, double le,
                              double lby, double lbz, double rd, double rx, double ry, double rz,
                              double re, double rby, double rbz, double bxi) {
  if constexpr (RS >= 10) return riemann_mhd_iso<RS - 10>(eos, ld, lx, ly, lz, lby, lbz, rd, rx, ry, rz, rby, rbz, bxi);
  else return riemann_mhd<RS, EO, FM>(eos.gamma, ld, lx, ly, lz, le, lby, lbz, rd, rx, ry, rz, re, rby, rbz, bxi);
}
template <int RS>
AKMI_DEV void riemann_hyd_e(const FaceEos &eos, double ld, double lx, double ly, double lz, double le,
                            double rd, double rx, double ry, double rz, double re, double &f_d,
                            double &f_mx, double &f_my, double &f_mz, double &f_e) {
  if constexpr (RS >= 10) {
    riemann_hyd_iso<RS - 10>(eos.iso_cs, ld, lx, ly, lz, rd, rx, ry, rz, f_d, f_mx, f_my, f_mz);
    f_e = 0.0;
  } else {
    riemann_hyd<RS>(eos.gamma, ld, lx, ly, lz, le, rd, rx, ry, rz, re, f_d, f_mx, f_my, f_mz, f_e);
  }
}

// EOS_Data by value (src/eos/eos.hpp:27-34)
struct Eos {
  double gamma, dfloor, pfloor, tfloor, sfloor, sigma_max, iso_cs;
  int is_ideal;
};

// Entropy-floor predicate of SingleC2P_Ideal* (src/eos/ideal_c2p_hyd.hpp:57-63):
//   spe_over_eps = gm1/pow(d,gm1); spe = spe_over_eps*e*di; if (spe <= sfloor) ...
// The reference pays a pow() per cell only to feed this comparison.  We evaluate a cheap
// bracket first and fall back to the exact expression only if the bracket cannot decide,
// so the outcome is identical while the common case costs a few flops.
AKMI_DEV bool entropy_floor_hit(double wd, double we, double di, double gm1, double sfloor,
                                double &spe_over_eps) {
  // cheap estimate with float transcendental: relative error << 1e-3
  float lg = __log2f((float)wd);
  double approx = gm1/(double)exp2f((float)gm1*lg);
  double spe_a = approx*we*di;
  if (spe_a > 2.0*sfloor && spe_a == spe_a && (double)lg == (double)lg && wd > 1.0e-30 &&
      wd < 1.0e30) {
    return false;
  }
  spe_over_eps = gm1/pow(wd, gm1);
  double spe = spe_over_eps*we*di;
  return (spe <= sfloor);
}

// SingleC2P_IdealHyd, src/eos/ideal_c2p_hyd.hpp:22-66
AKMI_DEV void c2p_hyd(const Eos &eos, double &ud, double umx, double umy, double umz,
                      double &ue, double &wd, double &wvx, double &wvy, double &wvz,
                      double &we, bool &dfl, bool &efl, bool &tfl) {
  const double efloor = eos.pfloor/(eos.gamma - 1.0);
  const double gm1 = eos.gamma - 1.0;
  if (ud < eos.dfloor) { ud = eos.dfloor; dfl = true; }
  wd = ud;
  double di = 1.0/ud;
  wvx = di*umx; wvy = di*umy; wvz = di*umz;
  double e_k = 0.5*di*(sqr(umx) + sqr(umy) + sqr(umz));
  we = (ue - e_k);
  if (we < efloor) { we = efloor; ue = efloor + e_k; efl = true; }
  if (gm1*we*di < eos.tfloor) { we = wd*eos.tfloor/gm1; ue = we + e_k; tfl = true; }
  double spe_over_eps;
  if (entropy_floor_hit(wd, we, di, gm1, eos.sfloor, spe_over_eps)) {
    we = wd*eos.sfloor/spe_over_eps;
    efl = true;
  }
}

// SingleC2P_IdealMHD, src/eos/ideal_c2p_mhd.hpp:20-67
AKMI_DEV void c2p_mhd(const Eos &eos, double &ud, double umx, double umy, double umz,
                      double &ue, double ubx, double uby, double ubz, double &wd, double &wvx,
                      double &wvy, double &wvz, double &we, bool &dfl, bool &efl, bool &tfl) {
  const double b2 = sqr(ubx) + sqr(uby) + sqr(ubz);
  const double dfloor_ = fmax(eos.dfloor, b2/eos.sigma_max);
  const double efloor = eos.pfloor/(eos.gamma - 1.0);
  const double gm1 = eos.gamma - 1.0;
  if (ud < dfloor_) { ud = dfloor_; dfl = true; }
  wd = ud;
  double di = 1.0/ud;
  wvx = di*umx; wvy = di*umy; wvz = di*umz;
  double e_k = 0.5*di*(sqr(umx) + sqr(umy) + sqr(umz));
  double e_m = 0.5*(sqr(ubx) + sqr(uby) + sqr(ubz));
  we = (ue - e_k - e_m);
  if (we < efloor) { we = efloor; ue = efloor + e_k + e_m; efl = true; }
  if (gm1*we*di < eos.tfloor) { we = wd*eos.tfloor/gm1; ue = we + e_k + e_m; tfl = true; }
  double spe_over_eps;
  if (entropy_floor_hit(wd, we, di, gm1, eos.sfloor, spe_over_eps)) {
    we = wd*eos.sfloor/spe_over_eps;
    efl = true;
  }
}

}  // namespace akmi
#endif  // AKMI_NUMERICS_HPP_
