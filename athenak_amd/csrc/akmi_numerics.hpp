// akmi_numerics.hpp -- per-cell / per-face device arithmetic of the MeshBlock update.
//
// Written for gfx950 (wave64, fp64 VALU).  Everything here lives in registers: a face's two states are
// reconstructed from the cell stencil and handed straight to the Riemann solver, so the reference's global L/R
// buffers (src/hydro/hydro.hpp:102-113, src/mhd/mhd.hpp:134-137) never exist.
//
// Bit parity.  The library is compiled with -ffp-contract=off and every VALUE below is produced by the same
// sequence of IEEE operations as in the reference function cited next to it (a sum keeps its association, a
// division stays a division), so results compare bit for bit with the CPU path.  What parity does NOT fix is the
// shape of the code: which values are alive at the same time, what is computed for both sides of a face and what
// only for the side the answer comes from, what is shared between the two faces of a cell.  That shape is chosen
// here for the register file (each function says how); the tables "value <- reference line" keep the audit
// trail.  tests/test_numerics_host.py compiles this header for the CPU and compares it with the CPU checker of the test suite.
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

// =======================================================================================
// Reconstruction.  One call works on ONE cell and returns the values that cell contributes to its two faces
// (ReconCellT, src/reconstruct/recon.hpp:40-118: cell c feeds the left state of face c+1 and the right state of
// face c).  The sweeps are built around that: a lane / a march step reconstructs its cell once and hands one of
// the two values to its neighbour, so every limiter is evaluated once per cell and direction.
//   up   = value at the cell's upper face  (reference: ql(i+1))
//   down = value at the cell's lower face  (reference: qr(i))
// Arguments are the stencil in memory order: (.., below, here, above, ..).
// =======================================================================================

// piecewise linear, van Leer's harmonic mean of the two one-sided differences (src/reconstruct/plm.hpp:20-37):
//   half = (dl*dr)/(dl + dr) where the differences agree in sign, 0 otherwise; faces = here +- half
AKMI_DEV void plm(double below, double here, double above, double &up, double &down) {
  const double dl = here - below, dr = above - here;
  const double same = dl*dr;                        // > 0: a monotone stretch
  double half = same/(dl + dr);
  if (same <= 0.0) half = 0.0;
  up = here + half;
  down = here - half;
}

// fourth-order face values shared by the PPM variants: (7(a + b) - (a' + b'))/12 with a, b the two cells at the
// face and a', b' the next ones out (ppm.hpp:47-48, :87-88)
AKMI_DEV double face4(double in_a, double in_b, double out_a, double out_b) {
  return (7.*(in_a + in_b) - (out_a + out_b))/12.0;
}
// clip x into the interval spanned by p and q (ppm.hpp:51-54)
AKMI_DEV double clip_between(double x, double p, double q) {
  x = fmax(x, fmin(p, q));
  return fmin(x, fmax(p, q));
}
// Colella-Woodward steepness limiter on the two deviations from the cell average (ppm.hpp:60-72, :166-171): where
// one deviation is more than twice the other the parabola would overshoot inside the cell; pull the large one in
AKMI_DEV void cw_pull_in(double here, double &lo, double &hi) {
  const double dhi = hi - here, dlo = lo - here;
  if (fabs(dhi) >= 2.0*fabs(dlo)) hi = here - 2.0*dlo;
  if (fabs(dlo) >= 2.0*fabs(dhi)) lo = here - 2.0*dhi;
}

// PPM4, src/reconstruct/ppm.hpp:44-77 (Colella-Woodward limiters).  Stencil b2 b1 here a1 a2.
AKMI_DEV void ppm4(double b2, double b1, double here, double a1, double a2, double &up, double &down) {
  double lo = clip_between(face4(here, b1, b2, a1), here, b1);
  double hi = clip_between(face4(here, a1, b1, a2), here, a1);
  if (((hi - here)*(lo - here)) >= 0.0) {           // extremum inside the cell: flat
    lo = here;
    hi = here;
  } else {
    cw_pull_in(here, lo, hi);
  }
  up = hi;
  down = lo;
}

AKMI_DEV double sgn(double x) { return (x < 0.0) ? -1.0 : 1.0; }   // SIGN, src/athena.hpp:52

// Colella-Sekora second-derivative limiter (ppm.hpp:97-100, :108-111, :132-140): when the three (or four)
// curvatures agree in sign, the smallest of 1.25 x the neighbours' and |centre|, with the centre's sign
AKMI_DEV double cs_limited(double centre, double left, double right, double cap) {
  double lim = 0.0;
  if (centre > 0.0 && left > 0.0 && right > 0.0) lim = sgn(centre)*fmin(1.25*cap, fabs(centre));
  if (centre < 0.0 && left < 0.0 && right < 0.0) lim = sgn(centre)*fmin(1.25*cap, fabs(centre));
  return lim;
}

// PPMX, src/reconstruct/ppm.hpp:84-181 (Colella & Sekora extremum-preserving limiters)
AKMI_DEV void ppmx(double b2, double b1, double here, double a1, double a2, double &up, double &down) {
  double lo = face4(here, b1, b2, a1);
  double hi = face4(here, a1, b1, a2);
  // curvatures of the three-cell parabolas centred on b1, here, a1
  const double curv_b = (b2 + here) - 2.0*b1;
  const double curv_0 = (b1 + a1) - 2.0*here;
  const double curv_a = (here + a2) - 2.0*a1;
  {  // lower face value outside the range of its two cells -> rebuild it from the limited curvature (:93-102)
    const double c = 3.0*((b1 + here) - 2.0*lo);
    const double lim = cs_limited(c, curv_b, curv_0, fmin(fabs(curv_b), fabs(curv_0)));
    if (((b1 - lo)*(here - lo)) > 0.0) lo = 0.5*(here + b1) - lim/6.0;
  }
  {  // upper face value (:104-113)
    const double c = 3.0*((here + a1) - 2.0*hi);
    const double lim = cs_limited(c, curv_0, curv_a, fmin(fabs(curv_0), fabs(curv_a)));
    if (((here - hi)*(a1 - hi)) > 0.0) hi = 0.5*(here + a1) - lim/6.0;
  }
  const double extremum_in = (hi - here)*(here - lo);
  const double extremum_nb = (b1 - here)*(here - a1);
  if (extremum_in <= 0.0 || extremum_nb <= 0.0) {    // local extremum: scale both deviations (:118-146)
    const double c = 6.0*(lo + hi - 2.0*here);
    double cap = fmin(fabs(curv_b), fabs(curv_a));
    cap = fmin(fabs(curv_0), cap);
    double lim = 0.0;
    if (curv_0 > 0.0 && curv_b > 0.0 && curv_a > 0.0 && c > 0.0) lim = sgn(c)*fmin(1.25*cap, fabs(c));
    if (curv_0 < 0.0 && curv_b < 0.0 && curv_a < 0.0 && c < 0.0) lim = sgn(c)*fmin(1.25*cap, fabs(c));
    double scale = 0.0;
    if (fabs(c) > (1.0e-12)*fmax(fabs(b1), fmax(fabs(here), fabs(a1)))) scale = lim/c;
    lo = here + (lo - here)*scale;
    hi = here + (hi - here)*scale;
  } else {
    cw_pull_in(here, lo, hi);
  }
  up = hi;
  down = lo;
}

// Jiang-Shu smoothness indicators of the three sub-stencils (wenoz.hpp:32-43 == teno.hpp:33-44)
AKMI_DEV void smoothness3(double b2, double b1, double here, double a1, double a2, double &s_lo,
                          double &s_mid, double &s_hi) {
  const double k13 = 13./12., k4 = 0.25;
  s_lo = k13*sqr(b2 + here - 2.0*b1) + k4*sqr(b2 + 3.0*here - 4.0*b1);
  s_mid = k13*sqr(b1 + a1 - 2.0*here) + k4*sqr(b1 - a1);
  s_hi = k13*sqr(a2 + here - 2.0*a1) + k4*sqr(a2 + 3.0*here - 4.0*a1);
}
// the two fifth-order face values from un-normalised weights (wenoz.hpp:58-81 == teno.hpp:64-86): weights
// (u0,u1,u2) for the upper face, (d0,u1,d2) for the lower one -- the middle weight is shared
AKMI_DEV void weighted_faces(double b2, double b1, double here, double a1, double a2, double u0, double u1,
                             double u2, double d0, double d2, double &up, double &down) {
  {
    const double p0 = (2.0*b2 - 7.0*b1 + 11.0*here);
    const double p1 = (-1.0*b1 + 5.0*here + 2.0*a1);
    const double p2 = (2.0*here + 5.0*a1 - a2);
    up = (p0*u0 + p1*u1 + p2*u2)/(6.0*(u0 + u1 + u2));
  }
  {
    const double p0 = (2.0*a2 - 7.0*a1 + 11.0*here);
    const double p1 = (-1.0*a1 + 5.0*here + 2.0*b1);
    const double p2 = (2.0*here + 5.0*b1 - b2);
    down = (p0*d0 + p1*u1 + p2*d2)/(6.0*(d0 + u1 + d2));
  }
}

// WENO-Z, src/reconstruct/wenoz.hpp:29-84
AKMI_DEV void wenoz(double b2, double b1, double here, double a1, double a2, double &up, double &down) {
  double s0, s1, s2;
  smoothness3(b2, b1, here, a1, a2, s0, s1, s2);
  const double guard = 1.0e-42;
  const double tau = fabs(s0 - s2);
  const double z0 = sqr(tau/(s0 + guard)), z1 = sqr(tau/(s1 + guard)), z2 = sqr(tau/(s2 + guard));
  weighted_faces(b2, b1, here, a1, a2, 0.1*(1.0 + z0), 0.6*(1.0 + z1), 0.3*(1.0 + z2), 0.1*(1.0 + z2),
                 0.3*(1.0 + z0), up, down);
}

// TENO, src/reconstruct/teno.hpp:30-89: a sub-stencil is either in (weight of the linear scheme) or out
AKMI_DEV double cube(double x) { return x*x*x; }
AKMI_DEV void teno(double b2, double b1, double here, double a1, double a2, double &up, double &down) {
  double s0, s1, s2;
  smoothness3(b2, b1, here, a1, a2, s0, s1, s2);
  const double guard = 1.0e-40, cut = 1.0e-6;
  const double g0 = 1.0/sqr(cube(s0 + guard)), g1 = 1.0/sqr(cube(s1 + guard)), g2 = 1.0/sqr(cube(s2 + guard));
  const double total = g0 + g1 + g2;
  const double in0 = (g0 < cut*total ? 0.0 : 1.0), in1 = (g1 < cut*total ? 0.0 : 1.0), in2 = (g2 < cut*total ? 0.0 : 1.0);
  weighted_faces(b2, b1, here, a1, a2, 0.1*in0, 0.6*in1, 0.3*in2, 0.1*in2, 0.3*in0, up, down);
}

// what the flux kernels need of EOS_Data: gamma and the floors of the L/R states
// (recon.hpp:52-53: dfloor, efloor = pfloor/(gamma-1))
struct FaceEos { double gamma, dfloor, efloor, iso_cs; };

// five-point reconstructions behind one name.  RECON: 2 ppm4, 3 ppmx, 4 wenoz, 5 teno
template <int RECON>
AKMI_DEV void recon5(double b2, double b1, double here, double a1, double a2, double &up, double &down) {
  if constexpr (RECON == 2) ppm4(b2, b1, here, a1, a2, up, down);
  else if constexpr (RECON == 3) ppmx(b2, b1, here, a1, a2, up, down);
  else if constexpr (RECON == 4) wenoz(b2, b1, here, a1, a2, up, down);
  else teno(b2, b1, here, a1, a2, up, down);
}

// floors of ReconCellT (recon.hpp:72-103): only in the ppmx/wenoz/teno branches, only for the
// fluid density (FL == 1) and internal energy (FL == 2); FL == 0: velocities, B
template <int RECON, int FL>
AKMI_DEV void floor_lr(const FaceEos &eos, double &a, double &b) {
  if constexpr (RECON >= 3 && FL == 1) { a = fmax(a, eos.dfloor); b = fmax(b, eos.dfloor); }
  if constexpr (RECON >= 3 && FL == 2) { a = fmax(a, eos.efloor); b = fmax(b, eos.efloor); }
}

// The two states of ONE face from a stencil of cells (thread-per-face kernels): the face lies between st[LO-1] and
// st[LO]; its left state is the `up` value of the cell below it, its right state the `down` value of the cell above.
// W cells: 2 (donor cell), 4 (PLM), 6 (five-point schemes).
template <int RECON> constexpr int stencil_w() { return RECON == 0 ? 2 : (RECON == 1 ? 4 : 6); }
template <int RECON> constexpr int stencil_lo() { return RECON == 0 ? 1 : (RECON == 1 ? 2 : 3); }
template <int RECON, int FL = 0>
AKMI_DEV void face_states_v(const double *st, const FaceEos &eos, double &left, double &right) {
  double unused;
  if constexpr (RECON == 1) {
    plm(st[0], st[1], st[2], left, unused);
    plm(st[1], st[2], st[3], unused, right);
  } else if constexpr (RECON >= 2) {
    recon5<RECON>(st[0], st[1], st[2], st[3], st[4], left, unused);
    recon5<RECON>(st[1], st[2], st[3], st[4], st[5], unused, right);
    floor_lr<RECON, FL>(eos, left, right);
  } else {
    left = st[0];
    right = st[1];
  }
}
// ... the stencil fetched from memory: q points at the cell above the face, s is the element stride along the sweep
template <int RECON, int FL = 0>
AKMI_DEV void face_states(const double *__restrict__ q, long s, const FaceEos &eos, double &left, double &right) {
  constexpr int W = stencil_w<RECON>(), LO = stencil_lo<RECON>();
  double st[W];
#pragma unroll
  for (int c = 0; c < W; ++c) st[c] = q[(c - LO)*s];
  face_states_v<RECON, FL>(st, eos, left, right);
}
// ... addressed as (wave-uniform base pointer) + (32-bit per-lane BYTE offset): the stencil neighbours differ only in
// the uniform part, so the lane keeps ONE offset register and the neighbour addresses are formed on the scalar unit
template <int RECON, int FL = 0>
AKMI_DEV void face_states_u(const double *__restrict__ base, unsigned ob, long s, const FaceEos &eos, double &left,
                            double &right) {
  constexpr int W = stencil_w<RECON>(), LO = stencil_lo<RECON>();
  double st[W];
#pragma unroll
  for (int c = 0; c < W; ++c)
    st[c] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base + (c - LO)*s) + ob);
  face_states_v<RECON, FL>(st, eos, left, right);
}

// =======================================================================================
// Hydrodynamic Riemann solvers.  A state is (d, u, v, w, e) = density, velocity along the sweep, the two
// transverse velocities, internal energy density; the flux comes back as (d, mx, my, mz, E) in the same frame.
// =======================================================================================

// what a side contributes to every ideal-gas solver: pressure, total energy, mass flux
struct GasSide { double p, E; };
AKMI_DEV GasSide gas_side(double gamma, double igm1, double d, double u, double v, double w, double e) {
  GasSide s;
  s.p = (gamma - 1.0)*e;
  s.E = s.p*igm1 + 0.5*d*(sqr(u) + sqr(v) + sqr(w));
  return s;
}

// HLLC, src/hydro/rsolvers/hllc_hyd.hpp:20-115 (Toro's two-rarefaction pressure estimate for the wave speeds,
// contact speed and contact pressure from the two HLL-like fluxes).
//   value                     reference line
//   cl, cr (sound speeds)     :46-47        pmid, shock factors gl, gr   :56-62
//   sl, sr (outer speeds)     :65-66        bp, bm (signed bounds)       :70-71
//   contact speed sc, pc      :82-87        flux weights wl, wr, wp      :100-108
// FM: the four square roots through sqrt_x (same bits, ten instructions instead of eighteen behind a wave-uniform
// range test); a flag per call site, as for hlld().
template <bool FM = false>
AKMI_DEV void hllc(double gamma, double dl, double ul, double vl, double wl, double el, double dr, double ur,
                   double vr, double wr, double er, double &f_d, double &f_mx, double &f_my, double &f_mz,
                   double &f_e) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const double shock_k = (gamma + 1.0)/(2.0*gamma);
  const GasSide L = gas_side(gamma, igm1, dl, ul, vl, wl, el), R = gas_side(gamma, igm1, dr, ur, vr, wr, er);
  const double cl = sqrt_x<FM>(gamma*L.p/dl), cr = sqrt_x<FM>(gamma*R.p/dr);
  // two-rarefaction middle pressure, then the shock corrections of the outer speeds
  const double zbar = 0.25*(dl + dr)*(cl + cr);
  const double pmid = 0.5*(L.p + R.p + (ul - ur)*zbar);
  const double gl = (pmid <= L.p) ? 1.0 : sqrt_x<FM>(1.0 + shock_k*((pmid/L.p) - 1.0));
  const double gr = (pmid <= R.p) ? 1.0 : sqrt_x<FM>(1.0 + shock_k*((pmid/R.p) - 1.0));
  const double sl = ul - cl*gl, sr = ur + cr*gr;
  const double bp = sr > 0.0 ? sr : 1.0e-20;        // signed bounds of the fan
  const double bm = sl < 0.0 ? sl : -1.0e-20;
  // contact: speed and pressure from the jump conditions across the two outer waves
  const double rel_l = ul - sl, rel_r = ur - sr;
  const double tl = L.p + rel_l*dl*ul, tr = R.p + rel_r*dr*ur;
  const double ml = dl*rel_l, mr = -(dr*rel_r);
  const double sc = (tl - tr)/(ml + mr);
  double pc = (ml*tr + mr*tl)/(ml + mr);
  pc = pc > 0.0 ? pc : 0.0;
  // fluxes along the lines x/t = bm and x/t = bp
  const double ql = dl*(ul - bm), qr = dr*(ur - bp);
  const double fl[5] = {ql, ql*ul + L.p, ql*vl, ql*wl, L.E*(ul - bm) + L.p*ul};
  const double fr[5] = {qr, qr*ur + R.p, qr*vr, qr*wr, R.E*(ur - bp) + R.p*ur};
  double wgt_l, wgt_r, wgt_p;                        // weights of fl, fr and of the contact pressure
  if (sc >= 0.0) {
    wgt_l = sc/(sc - bm);
    wgt_r = 0.0;
    wgt_p = -bm/(sc - bm);
  } else {
    wgt_l = 0.0;
    wgt_r = -sc/(bp - sc);
    wgt_p = bp/(bp - sc);
  }
  f_d = wgt_l*fl[0] + wgt_r*fr[0];
  f_mx = wgt_l*fl[1] + wgt_r*fr[1] + wgt_p*pc;
  f_my = wgt_l*fl[2] + wgt_r*fr[2];
  f_mz = wgt_l*fl[3] + wgt_r*fr[3];
  f_e = wgt_l*fl[4] + wgt_r*fr[4] + wgt_p*pc*sc;
}

// LLF, src/hydro/rsolvers/llf_hyd_singlestate.hpp:28-78 (ideal gas): half the sum of the two physical fluxes
// minus half the largest signal speed times the jump of the conserved state
AKMI_DEV void llf_hyd(double gamma, double dl, double ul, double vl, double wl, double el, double dr, double ur,
                      double vr, double wr, double er, double &f_d, double &f_mx, double &f_my, double &f_mz,
                      double &f_e) {
  const double ml = dl*ul, mr = dr*ur;
  const double pl = (gamma - 1.0)*el, pr = (gamma - 1.0)*er;
  const double El = el + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(wl));
  const double Er = er + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(wr));
  double sum[5] = {ml + mr, ml*ul + mr*ur, ml*vl + mr*vr, ml*wl + mr*wr, 0.0};
  sum[1] += (pl + pr);
  sum[4] = (El + pl)*ul + (Er + pr)*ur;
  const double cl = sqrt(gamma*pl/dl), cr = sqrt(gamma*pr/dr);
  const double smax = fmax((fabs(ul) + cl), (fabs(ur) + cr));
  const double jump[5] = {dr - dl, dr*ur - dl*ul, dr*vr - dl*vl, dr*wr - dl*wl, Er - El};
  f_d = 0.5*(sum[0] - smax*jump[0]);
  f_mx = 0.5*(sum[1] - smax*jump[1]);
  f_my = 0.5*(sum[2] - smax*jump[2]);
  f_mz = 0.5*(sum[3] - smax*jump[3]);
  f_e = 0.5*(sum[4] - smax*jump[4]);
}

// Einfeldt's blend of the two fluxes taken along the lines x/t = bm <= 0 <= bp (hlle_*.hpp, last block): with
// t = (bp + bm)/(2 (bp - bm)) the HLL flux is (fl + fr)/2 + (fl - fr) t
AKMI_DEV double hll_tilt(double bp, double bm) {
  double t = 0.0;
  if (bp != bm) t = 0.5*(bp + bm)/(bp - bm);
  return t;
}

// HLLE, src/hydro/rsolvers/hlle_hyd.hpp:27-129 (ideal gas): Roe-averaged velocity and enthalpy bound the fan
AKMI_DEV void hlle_hyd(double gamma, double dl, double ul, double vl, double wl, double el, double dr, double ur,
                       double vr, double wr, double er, double &f_d, double &f_mx, double &f_my, double &f_mz,
                       double &f_e) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const GasSide L = gas_side(gamma, igm1, dl, ul, vl, wl, el), R = gas_side(gamma, igm1, dr, ur, vr, wr, er);
  const double rl = sqrt(dl), rr = sqrt(dr);
  const double inorm = 1.0/(rl + rr);
  const double u_roe = (rl*ul + rr*ur)*inorm, v_roe = (rl*vl + rr*vr)*inorm, w_roe = (rl*wl + rr*wr)*inorm;
  const double h_roe = ((L.E + L.p)/rl + (R.E + R.p)/rr)*inorm;
  const double cl = sqrt(gamma*L.p/dl), cr = sqrt(gamma*R.p/dr);
  double c_roe = h_roe - 0.5*(sqr(u_roe) + sqr(v_roe) + sqr(w_roe));
  c_roe = (c_roe < 0.0) ? 0.0 : sqrt(gm1*c_roe);
  const double smin = fmin((u_roe - c_roe), (ul - cl)), smax = fmax((u_roe + c_roe), (ur + cr));
  const double bp = (smax > 0.0) ? smax : 1.0e-20;
  const double bm = (smin < 0.0) ? smin : -1.0e-20;
  const double rel_l = ul - bm, rel_r = ur - bp;
  double fl[5] = {dl*rel_l, dl*ul*rel_l, dl*vl*rel_l, dl*wl*rel_l, L.E*rel_l + L.p*ul};
  double fr[5] = {dr*rel_r, dr*ur*rel_r, dr*vr*rel_r, dr*wr*rel_r, R.E*rel_r + R.p*ur};
  fl[1] += L.p;
  fr[1] += R.p;
  const double t = hll_tilt(bp, bm);
  f_d = 0.5*(fl[0] + fr[0]) + t*(fl[0] - fr[0]);
  f_mx = 0.5*(fl[1] + fr[1]) + t*(fl[1] - fr[1]);
  f_my = 0.5*(fl[2] + fr[2]) + t*(fl[2] - fr[2]);
  f_mz = 0.5*(fl[3] + fr[3]) + t*(fl[3] - fr[3]);
  f_e = 0.5*(fl[4] + fr[4]) + t*(fl[4] - fr[4]);
}

// Roe's linearisation with an LLF fallback, src/hydro/rsolvers/roe_hyd.hpp:40-268 (RoeFluxAdb :183-268).
// flux = (fl + fr)/2 - sum_k |lambda_k| alpha_k r_k / 2 over the five waves; if an intermediate density of the
// linearised fan is negative the face falls back to LLF; supersonic faces take the upwind flux.
AKMI_DEV void roe_hyd(double gamma, double dl, double ul, double vl, double wl, double el, double dr, double ur,
                      double vr, double wr, double er, double &f_d, double &f_mx, double &f_my, double &f_mz,
                      double &f_e) {
  const double gm1 = gamma - 1.0;
  const double pl = (gamma - 1.0)*el, pr = (gamma - 1.0)*er;
  const double rl = sqrt(dl), rr = sqrt(dr);
  const double inorm = 1.0/(rl + rr);
  const double u = (rl*ul + rr*ur)*inorm, v = (rl*vl + rr*vr)*inorm, w = (rl*wl + rr*wr)*inorm;
  const double El = pl/gm1 + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(wl));
  const double Er = pr/gm1 + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(wr));
  const double h = ((El + pl)/rl + (Er + pr)/rr)*inorm;
  const double ml = dl*ul, mr = dr*ur;
  double fl[5] = {ml, ml*ul, ml*vl, ml*wl, (El + pl)*ul};
  double fr[5] = {mr, mr*ur, mr*vr, mr*wr, (Er + pr)*ur};
  fl[1] += pl;
  fr[1] += pr;
  const double jump[5] = {dr - dl, dr*ur - dl*ul, dr*vr - dl*vl, dr*wr - dl*wl, Er - El};
  double f[5];
#pragma unroll
  for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  // eigen-decomposition of the jump at the Roe state (RoeFluxAdb)
  const double vsq = u*u + v*v + w*w;
  const double q = h - 0.5*vsq;
  const double csq = (q < 0.0) ? (double)(FLT_MIN) : gm1*q;
  const double c = sqrt(csq);
  const double lam_lo = u - c, lam_hi = u + c;
  const double half_icsq = 0.5/csq, g_csq = gm1/csq;
  double a_lo = jump[0]*(0.5*gm1*vsq + u*c);        // acoustic wave u - c
  a_lo -= jump[1]*(gm1*u + c);
  a_lo -= jump[2]*gm1*v;
  a_lo -= jump[3]*gm1*w;
  a_lo += jump[4]*gm1;
  a_lo *= half_icsq;
  double a_v = jump[0]*(-v);                         // the two shear waves
  a_v += jump[2];
  double a_w = jump[0]*(-w);
  a_w += jump[3];
  double a_s = jump[0]*(1.0 - half_icsq*gm1*vsq);    // entropy wave
  a_s += jump[1]*g_csq*u;
  a_s += jump[2]*g_csq*v;
  a_s += jump[3]*g_csq*w;
  a_s -= jump[4]*g_csq;
  double a_hi = jump[0]*(0.5*gm1*vsq - u*c);         // acoustic wave u + c
  a_hi -= jump[1]*(gm1*u - c);
  a_hi -= jump[2]*gm1*v;
  a_hi -= jump[3]*gm1*w;
  a_hi += jump[4]*gm1;
  a_hi *= half_icsq;
  const double k_lo = -0.5*fabs(lam_lo)*a_lo, k_v = -0.5*fabs(u)*a_v, k_w = -0.5*fabs(u)*a_w,
               k_s = -0.5*fabs(u)*a_s, k_hi = -0.5*fabs(lam_hi)*a_hi;
  bool fallback = false;
  double dmid = dl + a_lo;
  if (dmid < 0.0) fallback = true;
  dmid += a_s;
  if (dmid < 0.0) fallback = true;
  f[0] += k_lo;
  f[0] += k_s;
  f[0] += k_hi;
  f[1] += k_lo*(u - c);
  f[1] += k_s*u;
  f[1] += k_hi*(u + c);
  f[2] += k_lo*v;
  f[2] += k_v;
  f[2] += k_s*v;
  f[2] += k_hi*v;
  f[3] += k_lo*w;
  f[3] += k_w;
  f[3] += k_s*w;
  f[3] += k_hi*w;
  f[4] += k_lo*(h - u*c);
  f[4] += k_v*v;
  f[4] += k_w*w;
  f[4] += k_s*0.5*vsq;
  f[4] += k_hi*(h + u*c);
  if (lam_lo >= 0.0) {
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = fl[n];
  }
  if (lam_hi <= 0.0) {
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = fr[n];
  }
  if (fallback) {
    const double cl = sqrt(gamma*pl/dl), cr = sqrt(gamma*pr/dr);
    const double half_smax = 0.5*fmax((fabs(ul) + cl), (fabs(ur) + cr));
#pragma unroll
    for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]) - half_smax*jump[n];
  }
  f_d = f[0]; f_mx = f[1]; f_my = f[2]; f_mz = f[3]; f_e = f[4];
}

// Advect, src/hydro/rsolvers/advect_hyd.hpp:19-55: upwind flux by the sign of the left normal velocity
// (kinematic runs).  As in the reference the transverse components are velocity times velocity (no density
// factor) and the energy component is e_int*u.
AKMI_DEV void advect_hyd(double dl, double ul, double vl, double wl, double el, double dr, double ur, double vr,
                         double wr, double er, double &f_d, double &f_mx, double &f_my, double &f_mz, double &f_e) {
  const bool from_left = ul >= 0.0;
  const double d = from_left ? dl : dr, u = from_left ? ul : ur, v = from_left ? vl : vr, w = from_left ? wl : wr,
               e = from_left ? el : er;
  f_d = d*u; f_mx = d*u*u; f_my = v*u; f_mz = w*u; f_e = e*u;
}

// Hydro_RSolver selection at compile time: RS = AKMI_RS_LLF 0, HLLE 1, HLLC 2, ROE 4, ADVECT 5
template <int RS, bool FM = false>
AKMI_DEV void riemann_hyd(double gamma, double dl, double ul, double vl, double wl, double el, double dr,
                          double ur, double vr, double wr, double er, double &f_d, double &f_mx, double &f_my,
                          double &f_mz, double &f_e) {
  if constexpr (RS == 0) llf_hyd(gamma, dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 1) hlle_hyd(gamma, dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 4) roe_hyd(gamma, dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
  else if constexpr (RS == 5) advect_hyd(dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
  else hllc<FM>(gamma, dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
}

// =======================================================================================
// MHD.  A state is (d, u, v, w, e, by, bz) with u and the face field bn along the sweep; fluxes are Cons1D in the
// same frame: .by / .bz are F(by), F(bz) -- the caller stores ey = -F(by), ez = +F(bz) (hlld_mhd.hpp:346-347).
// =======================================================================================
struct Cons1D { double d, mx, my, mz, e, by, bz; };

// fast magnetosonic speed from  a^2 = gamma p,  ct^2 = by^2 + bz^2,  bn  (IdealMHDFastSpeed, src/eos/eos.hpp:49-57):
// cf^2 = ( (bn^2 + ct^2 + a^2) + sqrt((bn^2 + ct^2 - a^2)^2 + 4 a^2 ct^2) ) / (2 d)
// FM: the short square root (sqrt_x) -- chosen per call site, like the early-outs of hlld(): it pays in the
// issue-bound k_sweep12s (1137 -> 1099 us) and costs the memory-latency-bound marches registers and basic blocks
// (x3 march 984 -> 1020 us), profiles/r03_ab1.txt
template <bool FM = false>
AKMI_DEV double fast_speed_of(double asq, double d, double bn, double by, double bz) {
  const double ct2 = by*by + bz*bz;
  const double total = bn*bn + ct2 + asq;
  const double diff = bn*bn + ct2 - asq;
  return sqrt_x<FM>(0.5*(total + sqrt_x<FM>(diff*diff + 4.0*asq*ct2))/d);
}
template <bool FM = false>
AKMI_DEV double fast_speed(double gamma, double d, double p, double bn, double by, double bz) {
  return fast_speed_of<FM>(gamma*p, d, bn, by, bz);
}

// HLLD (Miyoshi & Kusano 2005, ideal gas), src/mhd/rsolvers/hlld_mhd.hpp:41-347.
//
// The fan has five waves  sL <= sAL <= sM <= sAR <= sR  (outer fast waves, the two rotational waves, the contact) and
// the flux is, per face,
//     F_S                                  outside the fan                       (S = the upwind side)
//     F_S + sS (U*_S - U_S)                between an outer and a rotational wave
//     F_S + sS (U*_S - U_S) + sAS (U**_S - U*_S)    between a rotational wave and the contact
// i.e. every face needs the physical flux, the star state and (sometimes) the double-star state of ONE side only --
// the side the contact has moved away from.  The reference evaluates both sides, all four intermediate states and all
// differences and picks at the end; here the order is
//   1. what both sides contribute to the five speeds and the total star pressure        (scalars, both sides)
//   2. the selection                                                                    (per lane)
//   3. if a lane of the wave needs a double-star state: the transverse star components of BOTH sides and the four
//      common double-star values (they mix the two sides)
//   4. everything else -- F, U, U*, E*, U**, the differences and the sum -- for the lane's own side S, on operands
//      picked per lane with selects.  Left and right formulas are the same sequence of operations on mirrored
//      operands; the two places where a sign differs (sAL = sM - |bn|/sqrt(d*), sAR = sM + ..;  E**_L = E* - X,
//      E**_R = E* + X) are x + y == x - (-y), exact in IEEE arithmetic.
// Per value the operations are the reference's, so the result is bit-identical; what changes is that a lane carries
// one side's seven-vectors instead of two sides' (registers) and computes one flux, one star state and one set of
// differences instead of two, four and four (instructions).
//   value (S = l or r)             reference lines          value                          reference lines
//   E, pt, cf                      :66-88                    transverse star (my,mz,by,bz)   :171-190 (l) :203-222 (r)
//   sL, sR, sM                     :91-92, :128-131          vb*, E*                          :192-200, :224-232
//   d*, 1/d*, sqrt(d*), sAL, sAR   :139-151                  double star                      :235-277
//   pt* (both), mean               :154-156                  differences, selection, sum      :280-344
// EO: wave-uniform early-outs -- skip step 3 when no lane of the wave needs a double-star state, and steps 3-4 when
// every lane is outside the fan.  FM: short square roots.  Both are chosen per call site (registers, see callers).
#ifndef AKMI_HLLD_EARLYOUT
#define AKMI_HLLD_EARLYOUT 1
#endif
struct StarT { double my, mz, by, bz; };             // transverse components of a star state
// :171-190 / :203-222 -- d = density, a = s - u, c = s - sM, da = d*a, ds = star density of the side
AKMI_DEV StarT hlld_star_transverse(double d, double a, double c, double da, double ds, double v, double w,
                                    double by, double bz, double bn, double bn2, double small_pt) {
  StarT s;
  const double den = da*c - bn2;
  if (fabs(den) < small_pt) {                        // the rotational wave coincides with the outer one
    s.my = ds*v;
    s.mz = ds*w;
    s.by = by;
    s.bz = bz;
  } else {
    const double kv = bn*(a - c)/den;
    s.my = ds*(v - by*kv);
    s.mz = ds*(w - bz*kv);
    const double kb = (d*sqr(a) - bn2)/den;
    s.by = by*kb;
    s.bz = bz*kb;
  }
  return s;
}

template <bool EO = false, bool FM = false>
AKMI_DEV Cons1D hlld(double gamma, double dl, double ul, double vl, double wl, double el, double byl, double bzl,
                     double dr, double ur, double vr, double wr, double er, double byr, double bzr, double bn) {
  constexpr double SMALL = 1.0e-4;                   // HLLD_SMALL_NUMBER, hlld_mhd.hpp:18
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const double bn2 = bn*bn;
  // ---- 1. both sides: pressure, total energy, total pressure, fast speed -> the five speeds, the star densities
  const double pl = (gamma - 1.0)*el, pr = (gamma - 1.0)*er;
  const double pml = 0.5*(bn2 + (sqr(byl) + sqr(bzl))), pmr = 0.5*(bn2 + (sqr(byr) + sqr(bzr)));
  const double El = pl*igm1 + 0.5*dl*(sqr(ul) + (sqr(vl) + sqr(wl))) + pml;
  const double Er = pr*igm1 + 0.5*dr*(sqr(ur) + (sqr(vr) + sqr(wr))) + pmr;
  const double cfl = fast_speed<FM>(gamma, dl, pl, bn, byl, bzl), cfr = fast_speed<FM>(gamma, dr, pr, bn, byr, bzr);
  const double sL = fmin(ul - cfl, ur - cfr), sR = fmax(ul + cfl, ur + cfr);
  const double ptl = pl + pml, ptr = pr + pmr;
  const double al = sL - ul, ar = sR - ur;                                      // speed of the outer wave relative to the gas
  const double sM = (ar*(ur*dr) - al*(ul*dl) + (ptl - ptr))/(ar*dr - al*dl);    // contact
  const double cl = sL - sM, cr = sR - sM;
  const double icl = rcp_x<FM && AKMI_HLLD_FAST_RCP>(cl), icr = rcp_x<FM && AKMI_HLLD_FAST_RCP>(cr);
  const double dal = dl*al, dar = dr*ar;
  const double dsl = dal*icl, dsr = dar*icr;                                    // star densities
  const double idsl = rcp_x<FM && AKMI_HLLD_FAST_RCP>(dsl), idsr = rcp_x<FM && AKMI_HLLD_FAST_RCP>(dsr);
  const double rl = sqrt_x<FM>(dsl), rr = sqrt_x<FM>(dsr);
  const double sAL = sM - fabs(bn)/rl, sAR = sM + fabs(bn)/rr;                   // rotational waves
  const double ptsl = ptl + dal*(sM - ul), ptsr = ptr + dar*(sM - ur);
  const double pts = 0.5*(ptsr + ptsl);                                          // total pressure of the star region
  // ---- 2. which of the six fluxes (the chain of comparisons of :313-344; NaNs fall through alike)
  const int sel = (sL >= 0.0) ? 0 : (sR <= 0.0) ? 1 : (sAL >= 0.0) ? 2 : (sM >= 0.0) ? 3 : (sAR > 0.0) ? 4 : 5;
  const bool left = (sel == 0) || (sel == 2) || (sel == 3);      // the side the flux is built on
  const bool inside = sel >= 2;                                  // needs a star state
  const bool twice = (sel == 3) || (sel == 4);                   // needs a double-star state
  bool any_inside = true, any_twice = true;
#if AKMI_HLLD_EARLYOUT
  if constexpr (EO) { any_inside = __any(inside); any_twice = __any(twice); }
#endif
  // ---- operands of the lane's own side
  const double d = left ? dl : dr, u = left ? ul : ur, v = left ? vl : vr, w = left ? wl : wr;
  const double by = left ? byl : byr, bz = left ? bzl : bzr;
  const double E = left ? El : Er, pt = left ? ptl : ptr;
  const double mx = u*d, my = v*d, mz = w*d;
  // physical flux of that side (:94-110)
  Cons1D F;
  F.d = mx;
  F.mx = mx*u + pt - bn2;
  F.my = my*u - bn*by;
  F.mz = mz*u - bn*bz;
  F.e = u*(E + pt - bn2) - bn*(v*by + w*bz);
  F.by = by*u - bn*v;
  F.bz = bz*u - bn*w;
  if (!any_inside) return F;                                     // the whole wave is outside its fans
  const double small_pt = (SMALL)*pts;
  // ---- 3. double-star states mix the sides: transverse star components of both, then the four common values
  StarT T;                                                       // transverse star components of the lane's side
  double ds_my = 0.0, ds_mz = 0.0, ds_by = 0.0, ds_bz = 0.0, ds_vb = 0.0;      // common double-star v, w, by, bz, v.B
  bool degenerate = true;                                        // :235 -- no rotational discontinuity: U** = U*
  double bsign = 1.0;
  if (any_twice) {
    const StarT Tl = hlld_star_transverse(dl, al, cl, dal, dsl, vl, wl, byl, bzl, bn, bn2, small_pt);
    const StarT Tr = hlld_star_transverse(dr, ar, cr, dar, dsr, vr, wr, byr, bzr, bn, bn2, small_pt);
    degenerate = 0.5*bn2 < small_pt;
    if (!degenerate) {
      const double inorm = rcp_x<FM && AKMI_HLLD_FAST_RCP>(rl + rr);
      bsign = (bn > 0.0 ? 1.0 : -1.0);
      const double vsl = Tl.my*idsl, vsr = Tr.my*idsr, wsl = Tl.mz*idsl, wsr = Tr.mz*idsr;   // star velocities
      ds_my = inorm*(rl*vsl + rr*vsr + bsign*(Tr.by - Tl.by));
      ds_mz = inorm*(rl*wsl + rr*wsr + bsign*(Tr.bz - Tl.bz));
      ds_by = inorm*(rl*Tr.by + rr*Tl.by + bsign*rl*rr*(vsr - vsl));
      ds_bz = inorm*(rl*Tr.bz + rr*Tl.bz + bsign*rl*rr*(wsr - wsl));
      // v.B of the double-star region, from the LEFT double-star state (:272)
      ds_vb = sM*bn + ((dsl*ds_my)*ds_by + (dsl*ds_mz)*ds_bz)/dsl;
    }
    T.my = left ? Tl.my : Tr.my; T.mz = left ? Tl.mz : Tr.mz;
    T.by = left ? Tl.by : Tr.by; T.bz = left ? Tl.bz : Tr.bz;
  }
  // ---- 4. the lane's side
  const double a = left ? al : ar, c = left ? cl : cr, ic = left ? icl : icr;
  const double ds = left ? dsl : dsr, ids = left ? idsl : idsr;
  const double sO = left ? sL : sR, sA = left ? sAL : sAR;
  if (!any_twice) T = hlld_star_transverse(d, a, c, left ? dal : dar, ds, v, w, by, bz, bn, bn2, small_pt);
  Cons1D S;                                                      // star state (:169-200 / :201-232)
  S.d = ds;
  S.mx = ds*sM;
  S.my = T.my; S.mz = T.mz; S.by = T.by; S.bz = T.bz;
  const double vb = (S.mx*bn + (S.my*S.by + S.mz*S.bz))*ids;
  S.e = (a*E - pt*u + pts*sM + bn*(u*bn + (v*by + w*bz) - vb))*ic;
  // flux = F + sO (U* - U) [+ sA (U** - U*)]
  Cons1D out;
  out.d = F.d + sO*(S.d - d);
  out.mx = F.mx + sO*(S.mx - mx);
  out.my = F.my + sO*(S.my - my);
  out.mz = F.mz + sO*(S.mz - mz);
  out.e = F.e + sO*(S.e - E);
  out.by = F.by + sO*(S.by - by);
  out.bz = F.bz + sO*(S.bz - bz);
  if (any_twice) {
    Cons1D D = S;                                                // double-star state; degenerate: the star state
    if (!degenerate) {
      D.my = ds*ds_my;
      D.mz = ds*ds_mz;
      D.by = ds_by;
      D.bz = ds_bz;
      const double r = left ? rl : rr;
      const double x = r*bsign*(vb - ds_vb);                     // E**_l = E* - x,  E**_r = E* + x
      D.e = S.e - (left ? x : -x);
    }
    const double t_d = out.d + sA*(D.d - S.d), t_mx = out.mx + sA*(D.mx - S.mx), t_my = out.my + sA*(D.my - S.my),
                 t_mz = out.mz + sA*(D.mz - S.mz), t_e = out.e + sA*(D.e - S.e), t_by = out.by + sA*(D.by - S.by),
                 t_bz = out.bz + sA*(D.bz - S.bz);
    if (twice) { out.d = t_d; out.mx = t_mx; out.my = t_my; out.mz = t_mz; out.e = t_e; out.by = t_by; out.bz = t_bz; }
  }
  if (!inside) out = F;
  return out;
}

// LLF for MHD, src/mhd/rsolvers/llf_mhd_singlestate.hpp:28-89 (ideal gas)
AKMI_DEV Cons1D llf_mhd(double gamma, double dl, double ul, double vl, double wl, double el, double byl,
                        double bzl, double dr, double ur, double vr, double wr, double er, double byr, double bzr,
                        double bn) {
  const double ml = dl*ul, mr = dr*ur;
  const double pml = 0.5*(sqr(byl) + sqr(bzl) - sqr(bn)), pmr = 0.5*(sqr(byr) + sqr(bzr) - sqr(bn));   // pm - bn^2
  const double pl = (gamma - 1.0)*el, pr = (gamma - 1.0)*er;
  const double El = el + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(wl)) + pml + sqr(bn);
  const double Er = er + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(wr)) + pmr + sqr(bn);
  double sum[7];                                      // F(L) + F(R)
  sum[0] = ml + mr;
  sum[1] = ml*ul + mr*ur + pml + pmr;
  sum[2] = ml*vl + mr*vr - bn*(byl + byr);
  sum[3] = ml*wl + mr*wr - bn*(bzl + bzr);
  sum[5] = byl*ul + byr*ur - bn*(vl + vr);
  sum[6] = bzl*ul + bzr*ur - bn*(wl + wr);
  sum[1] += (pl + pr);
  sum[4] = (El + pl + pml)*ul + (Er + pr + pmr)*ur;
  sum[4] -= bn*(byl*vl + bzl*wl);
  sum[4] -= bn*(byr*vr + bzr*wr);
  const double cfl = fast_speed(gamma, dl, pl, bn, byl, bzl), cfr = fast_speed(gamma, dr, pr, bn, byr, bzr);
  const double smax = fmax((fabs(ul) + cfl), (fabs(ur) + cfr));
  Cons1D f;
  f.d = 0.5*(sum[0] - smax*(dr - dl));
  f.mx = 0.5*(sum[1] - smax*(dr*ur - dl*ul));
  f.my = 0.5*(sum[2] - smax*(dr*vr - dl*vl));
  f.mz = 0.5*(sum[3] - smax*(dr*wr - dl*wl));
  f.e = 0.5*(sum[4] - smax*(Er - El));
  f.by = 0.5*(sum[5] - smax*(byr - byl));
  f.bz = 0.5*(sum[6] - smax*(bzr - bzl));
  return f;
}

// Advect for MHD, src/mhd/rsolvers/advect_mhd.hpp:18-58 (kinematic runs).  .e is NOT a flux: the reference leaves the
// energy flux untouched, and so do the callers of this function.
AKMI_DEV Cons1D advect_mhd(double dl, double ul, double vl, double wl, double byl, double bzl, double dr, double ur,
                           double vr, double wr, double byr, double bzr, double bn) {
  const bool from_left = ul >= 0.0;
  const double d = from_left ? dl : dr, u = from_left ? ul : ur, v = from_left ? vl : vr, w = from_left ? wl : wr;
  const double by = from_left ? byl : byr, bz = from_left ? bzl : bzr;
  Cons1D f;
  f.my = 0.0; f.mz = 0.0; f.e = 0.0;
  f.d = d*u; f.mx = d*u*u;
  f.by = -(-by*u + bn*v);                // the caller stores ey = -f.by
  f.bz = bz*u - bn*w;
  return f;
}

// Roe averages every HLLE variant for MHD starts from (hlle_mhd.hpp:45-72)
struct RoeMhd { double d, u, by, bz, x, y, rl, rr, inorm; };
AKMI_DEV RoeMhd roe_mhd_mean(double dl, double ul, double byl, double bzl, double dr, double ur, double byr,
                             double bzr) {
  RoeMhd m;
  m.rl = sqrt(dl);
  m.rr = sqrt(dr);
  m.inorm = 1.0/(m.rl + m.rr);
  m.d = m.rl*m.rr;
  m.u = (m.rl*ul + m.rr*ur)*m.inorm;
  m.by = (m.rr*byl + m.rl*byr)*m.inorm;
  m.bz = (m.rr*bzl + m.rl*bzr)*m.inorm;
  m.x = 0.5*(sqr(byl - byr) + sqr(bzl - bzr))/(sqr(m.rl + m.rr));
  m.y = 0.5*(dl + dr)/m.d;
  return m;
}
// fast speed of the Roe state from its three squared speeds (hlle_mhd.hpp:97-108)
AKMI_DEV double roe_fast(double vaxsq, double ct2, double asq) {
  const double total = vaxsq + ct2 + asq;
  const double diff = vaxsq + ct2 - asq;
  return sqrt(0.5*(total + sqrt(diff*diff + 4.0*asq*ct2)));
}
AKMI_DEV void hll_blend(const Cons1D &fl, const Cons1D &fr, double t, bool with_e, Cons1D &f) {
  f.d = 0.5*(fl.d + fr.d) + (fl.d - fr.d)*t;
  f.mx = 0.5*(fl.mx + fr.mx) + (fl.mx - fr.mx)*t;
  f.my = 0.5*(fl.my + fr.my) + (fl.my - fr.my)*t;
  f.mz = 0.5*(fl.mz + fr.mz) + (fl.mz - fr.mz)*t;
  f.e = with_e ? 0.5*(fl.e + fr.e) + (fl.e - fr.e)*t : 0.0;
  f.by = 0.5*(fl.by + fr.by) + (fl.by - fr.by)*t;
  f.bz = 0.5*(fl.bz + fr.bz) + (fl.bz - fr.bz)*t;
}

// HLLE for MHD, src/mhd/rsolvers/hlle_mhd.hpp:24-178 (ideal gas)
AKMI_DEV Cons1D hlle_mhd(double gamma, double dl, double ul, double vl, double wl, double el, double byl,
                         double bzl, double dr, double ur, double vr, double wr, double er, double byr, double bzr,
                         double bn) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const double pl = (gamma - 1.0)*el, pr = (gamma - 1.0)*er;
  const RoeMhd m = roe_mhd_mean(dl, ul, byl, bzl, dr, ur, byr, bzr);
  const double v_roe = (m.rl*vl + m.rr*vr)*m.inorm, w_roe = (m.rl*wl + m.rr*wr)*m.inorm;
  const double pml = 0.5*(bn*bn + sqr(byl) + sqr(bzl)), pmr = 0.5*(bn*bn + sqr(byr) + sqr(bzr));
  const double El = pl*igm1 + 0.5*dl*(sqr(ul) + sqr(vl) + sqr(wl)) + pml;
  const double Er = pr*igm1 + 0.5*dr*(sqr(ur) + sqr(vr) + sqr(wr)) + pmr;
  const double h_roe = ((El + pl + pml)/m.rl + (Er + pr + pmr)/m.rr)*m.inorm;
  const double cfl = fast_speed(gamma, dl, pl, bn, byl, bzl), cfr = fast_speed(gamma, dr, pr, bn, byr, bzr);
  const double btsq = sqr(m.by) + sqr(m.bz);
  const double vaxsq = bn*bn/m.d;
  const double bt_starsq = (gm1 - (gm1 - 1.0)*m.y)*btsq;
  const double hp = h_roe - (vaxsq + btsq/m.d);
  const double vsq = sqr(m.u) + sqr(v_roe) + sqr(w_roe);
  const double asq = fmax((gm1*(hp - 0.5*vsq) - (gm1 - 1.0)*m.x), 0.0);
  const double c_roe = roe_fast(vaxsq, bt_starsq/m.d, asq);
  const double smin = fmin((m.u - c_roe), (ul - cfl)), smax = fmax((m.u + c_roe), (ur + cfr));
  const double bp = smax > 0.0 ? smax : 1.0e-20;
  const double bm = smin < 0.0 ? smin : -1.0e-20;
  const double rel_l = ul - bm, rel_r = ur - bp;
  Cons1D fl, fr;
  fl.d = dl*rel_l;                        fr.d = dr*rel_r;
  fl.mx = dl*ul*rel_l + pml - sqr(bn);    fr.mx = dr*ur*rel_r + pmr - sqr(bn);
  fl.my = dl*vl*rel_l - bn*byl;           fr.my = dr*vr*rel_r - bn*byr;
  fl.mz = dl*wl*rel_l - bn*bzl;           fr.mz = dr*wr*rel_r - bn*bzr;
  fl.mx += pl;                            fr.mx += pr;
  fl.e = El*rel_l + ul*(pl + pml - bn*bn);
  fr.e = Er*rel_r + ur*(pr + pmr - bn*bn);
  fl.e -= bn*(byl*vl + bzl*wl);           fr.e -= bn*(byr*vr + bzr*wr);
  fl.by = byl*rel_l - bn*vl;              fr.by = byr*rel_r - bn*vr;
  fl.bz = bzl*rel_l - bn*wl;              fr.bz = bzr*rel_r - bn*wr;
  Cons1D f;
  hll_blend(fl, fr, hll_tilt(bp, bm), true, f);
  return f;
}

// MHD_RSolver selection at compile time: RS = AKMI_RS_LLF 0, HLLE 1, HLLD 3, ADVECT 5
template <int RS, bool EO = false, bool FM = false>
AKMI_DEV Cons1D riemann_mhd(double gamma, double dl, double ul, double vl, double wl, double el, double byl,
                            double bzl, double dr, double ur, double vr, double wr, double er, double byr,
                            double bzr, double bn) {
  if constexpr (RS == 5) return advect_mhd(dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
  else if constexpr (RS == 0) return llf_mhd(gamma, dl, ul, vl, wl, el, byl, bzl, dr, ur, vr, wr, er, byr, bzr, bn);
  else if constexpr (RS == 1) return hlle_mhd(gamma, dl, ul, vl, wl, el, byl, bzl, dr, ur, vr, wr, er, byr, bzr, bn);
  else return hlld<EO, FM>(gamma, dl, ul, vl, wl, el, byl, bzl, dr, ur, vr, wr, er, byr, bzr, bn);
}

// =======================================================================================
// Isothermal EOS (EOS_Data::is_ideal == false): states (d, u, v, w[, by, bz]), pressure cs^2 d, no energy equation.
// The reference keeps these in the same source files as the ideal-gas solvers (its "if (eos.is_ideal)" branches).
// =======================================================================================
AKMI_DEV void llf_hyd_iso(double cs, double dl, double ul, double vl, double wl, double dr, double ur, double vr,
                          double wr, double &f_d, double &f_mx, double &f_my, double &f_mz) {
  const double ml = dl*ul, mr = dr*ur;
  double sum_mx = ml*ul + mr*ur;
  sum_mx += sqr(cs)*(dl + dr);
  const double smax = fmax((fabs(ul) + cs), (fabs(ur) + cs));
  f_d = 0.5*((ml + mr) - smax*(dr - dl));
  f_mx = 0.5*(sum_mx - smax*(dr*ur - dl*ul));
  f_my = 0.5*((ml*vl + mr*vr) - smax*(dr*vr - dl*vl));
  f_mz = 0.5*((ml*wl + mr*wr) - smax*(dr*wr - dl*wl));
}

AKMI_DEV void hlle_hyd_iso(double cs, double dl, double ul, double vl, double wl, double dr, double ur, double vr,
                           double wr, double &f_d, double &f_mx, double &f_my, double &f_mz) {
  const double rl = sqrt(dl), rr = sqrt(dr);
  const double inorm = 1.0/(rl + rr);
  const double u_roe = (rl*ul + rr*ur)*inorm;
  const double smin = fmin((u_roe - cs), (ul - cs)), smax = fmax((u_roe + cs), (ur + cs));
  const double bp = (smax > 0.0) ? smax : 1.0e-20;
  const double bm = (smin < 0.0) ? smin : -1.0e-20;
  const double rel_l = ul - bm, rel_r = ur - bp;
  double fl[4] = {dl*rel_l, dl*ul*rel_l, dl*vl*rel_l, dl*wl*rel_l};
  double fr[4] = {dr*rel_r, dr*ur*rel_r, dr*vr*rel_r, dr*wr*rel_r};
  fl[1] += (cs*cs)*dl;
  fr[1] += (cs*cs)*dr;
  const double t = hll_tilt(bp, bm);
  f_d = 0.5*(fl[0] + fr[0]) + t*(fl[0] - fr[0]);
  f_mx = 0.5*(fl[1] + fr[1]) + t*(fl[1] - fr[1]);
  f_my = 0.5*(fl[2] + fr[2]) + t*(fl[2] - fr[2]);
  f_mz = 0.5*(fl[3] + fr[3]) + t*(fl[3] - fr[3]);
}

// roe_hyd.hpp:40-181 with RoeFluxIso (:275-346): four waves u - cs, u, u, u + cs
AKMI_DEV void roe_hyd_iso(double cs, double dl, double ul, double vl, double wl, double dr, double ur, double vr,
                          double wr, double &f_d, double &f_mx, double &f_my, double &f_mz) {
  const double rl = sqrt(dl), rr = sqrt(dr);
  const double inorm = 1.0/(rl + rr);
  const double u = (rl*ul + rr*ur)*inorm, v = (rl*vl + rr*vr)*inorm, w = (rl*wl + rr*wr)*inorm;
  const double ml = dl*ul, mr = dr*ur;
  double fl[4] = {ml, ml*ul, ml*vl, ml*wl};
  double fr[4] = {mr, mr*ur, mr*vr, mr*wr};
  fl[1] += (cs*cs)*dl;
  fr[1] += (cs*cs)*dr;
  const double jump[4] = {dr - dl, dr*ur - dl*ul, dr*vr - dl*vl, dr*wr - dl*wl};
  double f[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  const double lam_lo = u - cs, lam_hi = u + cs;
  double a_lo = jump[0]*(0.5 + 0.5*u/cs);
  a_lo -= jump[1]*0.5/cs;
  double a_v = jump[0]*(-v);
  a_v += jump[2];
  double a_w = jump[0]*(-w);
  a_w += jump[3];
  double a_hi = jump[0]*(0.5 - 0.5*u/cs);
  a_hi += jump[1]*0.5/cs;
  const double k_lo = -0.5*fabs(lam_lo)*a_lo, k_v = -0.5*fabs(u)*a_v, k_w = -0.5*fabs(u)*a_w,
               k_hi = -0.5*fabs(lam_hi)*a_hi;
  bool fallback = false;
  double dmid = dl + a_lo;
  if (dmid < 0.0) fallback = true;
  dmid += a_hi;
  if (dmid < 0.0) fallback = true;
  f[0] += k_lo;
  f[0] += k_hi;
  f[1] += k_lo*(u - cs);
  f[1] += k_hi*(u + cs);
  f[2] += k_lo*v;
  f[2] += k_v;
  f[2] += k_hi*v;
  f[3] += k_lo*w;
  f[3] += k_w;
  f[3] += k_hi*w;
  if (lam_lo >= 0.0) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = fl[n];
  }
  if (lam_hi <= 0.0) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = fr[n];
  }
  if (fallback) {
    const double half_smax = 0.5*fmax((fabs(ul) + cs), (fabs(ur) + cs));
#pragma unroll
    for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]) - half_smax*jump[n];
  }
  f_d = f[0]; f_mx = f[1]; f_my = f[2]; f_mz = f[3];
}

// RS as in riemann_hyd (hllc does not exist for the isothermal EOS)
template <int RS>
AKMI_DEV void riemann_hyd_iso(double cs, double dl, double ul, double vl, double wl, double dr, double ur,
                              double vr, double wr, double &f_d, double &f_mx, double &f_my, double &f_mz) {
  if constexpr (RS == 0) llf_hyd_iso(cs, dl, ul, vl, wl, dr, ur, vr, wr, f_d, f_mx, f_my, f_mz);
  else if constexpr (RS == 1) hlle_hyd_iso(cs, dl, ul, vl, wl, dr, ur, vr, wr, f_d, f_mx, f_my, f_mz);
  else if constexpr (RS == 5) {
    double unused;
    advect_hyd(dl, ul, vl, wl, 0.0, dr, ur, vr, wr, 0.0, f_d, f_mx, f_my, f_mz, unused);
  } else roe_hyd_iso(cs, dl, ul, vl, wl, dr, ur, vr, wr, f_d, f_mx, f_my, f_mz);
}

// isothermal fast speed, src/eos/eos.hpp:60-68: a^2 = cs^2 in the formula of fast_speed_of
AKMI_DEV double fast_speed_iso(double cs, double d, double bn, double by, double bz) {
  return fast_speed_of<false>((cs*cs)*d, d, bn, by, bz);
}

AKMI_DEV Cons1D llf_mhd_iso(double cs, double dl, double ul, double vl, double wl, double byl, double bzl,
                            double dr, double ur, double vr, double wr, double byr, double bzr, double bn) {
  const double ml = dl*ul, mr = dr*ur;
  const double pml = 0.5*(sqr(byl) + sqr(bzl) - sqr(bn)), pmr = 0.5*(sqr(byr) + sqr(bzr) - sqr(bn));
  double sum_mx = ml*ul + mr*ur + pml + pmr;
  sum_mx += sqr(cs)*(dl + dr);
  const double cfl = fast_speed_iso(cs, dl, bn, byl, bzl), cfr = fast_speed_iso(cs, dr, bn, byr, bzr);
  const double smax = fmax((fabs(ul) + cfl), (fabs(ur) + cfr));
  Cons1D f;
  f.d = 0.5*((ml + mr) - smax*(dr - dl));
  f.mx = 0.5*(sum_mx - smax*(dr*ur - dl*ul));
  f.my = 0.5*((ml*vl + mr*vr - bn*(byl + byr)) - smax*(dr*vr - dl*vl));
  f.mz = 0.5*((ml*wl + mr*wr - bn*(bzl + bzr)) - smax*(dr*wr - dl*wl));
  f.e = 0.0;
  f.by = 0.5*((byl*ul + byr*ur - bn*(vl + vr)) - smax*(byr - byl));
  f.bz = 0.5*((bzl*ul + bzr*ur - bn*(wl + wr)) - smax*(bzr - bzl));
  return f;
}

AKMI_DEV Cons1D hlle_mhd_iso(double cs, double dl, double ul, double vl, double wl, double byl, double bzl,
                             double dr, double ur, double vr, double wr, double byr, double bzr, double bn) {
  const RoeMhd m = roe_mhd_mean(dl, ul, byl, bzl, dr, ur, byr, bzr);
  const double pml = 0.5*(bn*bn + sqr(byl) + sqr(bzl)), pmr = 0.5*(bn*bn + sqr(byr) + sqr(bzr));
  const double cfl = fast_speed_iso(cs, dl, bn, byl, bzl), cfr = fast_speed_iso(cs, dr, bn, byr, bzr);
  const double btsq = sqr(m.by) + sqr(m.bz);
  const double vaxsq = bn*bn/m.d;
  const double bt_starsq = btsq*m.y;
  const double asq = cs*cs + m.x;
  const double c_roe = roe_fast(vaxsq, bt_starsq/m.d, asq);
  const double smin = fmin((m.u - c_roe), (ul - cfl)), smax = fmax((m.u + c_roe), (ur + cfr));
  const double bp = smax > 0.0 ? smax : 1.0e-20;
  const double bm = smin < 0.0 ? smin : -1.0e-20;
  const double rel_l = ul - bm, rel_r = ur - bp;
  Cons1D fl, fr;
  fl.d = dl*rel_l;                        fr.d = dr*rel_r;
  fl.mx = dl*ul*rel_l + pml - sqr(bn);    fr.mx = dr*ur*rel_r + pmr - sqr(bn);
  fl.my = dl*vl*rel_l - bn*byl;           fr.my = dr*vr*rel_r - bn*byr;
  fl.mz = dl*wl*rel_l - bn*bzl;           fr.mz = dr*wr*rel_r - bn*bzr;
  fl.mx += (cs*cs)*dl;                    fr.mx += (cs*cs)*dr;
  fl.e = 0.0;                             fr.e = 0.0;
  fl.by = byl*rel_l - bn*vl;              fr.by = byr*rel_r - bn*vr;
  fl.bz = bzl*rel_l - bn*wl;              fr.bz = bzr*rel_r - bn*wr;
  Cons1D f;
  hll_blend(fl, fr, hll_tilt(bp, bm), false, f);
  return f;
}

// isothermal HLLD (Mignone 2007), src/mhd/rsolvers/hlld_mhd.hpp:349-545.  Three regions between the outer waves sL, sR:
// left star, centre (between the rotational waves sAL, sAR around the HLL-averaged speed), right star; density and
// normal momentum are the HLL averages throughout the fan.
//   value                               reference lines        value                        reference lines
//   sL, sR, total pressures, F_l, F_r    :389-421               transverse star, per side     :455-490
//   HLL averages d*, F(d), F(mx), u*     :424-437               centre state                  :493-499
//   rotational speeds                    :440-441               selection                     :502-543
struct IsoT { double my, mz, by, bz; };
// :455-470 / :475-490 -- s = outer speed of the side, (d, u, v, w, by, bz) its state, (my, mz) its momenta
AKMI_DEV IsoT hlld_iso_transverse(double s, double sAL, double sAR, bool near, double dhll, double ustar, double d,
                                  double u, double v, double w, double my, double mz, double by, double bz,
                                  double bn, double bn2) {
  IsoT t;
  const double span = (s - sAL)*(s - sAR);
  if (near) {                                          // rotational and outer wave coincide on this side
    t.my = my; t.mz = mz; t.by = by; t.bz = bz;
  } else {
    const double km = bn*(ustar - u)/span;
    const double kb = (d*sqr(s - u) - bn2)/(dhll*span);
    t.my = dhll*v - by*km;
    t.mz = dhll*w - bz*km;
    t.by = by*kb;
    t.bz = bz*kb;
  }
  return t;
}
AKMI_DEV Cons1D hlld_iso(double cs, double dfloor_, double dl, double ul, double vl, double wl, double byl,
                         double bzl, double dr, double ur, double vr, double wr, double byr, double bzr, double bn) {
  constexpr double SMALL = 1.0e-4;
  const double mxl = ul*dl, myl = vl*dl, mzl = wl*dl, mxr = ur*dr, myr = vr*dr, mzr = wr*dr;
  const double cfl = fast_speed_iso(cs, dl, bn, byl, bzl), cfr = fast_speed_iso(cs, dr, bn, byr, bzr);
  const double sL = fmin(ul - cfl, ur - cfr), sR = fmax(ul + cfl, ur + cfr);
  const double bn2 = bn*bn;
  const double ptl = sqr(cs)*dl + 0.5*(bn2 + sqr(byl) + sqr(bzl)), ptr = sqr(cs)*dr + 0.5*(bn2 + sqr(byr) + sqr(bzr));
  Cons1D fl, fr;
  fl.d = mxl;                     fr.d = mxr;
  fl.mx = mxl*ul + ptl - bn2;     fr.mx = mxr*ur + ptr - bn2;
  fl.my = myl*ul - bn*byl;        fr.my = myr*ur - bn*byr;
  fl.mz = mzl*ul - bn*bzl;        fr.mz = mzr*ur - bn*bzr;
  fl.by = byl*ul - bn*vl;         fr.by = byr*ur - bn*vr;
  fl.bz = bzl*ul - bn*wl;         fr.bz = bzr*ur - bn*wr;
  // HLL averages over the fan
  const double ifan = 1.0/(sR - sL);
  double dhll = (sR*dr - sL*dl - fr.d + fl.d)*ifan;
  dhll = fmax(dhll, dfloor_);
  const double rhll = sqrt(dhll);
  const double fdhll = (sR*fl.d - sL*fr.d + sR*sL*(dr - dl))*ifan;
  const double fmxhll = (sR*fl.mx - sL*fr.mx + sR*sL*(mxr - mxl))*ifan;
  const double ustar = fdhll/dhll;
  const double mxhll = (sR*mxr - sL*mxl - fr.mx + fl.mx)*ifan;
  const double sAL = ustar - fabs(bn)/rhll, sAR = ustar + fabs(bn)/rhll;
  const IsoT Tl = hlld_iso_transverse(sL, sAL, sAR, fabs(sL - sAL) < (SMALL)*cs, dhll, ustar, dl, ul, vl, wl, myl,
                                      mzl, byl, bzl, bn, bn2);
  const IsoT Tr = hlld_iso_transverse(sR, sAL, sAR, fabs(sR - sAR) < (SMALL)*cs, dhll, ustar, dr, ur, vr, wr, myr,
                                      mzr, byr, bzr, bn, bn2);
  // centre state
  const double x = rhll*(bn > 0.0 ? 1.0 : -1.0);
  const double c_my = 0.5*(Tl.my + Tr.my + (Tr.by - Tl.by)*x);
  const double c_mz = 0.5*(Tl.mz + Tr.mz + (Tr.bz - Tl.bz)*x);
  const double c_by = 0.5*(Tl.by + Tr.by + (Tr.my - Tl.my)/x);
  const double c_bz = 0.5*(Tl.bz + Tr.bz + (Tr.mz - Tl.mz)/x);
  Cons1D f;
  f.e = 0.0;
  if (sL >= 0.0) {
    f.d = fl.d; f.mx = fl.mx; f.my = fl.my; f.mz = fl.mz; f.by = fl.by; f.bz = fl.bz;
  } else if (sR <= 0.0) {
    f.d = fr.d; f.mx = fr.mx; f.my = fr.my; f.mz = fr.mz; f.by = fr.by; f.bz = fr.bz;
  } else if (sAL >= 0.0) {
    f.d = fl.d + sL*(dhll - dl);
    f.mx = fl.mx + sL*(mxhll - mxl);
    f.my = fl.my + sL*(Tl.my - myl);
    f.mz = fl.mz + sL*(Tl.mz - mzl);
    f.by = fl.by + sL*(Tl.by - byl);
    f.bz = fl.bz + sL*(Tl.bz - bzl);
  } else if (sAR <= 0.0) {
    f.d = fr.d + sR*(dhll - dr);
    f.mx = fr.mx + sR*(mxhll - mxr);
    f.my = fr.my + sR*(Tr.my - myr);
    f.mz = fr.mz + sR*(Tr.mz - mzr);
    f.by = fr.by + sR*(Tr.by - byr);
    f.bz = fr.bz + sR*(Tr.bz - bzr);
  } else {
    f.d = dhll*ustar;
    f.mx = fmxhll;
    f.my = c_my*ustar - bn*c_by;
    f.mz = c_mz*ustar - bn*c_bz;
    f.by = c_by*ustar - bn*c_my/dhll;
    f.bz = c_bz*ustar - bn*c_mz/dhll;
  }
  return f;
}

template <int RS>
AKMI_DEV Cons1D riemann_mhd_iso(const FaceEos &eos, double dl, double ul, double vl, double wl, double byl,
                                double bzl, double dr, double ur, double vr, double wr, double byr, double bzr,
                                double bn) {
  if constexpr (RS == 5) return advect_mhd(dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
  else if constexpr (RS == 0) return llf_mhd_iso(eos.iso_cs, dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
  else if constexpr (RS == 1) return hlle_mhd_iso(eos.iso_cs, dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
  else return hlld_iso(eos.iso_cs, eos.dfloor, dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
}

// The fused stage kernels carry the equation of state in their Riemann-solver template parameter:
// RS >= 10 is the isothermal solver RS - 10 (llf 10, hlle 11, hlld 13, roe 14).  Isothermal states
// have no energy variable: slot 4 of the kernels' variable arrays stays unused (rs_iso<RS>()).
template <int RS> constexpr bool rs_iso() { return RS >= 10; }
template <int RS, bool EO = false, bool FM = false>
AKMI_DEV Cons1D riemann_mhd_e(const FaceEos &eos, double dl, double ul, double vl, double wl, double el,
                              double byl, double bzl, double dr, double ur, double vr, double wr, double er,
                              double byr, double bzr, double bn) {
  if constexpr (RS >= 10) return riemann_mhd_iso<RS - 10>(eos, dl, ul, vl, wl, byl, bzl, dr, ur, vr, wr, byr, bzr, bn);
  else return riemann_mhd<RS, EO, FM>(eos.gamma, dl, ul, vl, wl, el, byl, bzl, dr, ur, vr, wr, er, byr, bzr, bn);
}
template <int RS, bool FM = false>
AKMI_DEV void riemann_hyd_e(const FaceEos &eos, double dl, double ul, double vl, double wl, double el, double dr,
                            double ur, double vr, double wr, double er, double &f_d, double &f_mx, double &f_my,
                            double &f_mz, double &f_e) {
  if constexpr (RS >= 10) {
    riemann_hyd_iso<RS - 10>(eos.iso_cs, dl, ul, vl, wl, dr, ur, vr, wr, f_d, f_mx, f_my, f_mz);
    f_e = 0.0;
  } else {
    riemann_hyd<RS, FM>(eos.gamma, dl, ul, vl, wl, el, dr, ur, vr, wr, er, f_d, f_mx, f_my, f_mz, f_e);
  }
}

// =======================================================================================
// Conserved -> primitive, ideal gas (src/eos/ideal_c2p_hyd.hpp:22-66, ideal_c2p_mhd.hpp:20-67)
// =======================================================================================
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
// COLD: the stage kernel that converts the cells it finishes (k_hydro_stage3d2<.., C2P>) takes the exact expression through a
// real call -- inlined there, the constants of pow()'s polynomials are hoisted out of the k-loop into registers the kernel does
// not have and spilled (156 B of scratch, reloaded every step); the conversion kernels keep it inline
template <bool COLD = false> struct SpeExact {
  static AKMI_DEV double over_eps(double wd, double gm1) { return gm1/pow(wd, gm1); }
};
__device__ __attribute__((noinline)) inline double spe_over_eps_call(double wd, double gm1) { return gm1/pow(wd, gm1); }
template <> struct SpeExact<true> {
  static AKMI_DEV double over_eps(double wd, double gm1) { return spe_over_eps_call(wd, gm1); }
};
template <bool COLD = false>
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
  spe_over_eps = SpeExact<COLD>::over_eps(wd, gm1);
  double spe = spe_over_eps*we*di;
  return (spe <= sfloor);
}

// The conversion of one cell; e_other = every energy that is not thermal (kinetic [+ magnetic]).  Floors in the
// reference's order: density (caller), internal energy, temperature, entropy; ue is rewritten where a floor acts
// on the energy (the entropy floor only resets the primitive, as the reference does).
template <bool COLD = false>
AKMI_DEV void c2p_thermal(const Eos &eos, double wd, double di, double e_kin, double e_mag, bool mhd, double &ue,
                          double &we, bool &efl, bool &tfl) {
  const double efloor = eos.pfloor/(eos.gamma - 1.0);
  const double gm1 = eos.gamma - 1.0;
  we = mhd ? (ue - e_kin - e_mag) : (ue - e_kin);
  if (we < efloor) { we = efloor; ue = mhd ? (efloor + e_kin + e_mag) : (efloor + e_kin); efl = true; }
  if (gm1*we*di < eos.tfloor) { we = wd*eos.tfloor/gm1; ue = mhd ? (we + e_kin + e_mag) : (we + e_kin); tfl = true; }
  double spe_over_eps;
  if (entropy_floor_hit<COLD>(wd, we, di, gm1, eos.sfloor, spe_over_eps)) {
    we = wd*eos.sfloor/spe_over_eps;
    efl = true;
  }
}

// SingleC2P_IdealHyd
template <bool COLD = false>
AKMI_DEV void c2p_hyd(const Eos &eos, double &ud, double umx, double umy, double umz, double &ue, double &wd,
                      double &wvx, double &wvy, double &wvz, double &we, bool &dfl, bool &efl, bool &tfl) {
  if (ud < eos.dfloor) { ud = eos.dfloor; dfl = true; }
  wd = ud;
  const double di = 1.0/ud;
  wvx = di*umx; wvy = di*umy; wvz = di*umz;
  const double e_kin = 0.5*di*(sqr(umx) + sqr(umy) + sqr(umz));
  c2p_thermal<COLD>(eos, wd, di, e_kin, 0.0, false, ue, we, efl, tfl);
}

// SingleC2P_IdealMHD: (ubx, uby, ubz) the cell-centred field; the density floor rises with b^2/sigma_max
AKMI_DEV void c2p_mhd(const Eos &eos, double &ud, double umx, double umy, double umz, double &ue, double ubx,
                      double uby, double ubz, double &wd, double &wvx, double &wvy, double &wvz, double &we,
                      bool &dfl, bool &efl, bool &tfl) {
  const double bsq = sqr(ubx) + sqr(uby) + sqr(ubz);
  const double dfl_here = fmax(eos.dfloor, bsq/eos.sigma_max);
  if (ud < dfl_here) { ud = dfl_here; dfl = true; }
  wd = ud;
  const double di = 1.0/ud;
  wvx = di*umx; wvy = di*umy; wvz = di*umz;
  const double e_kin = 0.5*di*(sqr(umx) + sqr(umy) + sqr(umz));
  const double e_mag = 0.5*(sqr(ubx) + sqr(uby) + sqr(ubz));
  c2p_thermal(eos, wd, di, e_kin, e_mag, true, ue, we, efl, tfl);
}

}  // namespace akmi
#endif  // AKMI_NUMERICS_HPP_
