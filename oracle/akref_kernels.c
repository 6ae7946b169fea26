/* akref_kernels.c -- CPU ORACLE (test infrastructure, see akref.h): per-task restatements
 * of the reference's kernels in the reference's own split order.  Arithmetic follows the
 * cited reference lines operation by operation (same parenthesisation, divisions kept as
 * divisions) so results are bit-comparable with a -ffp-contract=off build.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "akref.h"

#define SQR(x) ((x)*(x))
enum { IDN = 0, IVX = 1, IVY = 2, IVZ = 3, IEN = 4 };
enum { IBX = 0, IBY = 1, IBZ = 2 };

static int g_threads = 1;
void akref_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}
int akref_get_threads(void) { return g_threads; }

/* index bookkeeping: RegionIndcs, src/mesh/mesh.cpp:285-330 */
typedef struct {
  int nmb, nvar, nx1, nx2, nx3, ng, N1, N2, N3;
  int is, ie, js, je, ks, ke, multi_d, three_d;
} G;

static G mkG(const akmi_pack *p) {
  G g;
  g.nmb = p->nmb; g.nvar = p->nvar; g.nx1 = p->nx1; g.nx2 = p->nx2; g.nx3 = p->nx3;
  g.ng = p->ng;
  g.multi_d = (p->nx2 > 1); g.three_d = (p->nx3 > 1);
  g.N1 = p->nx1 + 2*p->ng;
  g.N2 = g.multi_d ? p->nx2 + 2*p->ng : 1;
  g.N3 = g.three_d ? p->nx3 + 2*p->ng : 1;
  g.is = p->ng; g.ie = g.is + p->nx1 - 1;
  g.js = g.multi_d ? p->ng : 0; g.je = g.multi_d ? g.js + p->nx2 - 1 : 0;
  g.ks = g.three_d ? p->ng : 0; g.ke = g.three_d ? g.ks + p->nx3 - 1 : 0;
  return g;
}

static inline size_t ix5(int nv, int n3, int n2, int n1, int m, int n, int k, int j, int i) {
  return ((((size_t)m*nv + n)*n3 + k)*n2 + j)*n1 + i;
}
static inline size_t ix4(int n3, int n2, int n1, int m, int k, int j, int i) {
  return (((size_t)m*n3 + k)*n2 + j)*n1 + i;
}

/* cached scratch (wl/wr/bl/br L/R buffers etc., src/hydro/hydro.hpp:102-113) */
#define NWS 12
static double *g_ws[NWS];
static size_t g_wsn[NWS];
static double *ws_get(int slot, size_t n) {
  if (g_wsn[slot] < n) {
    free(g_ws[slot]);
    g_ws[slot] = (double *)malloc(n*sizeof(double));
    g_wsn[slot] = n;
    memset(g_ws[slot], 0, n*sizeof(double));
  }
  return g_ws[slot];
}

/* ------------------------------------------------------------------------------------
 * Reconstruction: src/reconstruct/plm.hpp:20-37, src/reconstruct/ppm.hpp:44-77 */
void akref_plm(double q_im1, double q_i, double q_ip1, double *ql_ip1, double *qr_i) {
  double dql = (q_i - q_im1);
  double dqr = (q_ip1 - q_i);
  double dq2 = dql*dqr;
  double dqm = dq2/(dql + dqr);
  if (dq2 <= 0.0) dqm = 0.0;
  *ql_ip1 = q_i + dqm;
  *qr_i   = q_i - dqm;
}

void akref_ppm4(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                double *ql_ip1, double *qr_i) {
  double qlv = (7.*(q_i + q_im1) - (q_im2 + q_ip1))/12.0;
  double qrv = (7.*(q_i + q_ip1) - (q_im1 + q_ip2))/12.0;
  qlv = fmax(qlv, fmin(q_i, q_im1));
  qlv = fmin(qlv, fmax(q_i, q_im1));
  qrv = fmax(qrv, fmin(q_i, q_ip1));
  qrv = fmin(qrv, fmax(q_i, q_ip1));
  double qc = qrv - q_i;
  double qd = qlv - q_i;
  if ((qc*qd) >= 0.0) {
    qlv = q_i;
    qrv = q_i;
  } else {
    if (fabs(qc) >= 2.0*fabs(qd)) qrv = q_i - 2.0*qd;
    if (fabs(qd) >= 2.0*fabs(qc)) qlv = q_i - 2.0*qc;
  }
  *ql_ip1 = qrv;
  *qr_i   = qlv;
}

/* PPMX (Colella & Sekora extremum-preserving limiters), src/reconstruct/ppm.hpp:84-181 */
#define SGN(x) (((x) < 0.0) ? -1.0 : 1.0)          /* SIGN, src/athena.hpp:52 */
void akref_ppmx(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                double *ql_ip1, double *qr_i) {
  double qlv = (7.*(q_i + q_im1) - (q_im2 + q_ip1))/12.0;
  double qrv = (7.*(q_i + q_ip1) - (q_im1 + q_ip2))/12.0;
  /* face i-1/2: limited second derivative (:98-115) */
  double d2qc = 3.0*((q_im1 + q_i) - 2.0*qlv);
  double d2ql = (q_im2 + q_i) - 2.0*q_im1;
  double d2qr = (q_im1 + q_ip1) - 2.0*q_i;
  double d2qlim = 0.0;
  double lim_slope = fmin(fabs(d2ql), fabs(d2qr));
  if (d2qc > 0.0 && d2ql > 0.0 && d2qr > 0.0) d2qlim = SGN(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (d2qc < 0.0 && d2ql < 0.0 && d2qr < 0.0) d2qlim = SGN(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (((q_im1 - qlv)*(q_i - qlv)) > 0.0) qlv = 0.5*(q_i + q_im1) - d2qlim/6.0;
  /* face i+1/2 (:117-135) */
  d2qc = 3.0*((q_i + q_ip1) - 2.0*qrv);
  d2ql = d2qr;
  d2qr = (q_i + q_ip2) - 2.0*q_ip1;
  d2qlim = 0.0;
  lim_slope = fmin(fabs(d2ql), fabs(d2qr));
  if (d2qc > 0.0 && d2ql > 0.0 && d2qr > 0.0) d2qlim = SGN(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (d2qc < 0.0 && d2ql < 0.0 && d2qr < 0.0) d2qlim = SGN(d2qc)*fmin(1.25*lim_slope, fabs(d2qc));
  if (((q_i - qrv)*(q_ip1 - qrv)) > 0.0) qrv = 0.5*(q_i + q_ip1) - d2qlim/6.0;
  /* extrema (:137-166) or CW monotonisation (:167-177) */
  double qa = (qrv - q_i)*(q_i - qlv);
  double qb = (q_im1 - q_i)*(q_i - q_ip1);
  if (qa <= 0.0 || qb <= 0.0) {
    double d2q = 6.0*(qlv + qrv - 2.0*q_i);
    double e2qc = (q_im1 + q_ip1) - 2.0*q_i;
    double e2ql = (q_im2 + q_i) - 2.0*q_im1;
    double e2qr = (q_i + q_ip2) - 2.0*q_ip1;
    d2qlim = 0.0;
    lim_slope = fmin(fabs(e2ql), fabs(e2qr));
    lim_slope = fmin(fabs(e2qc), lim_slope);
    if (e2qc > 0.0 && e2ql > 0.0 && e2qr > 0.0 && d2q > 0.0)
      d2qlim = SGN(d2q)*fmin(1.25*lim_slope, fabs(d2q));
    if (e2qc < 0.0 && e2ql < 0.0 && e2qr < 0.0 && d2q < 0.0)
      d2qlim = SGN(d2q)*fmin(1.25*lim_slope, fabs(d2q));
    double rho = 0.0;
    if (fabs(d2q) > (1.0e-12)*fmax(fabs(q_im1), fmax(fabs(q_i), fabs(q_ip1)))) rho = d2qlim/d2q;
    qlv = q_i + (qlv - q_i)*rho;
    qrv = q_i + (qrv - q_i)*rho;
  } else {
    double qc = qrv - q_i;
    double qd = qlv - q_i;
    if (fabs(qc) >= 2.0*fabs(qd)) qrv = q_i - 2.0*qd;
    if (fabs(qd) >= 2.0*fabs(qc)) qlv = q_i - 2.0*qc;
  }
  *ql_ip1 = qrv;
  *qr_i   = qlv;
}

/* smoothness indicators shared by WENO-Z and TENO (Jiang & Shu 1996), wenoz.hpp:32-43 */
static inline void js_beta(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                           double beta[3]) {
  const double c0 = 13./12., c1 = 0.25;
  beta[0] = c0*SQR(q_im2 + q_i - 2.0*q_im1) + c1*SQR(q_im2 + 3.0*q_i - 4.0*q_im1);
  beta[1] = c0*SQR(q_im1 + q_ip1 - 2.0*q_i) + c1*SQR(q_im1 - q_ip1);
  beta[2] = c0*SQR(q_ip2 + q_i - 2.0*q_ip1) + c1*SQR(q_ip2 + 3.0*q_i - 4.0*q_ip1);
}

/* the two 5th-order face values from candidate stencils and un-normalised weights a0,a1,a2
 * (a1 shared; outer weights swap sides), wenoz.hpp:58-81 == teno.hpp:64-86 */
static inline void weno_faces(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                              double wa, double wb, double wc, double va, double vc,
                              double *ql_ip1, double *qr_i) {
  double f0 = (2.0*q_im2 - 7.0*q_im1 + 11.0*q_i);
  double f1 = (-1.0*q_im1 + 5.0*q_i + 2.0*q_ip1);
  double f2 = (2.0*q_i + 5.0*q_ip1 - q_ip2);
  double asum = 6.0*(wa + wb + wc);
  *ql_ip1 = (f0*wa + f1*wb + f2*wc)/asum;
  f0 = (2.0*q_ip2 - 7.0*q_ip1 + 11.0*q_i);
  f1 = (-1.0*q_ip1 + 5.0*q_i + 2.0*q_im1);
  f2 = (2.0*q_i + 5.0*q_im1 - q_im2);
  asum = 6.0*(va + wb + vc);
  *qr_i = (f0*va + f1*wb + f2*vc)/asum;
}

/* WENO-Z (Borges et al. 2008), src/reconstruct/wenoz.hpp:29-84 */
void akref_wenoz(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                 double *ql_ip1, double *qr_i) {
  double beta[3];
  js_beta(q_im2, q_im1, q_i, q_ip1, q_ip2, beta);
  const double epsL = 1.0e-42;
  const double tau_5 = fabs(beta[0] - beta[2]);
  double ind0 = SQR(tau_5/(beta[0] + epsL));
  double ind1 = SQR(tau_5/(beta[1] + epsL));
  double ind2 = SQR(tau_5/(beta[2] + epsL));
  weno_faces(q_im2, q_im1, q_i, q_ip1, q_ip2, 0.1*(1.0 + ind0), 0.6*(1.0 + ind1),
             0.3*(1.0 + ind2), 0.1*(1.0 + ind2), 0.3*(1.0 + ind0), ql_ip1, qr_i);
}

/* TENO (Fu et al. 2016/2019 cut-off weights), src/reconstruct/teno.hpp:30-89 */
void akref_teno(double q_im2, double q_im1, double q_i, double q_ip1, double q_ip2,
                double *ql_ip1, double *qr_i) {
  double beta[3];
  js_beta(q_im2, q_im1, q_i, q_ip1, q_ip2, beta);
  const double epsT = 1.0e-40, cT = 1.0e-6;
#define CUBE(x) ((x)*(x)*(x))
  double a0 = 1.0/SQR(CUBE(beta[0] + epsT));
  double a1 = 1.0/SQR(CUBE(beta[1] + epsT));
  double a2 = 1.0/SQR(CUBE(beta[2] + epsT));
#undef CUBE
  double asum = a0 + a1 + a2;
  double ind0 = (a0 < cT*asum ? 0.0 : 1.0);
  double ind1 = (a1 < cT*asum ? 0.0 : 1.0);
  double ind2 = (a2 < cT*asum ? 0.0 : 1.0);
  weno_faces(q_im2, q_im1, q_i, q_ip1, q_ip2, 0.1*ind0, 0.6*ind1, 0.3*ind2, 0.1*ind2, 0.3*ind0,
             ql_ip1, qr_i);
}

/* ReconCellT / ReconDispatch (src/reconstruct/recon.hpp:40-118,134-185): cell (k,j,i)
 * writes ql to face +1 along dir and qr to its own face index. */
static void recon_dir(const G *g, const akmi_pack *p, int apply_floors, int recon, int dir, int nv,
                      const double *q, double *ql, double *qr, int kl, int ku, int jl, int ju,
                      int il, int iu) {
  /* floors on the L/R states exist only in the ppmx/wenoz/teno branches and only for the
   * fluid primitives d and e (recon.hpp:59-103: dfloor, efloor = pfloor/(gamma-1)) */
  const double dfloor = p->dfloor, efloor = p->is_ideal ? p->pfloor/(p->gamma - 1.0) : 0.0;
  const int ideal = p->is_ideal;
  const int di = (dir == 0), dj = (dir == 1), dk = (dir == 2);
  const int N1 = g->N1, N2 = g->N2, N3 = g->N3;
  const long so = (long)dk*N2*N1 + (long)dj*N1 + di;   /* stencil offset */
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g->nmb; ++m)
    for (int n = 0; n < nv; ++n)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j) {
          size_t b = ix5(nv, N3, N2, N1, m, n, k, j, 0);
          for (int i = il; i <= iu; ++i) {
            size_t c = b + i;
            double a, bq;
            if (recon == AKMI_RECON_PLM) {
              akref_plm(q[c - so], q[c], q[c + so], &a, &bq);
            } else if (recon == AKMI_RECON_PPM4) {
              akref_ppm4(q[c - 2*so], q[c - so], q[c], q[c + so], q[c + 2*so], &a, &bq);
            } else if (recon == AKMI_RECON_PPMX || recon == AKMI_RECON_WENOZ ||
                       recon == AKMI_RECON_TENO) {
              if (recon == AKMI_RECON_PPMX)
                akref_ppmx(q[c - 2*so], q[c - so], q[c], q[c + so], q[c + 2*so], &a, &bq);
              else if (recon == AKMI_RECON_WENOZ)
                akref_wenoz(q[c - 2*so], q[c - so], q[c], q[c + so], q[c + 2*so], &a, &bq);
              else
                akref_teno(q[c - 2*so], q[c - so], q[c], q[c + so], q[c + 2*so], &a, &bq);
              if (apply_floors) {
                if (n == IDN) { a = fmax(a, dfloor); bq = fmax(bq, dfloor); }
                if (ideal && n == IEN) { a = fmax(a, efloor); bq = fmax(bq, efloor); }
              }
            } else {
              a = q[c]; bq = q[c];
            }
            ql[c + so] = a;
            qr[c] = bq;
          }
        }
}

/* ------------------------------------------------------------------------------------
 * HLLC, src/hydro/rsolvers/hllc_hyd.hpp:20-115.  wl/wr = (d, vx, vy, vz, e_int) in the
 * sweep-aligned frame; flx = (d, mx, my, mz, E). */
void akref_hllc(double gamma, const double wl[5], const double wr[5], double flx[5]) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  const double alpha = (gamma + 1.0)/(2.0*gamma);
  double wl_idn = wl[0], wl_ivx = wl[1], wl_ivy = wl[2], wl_ivz = wl[3];
  double wr_idn = wr[0], wr_ivx = wr[1], wr_ivy = wr[2], wr_ivz = wr[3];
  double wl_ipr = (gamma - 1.0)*wl[4];      /* IdealGasPressure, src/eos/eos.hpp:37-40 */
  double wr_ipr = (gamma - 1.0)*wr[4];
  double qa, qb, qc, qd, qe, qf;
  qa = sqrt(gamma*wl_ipr/wl_idn);           /* IdealHydroSoundSpeed, eos.hpp:43-46 */
  qb = sqrt(gamma*wr_ipr/wr_idn);
  double el = wl_ipr*igm1 + 0.5*wl_idn*(SQR(wl_ivx) + SQR(wl_ivy) + SQR(wl_ivz));
  double er = wr_ipr*igm1 + 0.5*wr_idn*(SQR(wr_ivx) + SQR(wr_ivy) + SQR(wr_ivz));
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
  flx[0] = qc*fl_d + qd*fr_d;
  flx[1] = qc*fl_mx + qd*fr_mx + qe*cp;
  flx[2] = qc*fl_my + qd*fr_my;
  flx[3] = qc*fl_mz + qd*fr_mz;
  flx[4] = qc*fl_e + qd*fr_e + qe*cp*am;
}

/* LLF (Rusanov), src/hydro/rsolvers/llf_hyd_singlestate.hpp:28-78 (ideal gas) */
void akref_llf_hyd(double gamma, const double wl[5], const double wr[5], double flx[5]) {
  double qa = wl[0]*wl[1];
  double qb = wr[0]*wr[1];
  double s_d = qa + qb;
  double s_mx = qa*wl[1] + qb*wr[1];
  double s_my = qa*wl[2] + qb*wr[2];
  double s_mz = qa*wl[3] + qb*wr[3];
  double pl = (gamma - 1.0)*wl[4];
  double pr = (gamma - 1.0)*wr[4];
  double el = wl[4] + 0.5*wl[0]*(SQR(wl[1]) + SQR(wl[2]) + SQR(wl[3]));
  double er = wr[4] + 0.5*wr[0]*(SQR(wr[1]) + SQR(wr[2]) + SQR(wr[3]));
  s_mx += (pl + pr);
  double s_e = (el + pl)*wl[1] + (er + pr)*wr[1];
  qa = sqrt(gamma*pl/wl[0]);
  qb = sqrt(gamma*pr/wr[0]);
  double a = fmax((fabs(wl[1]) + qa), (fabs(wr[1]) + qb));
  double du_d = a*(wr[0] - wl[0]);
  double du_mx = a*(wr[0]*wr[1] - wl[0]*wl[1]);
  double du_my = a*(wr[0]*wr[2] - wl[0]*wl[2]);
  double du_mz = a*(wr[0]*wr[3] - wl[0]*wl[3]);
  double du_e = a*(er - el);
  flx[0] = 0.5*(s_d - du_d);
  flx[1] = 0.5*(s_mx - du_mx);
  flx[2] = 0.5*(s_my - du_my);
  flx[3] = 0.5*(s_mz - du_mz);
  flx[4] = 0.5*(s_e - du_e);
}

/* HLLE with Roe-averaged + L/R wave-speed bounds, src/hydro/rsolvers/hlle_hyd.hpp:27-129 */
void akref_hlle_hyd(double gamma, const double wl[5], const double wr[5], double flx[5]) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0/gm1;
  double dl = wl[0], ul = wl[1], vl = wl[2], zl = wl[3], pl = (gamma - 1.0)*wl[4];
  double dr = wr[0], ur = wr[1], vr = wr[2], zr = wr[3], pr = (gamma - 1.0)*wr[4];
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_vy = (sqrtdl*vl + sqrtdr*vr)*isdlpdr;
  double roe_vz = (sqrtdl*zl + sqrtdr*zr)*isdlpdr;
  double el = pl*igm1 + 0.5*dl*(SQR(ul) + SQR(vl) + SQR(zl));
  double er = pr*igm1 + 0.5*dr*(SQR(ur) + SQR(vr) + SQR(zr));
  double hroe = ((el + pl)/sqrtdl + (er + pr)/sqrtdr)*isdlpdr;
  double qa = sqrt(gamma*pl/dl);
  double qb = sqrt(gamma*pr/dr);
  double a = hroe - 0.5*(SQR(roe_vx) + SQR(roe_vy) + SQR(roe_vz));
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
  flx[0] = 0.5*(fl_d + fr_d) + qa*(fl_d - fr_d);
  flx[1] = 0.5*(fl_mx + fr_mx) + qa*(fl_mx - fr_mx);
  flx[2] = 0.5*(fl_my + fr_my) + qa*(fl_my - fr_my);
  flx[3] = 0.5*(fl_mz + fr_mz) + qa*(fl_mz - fr_mz);
  flx[4] = 0.5*(fl_e + fr_e) + qa*(fl_e - fr_e);
}

/* Roe's linearised solver with LLF fallback, src/hydro/rsolvers/roe_hyd.hpp:40-268 (adiabatic:
 * RoeFluxAdb :183-268, eigen-decomposition of Stone et al. 2008 App. B) */
void akref_roe_hyd(double gamma, const double wl[5], const double wr[5], double flx[5]) {
  const double gm1 = gamma - 1.0;
  double wli[5], wri[5], fl[5], fr[5], du[5], ev[5], f[5];
  for (int n = 0; n < 4; ++n) { wli[n] = wl[n]; wri[n] = wr[n]; }
  wli[4] = (gamma - 1.0)*wl[4];
  wri[4] = (gamma - 1.0)*wr[4];
  double sqrtdl = sqrt(wli[0]);
  double sqrtdr = sqrt(wri[0]);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double v1 = (sqrtdl*wli[1] + sqrtdr*wri[1])*isdlpdr;
  double v2 = (sqrtdl*wli[2] + sqrtdr*wri[2])*isdlpdr;
  double v3 = (sqrtdl*wli[3] + sqrtdr*wri[3])*isdlpdr;
  double el = wli[4]/gm1 + 0.5*wli[0]*(SQR(wli[1]) + SQR(wli[2]) + SQR(wli[3]));
  double er = wri[4]/gm1 + 0.5*wri[0]*(SQR(wri[1]) + SQR(wri[2]) + SQR(wri[3]));
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
  for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  int llf_flag = 0;
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
    for (int n = 0; n < 5; ++n) co[n] = -0.5*fabs(ev[n])*a[n];
    double dens = wli[0] + a[0];
    if (dens < 0.0) llf_flag = 1;
    dens += a[3];
    if (dens < 0.0) llf_flag = 1;
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
  if (ev[0] >= 0.0) for (int n = 0; n < 5; ++n) f[n] = fl[n];      /* supersonic: upwind (:146-163) */
  if (ev[4] <= 0.0) for (int n = 0; n < 5; ++n) f[n] = fr[n];
  if (llf_flag != 0) {                                              /* negative density (:166-181) */
    double cl = sqrt(gamma*wli[4]/wli[0]);
    double cr = sqrt(gamma*wri[4]/wri[0]);
    double a = 0.5*fmax((fabs(wli[1]) + cl), (fabs(wri[1]) + cr));
    for (int n = 0; n < 5; ++n) f[n] = 0.5*(fl[n] + fr[n]) - a*du[n];
  }
  for (int n = 0; n < 5; ++n) flx[n] = f[n];
}

/* solver selection of Hydro::CalculateFluxes (hydro_fluxes.cpp: template <Hydro_RSolver>) */
/* Advect, src/hydro/rsolvers/advect_hyd.hpp:19-55 (kinematic runs): upwinded by the sign of the LEFT
 * normal velocity; transverse components are v_t*v_n (no density), energy e_int*v_n */
void akref_advect_hyd(int ideal, const double wl[5], const double wr[5], double flx[5]) {
  const double *w = (wl[1] >= 0.0) ? wl : wr;
  flx[0] = w[0]*w[1];
  flx[1] = w[0]*w[1]*w[1];
  flx[2] = w[2]*w[1];
  flx[3] = w[3]*w[1];
  if (ideal) flx[4] = w[4]*w[1];
}

static inline int hyd_riemann(int rs, double gamma, const double a[5], const double b[5],
                              double f[5]) {
  switch (rs) {
    case AKMI_RS_ADVECT: akref_advect_hyd(1, a, b, f); return 0;
    case AKMI_RS_LLF:  akref_llf_hyd(gamma, a, b, f); return 0;
    case AKMI_RS_HLLE: akref_hlle_hyd(gamma, a, b, f); return 0;
    case AKMI_RS_HLLC: akref_hllc(gamma, a, b, f); return 0;
    case AKMI_RS_ROE:  akref_roe_hyd(gamma, a, b, f); return 0;
  }
  return 1;
}

/* ---- isothermal branches (EOS_Data::is_ideal == false): states (d, vx, vy, vz), flux (d, mx,
 * my, mz).  Same source lines as the ideal-gas functions above. */
void akref_llf_hyd_iso(double cs, const double wl[4], const double wr[4], double flx[4]) {
  double qa = wl[0]*wl[1];
  double qb = wr[0]*wr[1];
  double s_d = qa + qb;
  double s_mx = qa*wl[1] + qb*wr[1];
  double s_my = qa*wl[2] + qb*wr[2];
  double s_mz = qa*wl[3] + qb*wr[3];
  s_mx += SQR(cs)*(wl[0] + wr[0]);
  qa = cs;
  qb = cs;
  double a = fmax((fabs(wl[1]) + qa), (fabs(wr[1]) + qb));
  double du_d = a*(wr[0] - wl[0]);
  double du_mx = a*(wr[0]*wr[1] - wl[0]*wl[1]);
  double du_my = a*(wr[0]*wr[2] - wl[0]*wl[2]);
  double du_mz = a*(wr[0]*wr[3] - wl[0]*wl[3]);
  flx[0] = 0.5*(s_d - du_d);
  flx[1] = 0.5*(s_mx - du_mx);
  flx[2] = 0.5*(s_my - du_my);
  flx[3] = 0.5*(s_mz - du_mz);
}

void akref_hlle_hyd_iso(double iso_cs, const double wl[4], const double wr[4], double flx[4]) {
  double dl = wl[0], ul = wl[1], vl = wl[2], zl = wl[3];
  double dr = wr[0], ur = wr[1], vr = wr[2], zr = wr[3];
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double a = iso_cs;
  double qa = iso_cs, qb = iso_cs;
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
  fl_mx += (iso_cs*iso_cs)*dl;
  fr_mx += (iso_cs*iso_cs)*dr;
  qa = 0.0;
  if (bp != bm) qa = 0.5*(bp + bm)/(bp - bm);
  flx[0] = 0.5*(fl_d + fr_d) + qa*(fl_d - fr_d);
  flx[1] = 0.5*(fl_mx + fr_mx) + qa*(fl_mx - fr_mx);
  flx[2] = 0.5*(fl_my + fr_my) + qa*(fl_my - fr_my);
  flx[3] = 0.5*(fl_mz + fr_mz) + qa*(fl_mz - fr_mz);
}

/* roe_hyd.hpp:40-181 with RoeFluxIso (:275-346) */
void akref_roe_hyd_iso(double iso_cs, const double wl[4], const double wr[4], double flx[4]) {
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
  for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]);
  int llf_flag = 0;
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
    for (int n = 0; n < 4; ++n) co[n] = -0.5*fabs(ev[n])*a[n];
    double dens = wl[0] + a[0];
    if (dens < 0.0) llf_flag = 1;
    dens += a[3];
    if (dens < 0.0) llf_flag = 1;
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
  if (ev[0] >= 0.0) for (int n = 0; n < 4; ++n) f[n] = fl[n];
  if (ev[3] <= 0.0) for (int n = 0; n < 4; ++n) f[n] = fr[n];
  if (llf_flag != 0) {
    double a = 0.5*fmax((fabs(wl[1]) + iso_cs), (fabs(wr[1]) + iso_cs));
    for (int n = 0; n < 4; ++n) f[n] = 0.5*(fl[n] + fr[n]) - a*du[n];
  }
  for (int n = 0; n < 4; ++n) flx[n] = f[n];
}

static inline int hyd_riemann_iso(int rs, double cs, const double a[4], const double b[4],
                                  double f[4]) {
  switch (rs) {
    case AKMI_RS_ADVECT: akref_advect_hyd(0, a, b, f); return 0;
    case AKMI_RS_LLF:  akref_llf_hyd_iso(cs, a, b, f); return 0;
    case AKMI_RS_HLLE: akref_hlle_hyd_iso(cs, a, b, f); return 0;
    case AKMI_RS_ROE:  akref_roe_hyd_iso(cs, a, b, f); return 0;
  }
  return 1;                       /* hllc is an ideal-gas solver (hydro.cpp: rsolver checks) */
}

/* IdealMHDFastSpeed, src/eos/eos.hpp:49-57 */
static inline double fast_speed(double gamma, double d, double p, double bx, double by,
                                double bz) {
  double asq = gamma*p;
  double ct2 = by*by + bz*bz;
  double qsq = bx*bx + ct2 + asq;
  double tmp = bx*bx + ct2 - asq;
  return sqrt(0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*ct2))/d);
}

typedef struct { double d, mx, my, mz, e, by, bz; } cons1d;
#define HLLD_SMALL_NUMBER 1.0e-4

/* HLLD ideal-gas branch, src/mhd/rsolvers/hlld_mhd.hpp:41-347.
 * wl/wr = (d, vx, vy, vz, e_int, by, bz) sweep-aligned; flx = (d,mx,my,mz,E,by,bz). */
void akref_hlld(double gamma, const double wl[7], const double wr[7], double bxi,
                double flx[7]) {
  double spd[5];
  double gm1 = gamma - 1.0;
  double igm1 = 1.0/gm1;
  double wl_idn = wl[0], wl_ivx = wl[1], wl_ivy = wl[2], wl_ivz = wl[3];
  double wl_iby = wl[5], wl_ibz = wl[6];
  double wr_idn = wr[0], wr_ivx = wr[1], wr_ivy = wr[2], wr_ivz = wr[3];
  double wr_iby = wr[5], wr_ibz = wr[6];
  double wl_ipr = (gamma - 1.0)*wl[4];
  double wr_ipr = (gamma - 1.0)*wr[4];

  double bxsq = bxi*bxi;
  double pbl = 0.5*(bxsq + (SQR(wl_iby) + SQR(wl_ibz)));
  double pbr = 0.5*(bxsq + (SQR(wr_iby) + SQR(wr_ibz)));
  double kel = 0.5*wl_idn*(SQR(wl_ivx) + (SQR(wl_ivy) + SQR(wl_ivz)));
  double ker = 0.5*wr_idn*(SQR(wr_ivx) + (SQR(wr_ivy) + SQR(wr_ivz)));

  cons1d ul, ur;
  ul.d = wl_idn; ul.mx = wl_ivx*ul.d; ul.my = wl_ivy*ul.d; ul.mz = wl_ivz*ul.d;
  ul.e = wl_ipr*igm1 + kel + pbl; ul.by = wl_iby; ul.bz = wl_ibz;
  ur.d = wr_idn; ur.mx = wr_ivx*ur.d; ur.my = wr_ivy*ur.d; ur.mz = wr_ivz*ur.d;
  ur.e = wr_ipr*igm1 + ker + pbr; ur.by = wr_iby; ur.bz = wr_ibz;

  double cfl = fast_speed(gamma, wl_idn, wl_ipr, bxi, wl_iby, wl_ibz);
  double cfr = fast_speed(gamma, wr_idn, wr_ipr, bxi, wr_iby, wr_ibz);
  spd[0] = fmin(wl_ivx - cfl, wr_ivx - cfr);
  spd[4] = fmax(wl_ivx + cfl, wr_ivx + cfr);

  double ptl = wl_ipr + pbl;
  double ptr = wr_ipr + pbr;

  cons1d fl, fr, flxi;
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

  double sdl = spd[0] - wl_ivx;
  double sdr = spd[4] - wr_ivx;
  spd[2] = (sdr*ur.mx - sdl*ul.mx + (ptl - ptr))/(sdr*ur.d - sdl*ul.d);

  double sdml = spd[0] - spd[2];
  double sdmr = spd[4] - spd[2];
  double sdml_inv = 1.0/sdml;
  double sdmr_inv = 1.0/sdmr;

  cons1d ulst, uldst, urdst, urst;
  ulst.d = ul.d*sdl*sdml_inv;
  urst.d = ur.d*sdr*sdmr_inv;
  double ulst_d_inv = 1.0/ulst.d;
  double urst_d_inv = 1.0/urst.d;
  double sqrtdl = sqrt(ulst.d);
  double sqrtdr = sqrt(urst.d);

  spd[1] = spd[2] - fabs(bxi)/sqrtdl;
  spd[3] = spd[2] + fabs(bxi)/sqrtdr;

  double ptstl = ptl + ul.d*sdl*(spd[2] - wl_ivx);
  double ptstr = ptr + ur.d*sdr*(spd[2] - wr_ivx);
  double ptst = 0.5*(ptstr + ptstl);

  ulst.mx = ulst.d*spd[2];
  if (fabs(ul.d*sdl*sdml - bxsq) < (HLLD_SMALL_NUMBER)*ptst) {
    ulst.my = ulst.d*wl_ivy;
    ulst.mz = ulst.d*wl_ivz;
    ulst.by = ul.by;
    ulst.bz = ul.bz;
  } else {
    double tmp = bxi*(sdl - sdml)/(ul.d*sdl*sdml - bxsq);
    ulst.my = ulst.d*(wl_ivy - ul.by*tmp);
    ulst.mz = ulst.d*(wl_ivz - ul.bz*tmp);
    tmp = (ul.d*SQR(sdl) - bxsq)/(ul.d*sdl*sdml - bxsq);
    ulst.by = ul.by*tmp;
    ulst.bz = ul.bz*tmp;
  }
  double vbstl = (ulst.mx*bxi + (ulst.my*ulst.by + ulst.mz*ulst.bz))*ulst_d_inv;
  ulst.e = (sdl*ul.e - ptl*wl_ivx + ptst*spd[2] +
            bxi*(wl_ivx*bxi + (wl_ivy*ul.by + wl_ivz*ul.bz) - vbstl))*sdml_inv;

  urst.mx = urst.d*spd[2];
  if (fabs(ur.d*sdr*sdmr - bxsq) < (HLLD_SMALL_NUMBER)*ptst) {
    urst.my = urst.d*wr_ivy;
    urst.mz = urst.d*wr_ivz;
    urst.by = ur.by;
    urst.bz = ur.bz;
  } else {
    double tmp = bxi*(sdr - sdmr)/(ur.d*sdr*sdmr - bxsq);
    urst.my = urst.d*(wr_ivy - ur.by*tmp);
    urst.mz = urst.d*(wr_ivz - ur.bz*tmp);
    tmp = (ur.d*SQR(sdr) - bxsq)/(ur.d*sdr*sdmr - bxsq);
    urst.by = ur.by*tmp;
    urst.bz = ur.bz*tmp;
  }
  double vbstr = (urst.mx*bxi + (urst.my*urst.by + urst.mz*urst.bz))*urst_d_inv;
  urst.e = (sdr*ur.e - ptr*wr_ivx + ptst*spd[2] +
            bxi*(wr_ivx*bxi + (wr_ivy*ur.by + wr_ivz*ur.bz) - vbstr))*sdmr_inv;

  if (0.5*bxsq < (HLLD_SMALL_NUMBER)*ptst) {
    uldst = ulst;
    urdst = urst;
  } else {
    double invsumd = 1.0/(sqrtdl + sqrtdr);
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
    tmp = spd[2]*bxi + (uldst.my*uldst.by + uldst.mz*uldst.bz)/uldst.d;
    uldst.e = ulst.e - sqrtdl*bxsig*(vbstl - tmp);
    urdst.e = urst.e + sqrtdr*bxsig*(vbstr - tmp);
  }

  uldst.d = spd[1]*(uldst.d - ulst.d);
  uldst.mx = spd[1]*(uldst.mx - ulst.mx);
  uldst.my = spd[1]*(uldst.my - ulst.my);
  uldst.mz = spd[1]*(uldst.mz - ulst.mz);
  uldst.e = spd[1]*(uldst.e - ulst.e);
  uldst.by = spd[1]*(uldst.by - ulst.by);
  uldst.bz = spd[1]*(uldst.bz - ulst.bz);

  ulst.d = spd[0]*(ulst.d - ul.d);
  ulst.mx = spd[0]*(ulst.mx - ul.mx);
  ulst.my = spd[0]*(ulst.my - ul.my);
  ulst.mz = spd[0]*(ulst.mz - ul.mz);
  ulst.e = spd[0]*(ulst.e - ul.e);
  ulst.by = spd[0]*(ulst.by - ul.by);
  ulst.bz = spd[0]*(ulst.bz - ul.bz);

  urdst.d = spd[3]*(urdst.d - urst.d);
  urdst.mx = spd[3]*(urdst.mx - urst.mx);
  urdst.my = spd[3]*(urdst.my - urst.my);
  urdst.mz = spd[3]*(urdst.mz - urst.mz);
  urdst.e = spd[3]*(urdst.e - urst.e);
  urdst.by = spd[3]*(urdst.by - urst.by);
  urdst.bz = spd[3]*(urdst.bz - urst.bz);

  urst.d = spd[4]*(urst.d - ur.d);
  urst.mx = spd[4]*(urst.mx - ur.mx);
  urst.my = spd[4]*(urst.my - ur.my);
  urst.mz = spd[4]*(urst.mz - ur.mz);
  urst.e = spd[4]*(urst.e - ur.e);
  urst.by = spd[4]*(urst.by - ur.by);
  urst.bz = spd[4]*(urst.bz - ur.bz);

  if (spd[0] >= 0.0) {
    flxi = fl;
  } else if (spd[4] <= 0.0) {
    flxi = fr;
  } else if (spd[1] >= 0.0) {
    flxi.d = fl.d + ulst.d;   flxi.mx = fl.mx + ulst.mx; flxi.my = fl.my + ulst.my;
    flxi.mz = fl.mz + ulst.mz; flxi.e = fl.e + ulst.e;
    flxi.by = fl.by + ulst.by; flxi.bz = fl.bz + ulst.bz;
  } else if (spd[2] >= 0.0) {
    flxi.d = fl.d + ulst.d + uldst.d;     flxi.mx = fl.mx + ulst.mx + uldst.mx;
    flxi.my = fl.my + ulst.my + uldst.my; flxi.mz = fl.mz + ulst.mz + uldst.mz;
    flxi.e = fl.e + ulst.e + uldst.e;
    flxi.by = fl.by + ulst.by + uldst.by; flxi.bz = fl.bz + ulst.bz + uldst.bz;
  } else if (spd[3] > 0.0) {
    flxi.d = fr.d + urst.d + urdst.d;     flxi.mx = fr.mx + urst.mx + urdst.mx;
    flxi.my = fr.my + urst.my + urdst.my; flxi.mz = fr.mz + urst.mz + urdst.mz;
    flxi.e = fr.e + urst.e + urdst.e;
    flxi.by = fr.by + urst.by + urdst.by; flxi.bz = fr.bz + urst.bz + urdst.bz;
  } else {
    flxi.d = fr.d + urst.d;   flxi.mx = fr.mx + urst.mx; flxi.my = fr.my + urst.my;
    flxi.mz = fr.mz + urst.mz; flxi.e = fr.e + urst.e;
    flxi.by = fr.by + urst.by; flxi.bz = fr.bz + urst.bz;
  }
  flx[0] = flxi.d; flx[1] = flxi.mx; flx[2] = flxi.my; flx[3] = flxi.mz; flx[4] = flxi.e;
  flx[5] = flxi.by; flx[6] = flxi.bz;
}

/* ------------------------------------------------------------------------------------ */
/* LLF for MHD, src/mhd/rsolvers/llf_mhd_singlestate.hpp:28-89 (ideal gas).  States and flux
 * as akref_hlld: w = (d,vx,vy,vz,e,by,bz), flx = (d,mx,my,mz,E,F(by),F(bz)); the caller stores
 * ey = -flx[5], ez = +flx[6]. */
void akref_llf_mhd(double gamma, const double wl[7], const double wr[7], double bxi,
                   double flx[7]) {
  double qa = wl[0]*wl[1];
  double qb = wr[0]*wr[1];
  double qc = 0.5*(SQR(wl[5]) + SQR(wl[6]) - SQR(bxi));
  double qd = 0.5*(SQR(wr[5]) + SQR(wr[6]) - SQR(bxi));
  double s_d = qa + qb;
  double s_mx = qa*wl[1] + qb*wr[1] + qc + qd;
  double s_my = qa*wl[2] + qb*wr[2] - bxi*(wl[5] + wr[5]);
  double s_mz = qa*wl[3] + qb*wr[3] - bxi*(wl[6] + wr[6]);
  double s_by = wl[5]*wl[1] + wr[5]*wr[1] - bxi*(wl[2] + wr[2]);
  double s_bz = wl[6]*wl[1] + wr[6]*wr[1] - bxi*(wl[3] + wr[3]);
  double pl = (gamma - 1.0)*wl[4];
  double pr = (gamma - 1.0)*wr[4];
  double el = wl[4] + 0.5*wl[0]*(SQR(wl[1]) + SQR(wl[2]) + SQR(wl[3])) + qc + SQR(bxi);
  double er = wr[4] + 0.5*wr[0]*(SQR(wr[1]) + SQR(wr[2]) + SQR(wr[3])) + qd + SQR(bxi);
  s_mx += (pl + pr);
  double s_e = (el + pl + qc)*wl[1] + (er + pr + qd)*wr[1];
  s_e -= bxi*(wl[5]*wl[2] + wl[6]*wl[3]);
  s_e -= bxi*(wr[5]*wr[2] + wr[6]*wr[3]);
  qa = fast_speed(gamma, wl[0], pl, bxi, wl[5], wl[6]);
  qb = fast_speed(gamma, wr[0], pr, bxi, wr[5], wr[6]);
  double a = fmax((fabs(wl[1]) + qa), (fabs(wr[1]) + qb));
  double du_d = a*(wr[0] - wl[0]);
  double du_mx = a*(wr[0]*wr[1] - wl[0]*wl[1]);
  double du_my = a*(wr[0]*wr[2] - wl[0]*wl[2]);
  double du_mz = a*(wr[0]*wr[3] - wl[0]*wl[3]);
  double du_e = a*(er - el);
  double du_by = a*(wr[5] - wl[5]);
  double du_bz = a*(wr[6] - wl[6]);
  flx[0] = 0.5*(s_d - du_d);
  flx[1] = 0.5*(s_mx - du_mx);
  flx[2] = 0.5*(s_my - du_my);
  flx[3] = 0.5*(s_mz - du_mz);
  flx[4] = 0.5*(s_e - du_e);
  flx[5] = 0.5*(s_by - du_by);      /* reference stores ey = -0.5*(...) == -(flx[5]) exactly */
  flx[6] = 0.5*(s_bz - du_bz);
}

/* HLLE for MHD with Roe-averaged fast speed (eq. B18 of Stone et al. 2008),
 * src/mhd/rsolvers/hlle_mhd.hpp:24-178 (ideal gas) */
void akref_hlle_mhd(double gamma, const double wl[7], const double wr[7], double bxi,
                    double flx[7]) {
  double gm1 = gamma - 1.0;
  double igm1 = 1.0/gm1;
  double dl = wl[0], ul = wl[1], vl = wl[2], zl = wl[3], pl = (gamma - 1.0)*wl[4], byl = wl[5],
         bzl = wl[6];
  double dr = wr[0], ur = wr[1], vr = wr[2], zr = wr[3], pr = (gamma - 1.0)*wr[4], byr = wr[5],
         bzr = wr[6];
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_d = sqrtdl*sqrtdr;
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_vy = (sqrtdl*vl + sqrtdr*vr)*isdlpdr;
  double roe_vz = (sqrtdl*zl + sqrtdr*zr)*isdlpdr;
  double roe_by = (sqrtdr*byl + sqrtdl*byr)*isdlpdr;
  double roe_bz = (sqrtdr*bzl + sqrtdl*bzr)*isdlpdr;
  double x = 0.5*(SQR(byl - byr) + SQR(bzl - bzr))/(SQR(sqrtdl + sqrtdr));
  double y = 0.5*(dl + dr)/roe_d;
  double pbl = 0.5*(bxi*bxi + SQR(byl) + SQR(bzl));
  double pbr = 0.5*(bxi*bxi + SQR(byr) + SQR(bzr));
  double el = pl*igm1 + 0.5*dl*(SQR(ul) + SQR(vl) + SQR(zl)) + pbl;
  double er = pr*igm1 + 0.5*dr*(SQR(ur) + SQR(vr) + SQR(zr)) + pbr;
  double hroe = ((el + pl + pbl)/sqrtdl + (er + pr + pbr)/sqrtdr)*isdlpdr;
  double cl = fast_speed(gamma, dl, pl, bxi, byl, bzl);
  double cr = fast_speed(gamma, dr, pr, bxi, byr, bzr);
  double btsq = SQR(roe_by) + SQR(roe_bz);
  double vaxsq = bxi*bxi/roe_d;
  double bt_starsq = (gm1 - (gm1 - 1.0)*y)*btsq;
  double hp = hroe - (vaxsq + btsq/roe_d);
  double vsq = SQR(roe_vx) + SQR(roe_vy) + SQR(roe_vz);
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
  double fl_mx = dl*ul*vxl + pbl - SQR(bxi);
  double fr_mx = dr*ur*vxr + pbr - SQR(bxi);
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
  flx[0] = 0.5*(fl_d + fr_d) + (fl_d - fr_d)*tmp;
  flx[1] = 0.5*(fl_mx + fr_mx) + (fl_mx - fr_mx)*tmp;
  flx[2] = 0.5*(fl_my + fr_my) + (fl_my - fr_my)*tmp;
  flx[3] = 0.5*(fl_mz + fr_mz) + (fl_mz - fr_mz)*tmp;
  flx[4] = 0.5*(fl_e + fr_e) + (fl_e - fr_e)*tmp;
  flx[5] = 0.5*(fl_by + fr_by) + (fl_by - fr_by)*tmp;   /* ey = -0.5*(..) - (..)*tmp == -flx[5] */
  flx[6] = 0.5*(fl_bz + fr_bz) + (fl_bz - fr_bz)*tmp;
}

/* Advect for MHD, src/mhd/rsolvers/advect_mhd.hpp:18-58 (kinematic runs): mass and normal-momentum
 * flux upwinded by the sign of the left normal velocity, transverse momentum fluxes zero, EMFs of
 * the upwind state; the ENERGY flux is not touched by the reference (flx[4] is left as it is).
 * nb = index of the first field slot (5 ideal gas, 4 isothermal) */
void akref_advect_mhd(int nb, const double wl[7], const double wr[7], double bxi, double flx[7]) {
  const double *w = (wl[1] >= 0.0) ? wl : wr;
  flx[0] = w[0]*w[1];
  flx[1] = w[0]*w[1]*w[1];
  flx[2] = 0.0;
  flx[3] = 0.0;
  /* ey = -by*vx + bxi*vy is stored as ey = -flx[nb]; ez = bz*vx - bxi*vz = flx[nb+1] */
  flx[nb] = -(-w[nb]*w[1] + bxi*w[2]);
  flx[nb + 1] = w[nb + 1]*w[1] - bxi*w[3];
}

static inline int mhd_riemann(int rs, double gamma, const double a[7], const double b[7],
                              double bxi, double f[7]) {
  switch (rs) {
    case AKMI_RS_ADVECT: akref_advect_mhd(5, a, b, bxi, f); return 0;
    case AKMI_RS_LLF:  akref_llf_mhd(gamma, a, b, bxi, f); return 0;
    case AKMI_RS_HLLE: akref_hlle_mhd(gamma, a, b, bxi, f); return 0;
    case AKMI_RS_HLLD: akref_hlld(gamma, a, b, bxi, f); return 0;
  }
  return 1;
}

/* isothermal fast speed, src/eos/eos.hpp:60-68 */
static inline double fast_speed_iso(double cs, double d, double bx, double by, double bz) {
  double asq = (cs*cs)*d;
  double ct2 = by*by + bz*bz;
  double qsq = bx*bx + ct2 + asq;
  double tmp = bx*bx + ct2 - asq;
  return sqrt(0.5*(qsq + sqrt(tmp*tmp + 4.0*asq*ct2))/d);
}

/* isothermal MHD solvers: states (d,vx,vy,vz,by,bz), flux (d,mx,my,mz,F(by),F(bz)) */
void akref_llf_mhd_iso(double cs, const double wl[6], const double wr[6], double bxi,
                       double flx[6]) {
  double qa = wl[0]*wl[1];
  double qb = wr[0]*wr[1];
  double qc = 0.5*(SQR(wl[4]) + SQR(wl[5]) - SQR(bxi));
  double qd = 0.5*(SQR(wr[4]) + SQR(wr[5]) - SQR(bxi));
  double s_d = qa + qb;
  double s_mx = qa*wl[1] + qb*wr[1] + qc + qd;
  double s_my = qa*wl[2] + qb*wr[2] - bxi*(wl[4] + wr[4]);
  double s_mz = qa*wl[3] + qb*wr[3] - bxi*(wl[5] + wr[5]);
  double s_by = wl[4]*wl[1] + wr[4]*wr[1] - bxi*(wl[2] + wr[2]);
  double s_bz = wl[5]*wl[1] + wr[5]*wr[1] - bxi*(wl[3] + wr[3]);
  s_mx += SQR(cs)*(wl[0] + wr[0]);
  qa = fast_speed_iso(cs, wl[0], bxi, wl[4], wl[5]);
  qb = fast_speed_iso(cs, wr[0], bxi, wr[4], wr[5]);
  double a = fmax((fabs(wl[1]) + qa), (fabs(wr[1]) + qb));
  flx[0] = 0.5*(s_d - a*(wr[0] - wl[0]));
  flx[1] = 0.5*(s_mx - a*(wr[0]*wr[1] - wl[0]*wl[1]));
  flx[2] = 0.5*(s_my - a*(wr[0]*wr[2] - wl[0]*wl[2]));
  flx[3] = 0.5*(s_mz - a*(wr[0]*wr[3] - wl[0]*wl[3]));
  flx[4] = 0.5*(s_by - a*(wr[4] - wl[4]));
  flx[5] = 0.5*(s_bz - a*(wr[5] - wl[5]));
}

void akref_hlle_mhd_iso(double iso_cs, const double wl[6], const double wr[6], double bxi,
                        double flx[6]) {
  double dl = wl[0], ul = wl[1], vl = wl[2], zl = wl[3], byl = wl[4], bzl = wl[5];
  double dr = wr[0], ur = wr[1], vr = wr[2], zr = wr[3], byr = wr[4], bzr = wr[5];
  double sqrtdl = sqrt(dl);
  double sqrtdr = sqrt(dr);
  double isdlpdr = 1.0/(sqrtdl + sqrtdr);
  double roe_d = sqrtdl*sqrtdr;
  double roe_vx = (sqrtdl*ul + sqrtdr*ur)*isdlpdr;
  double roe_by = (sqrtdr*byl + sqrtdl*byr)*isdlpdr;
  double roe_bz = (sqrtdr*bzl + sqrtdl*bzr)*isdlpdr;
  double x = 0.5*(SQR(byl - byr) + SQR(bzl - bzr))/(SQR(sqrtdl + sqrtdr));
  double y = 0.5*(dl + dr)/roe_d;
  double pbl = 0.5*(bxi*bxi + SQR(byl) + SQR(bzl));
  double pbr = 0.5*(bxi*bxi + SQR(byr) + SQR(bzr));
  double cl = fast_speed_iso(iso_cs, dl, bxi, byl, bzl);
  double cr = fast_speed_iso(iso_cs, dr, bxi, byr, bzr);
  double btsq = SQR(roe_by) + SQR(roe_bz);
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
  double fl_mx = dl*ul*vxl + pbl - SQR(bxi);
  double fr_mx = dr*ur*vxr + pbr - SQR(bxi);
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
  flx[0] = 0.5*(fl_d + fr_d) + (fl_d - fr_d)*tmp;
  flx[1] = 0.5*(fl_mx + fr_mx) + (fl_mx - fr_mx)*tmp;
  flx[2] = 0.5*(fl_my + fr_my) + (fl_my - fr_my)*tmp;
  flx[3] = 0.5*(fl_mz + fr_mz) + (fl_mz - fr_mz)*tmp;
  flx[4] = 0.5*(fl_by + fr_by) + (fl_by - fr_by)*tmp;
  flx[5] = 0.5*(fl_bz + fr_bz) + (fl_bz - fr_bz)*tmp;
}

/* isothermal HLLD (Mignone 2007), src/mhd/rsolvers/hlld_mhd.hpp:349-545 */
void akref_hlld_iso(double iso_cs, double dfloor_, const double wl[6], const double wr[6], double bxi,
                    double flx[6]) {
  double wl_idn = wl[0], wl_ivx = wl[1], wl_ivy = wl[2], wl_ivz = wl[3], wl_iby = wl[4], wl_ibz = wl[5];
  double wr_idn = wr[0], wr_ivx = wr[1], wr_ivy = wr[2], wr_ivz = wr[3], wr_iby = wr[4], wr_ibz = wr[5];
  double spd[5];
  double ul_d = wl_idn, ul_mx = wl_ivx*ul_d, ul_my = wl_ivy*ul_d, ul_mz = wl_ivz*ul_d;
  double ul_by = wl_iby, ul_bz = wl_ibz;
  double ur_d = wr_idn, ur_mx = wr_ivx*ur_d, ur_my = wr_ivy*ur_d, ur_mz = wr_ivz*ur_d;
  double ur_by = wr_iby, ur_bz = wr_ibz;
  double cfl = fast_speed_iso(iso_cs, wl_idn, bxi, wl_iby, wl_ibz);
  double cfr = fast_speed_iso(iso_cs, wr_idn, bxi, wr_iby, wr_ibz);
  spd[0] = fmin(wl_ivx - cfl, wr_ivx - cfr);
  spd[4] = fmax(wl_ivx + cfl, wr_ivx + cfr);
  double bxsq = bxi*bxi;
  double ptl = SQR(iso_cs)*wl_idn + 0.5*(bxsq + SQR(wl_iby) + SQR(wl_ibz));
  double ptr = SQR(iso_cs)*wr_idn + 0.5*(bxsq + SQR(wr_iby) + SQR(wr_ibz));
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
  double idspd = 1.0/(spd[4] - spd[0]);
  double dhll = (spd[4]*ur_d - spd[0]*ul_d - fr_d + fl_d)*idspd;
  dhll = fmax(dhll, dfloor_);
  double sqrtdhll = sqrt(dhll);
  double fdhll = (spd[4]*fl_d - spd[0]*fr_d + spd[4]*spd[0]*(ur_d - ul_d))*idspd;
  double fmxhll = (spd[4]*fl_mx - spd[0]*fr_mx + spd[4]*spd[0]*(ur_mx - ul_mx))*idspd;
  double ustar = fdhll/dhll;
  double mxhll = (spd[4]*ur_mx - spd[0]*ul_mx - fr_mx + fl_mx)*idspd;
  spd[1] = ustar - fabs(bxi)/sqrtdhll;
  spd[3] = ustar + fabs(bxi)/sqrtdhll;
  double ulst_my, ulst_mz, ulst_by, ulst_bz, urst_my, urst_mz, urst_by, urst_bz;
  double tmp = (spd[0] - spd[1])*(spd[0] - spd[3]);
  if (fabs(spd[0] - spd[1]) < (HLLD_SMALL_NUMBER)*iso_cs) {
    ulst_my = ul_my; ulst_mz = ul_mz; ulst_by = ul_by; ulst_bz = ul_bz;
  } else {
    double mfact = bxi*(ustar - wl_ivx)/tmp;
    double bfact = (ul_d*SQR(spd[0] - wl_ivx) - bxsq)/(dhll*tmp);
    ulst_my = dhll*wl_ivy - ul_by*mfact;
    ulst_mz = dhll*wl_ivz - ul_bz*mfact;
    ulst_by = ul_by*bfact;
    ulst_bz = ul_bz*bfact;
  }
  tmp = (spd[4] - spd[1])*(spd[4] - spd[3]);
  if (fabs(spd[4] - spd[3]) < (HLLD_SMALL_NUMBER)*iso_cs) {
    urst_my = ur_my; urst_mz = ur_mz; urst_by = ur_by; urst_bz = ur_bz;
  } else {
    double mfact = bxi*(ustar - wr_ivx)/tmp;
    double bfact = (ur_d*SQR(spd[4] - wr_ivx) - bxsq)/(dhll*tmp);
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
  if (spd[0] >= 0.0) {
    flx[0] = fl_d; flx[1] = fl_mx; flx[2] = fl_my; flx[3] = fl_mz; flx[4] = fl_by; flx[5] = fl_bz;
  } else if (spd[4] <= 0.0) {
    flx[0] = fr_d; flx[1] = fr_mx; flx[2] = fr_my; flx[3] = fr_mz; flx[4] = fr_by; flx[5] = fr_bz;
  } else if (spd[1] >= 0.0) {
    flx[0] = fl_d + spd[0]*(dhll - ul_d);
    flx[1] = fl_mx + spd[0]*(mxhll - ul_mx);
    flx[2] = fl_my + spd[0]*(ulst_my - ul_my);
    flx[3] = fl_mz + spd[0]*(ulst_mz - ul_mz);
    flx[4] = fl_by + spd[0]*(ulst_by - ul_by);
    flx[5] = fl_bz + spd[0]*(ulst_bz - ul_bz);
  } else if (spd[3] <= 0.0) {
    flx[0] = fr_d + spd[4]*(dhll - ur_d);
    flx[1] = fr_mx + spd[4]*(mxhll - ur_mx);
    flx[2] = fr_my + spd[4]*(urst_my - ur_my);
    flx[3] = fr_mz + spd[4]*(urst_mz - ur_mz);
    flx[4] = fr_by + spd[4]*(urst_by - ur_by);
    flx[5] = fr_bz + spd[4]*(urst_bz - ur_bz);
  } else {
    flx[0] = dhll*ustar;
    flx[1] = fmxhll;
    flx[2] = ucst_my*ustar - bxi*ucst_by;
    flx[3] = ucst_mz*ustar - bxi*ucst_bz;
    flx[4] = ucst_by*ustar - bxi*ucst_my/ucst_d;
    flx[5] = ucst_bz*ustar - bxi*ucst_mz/ucst_d;
  }
}

static inline int mhd_riemann_iso(int rs, double cs, double dfloor_, const double a[6],
                                  const double b[6], double bxi, double f[6]) {
  switch (rs) {
    case AKMI_RS_LLF:  akref_llf_mhd_iso(cs, a, b, bxi, f); return 0;
    case AKMI_RS_HLLE: akref_hlle_mhd_iso(cs, a, b, bxi, f); return 0;
    case AKMI_RS_HLLD: akref_hlld_iso(cs, dfloor_, a, b, bxi, f); return 0;
  }
  return 1;
}

int akref_copy_cons(const akmi_pack *p, const double *u0, double *u1) {
  G g = mkG(p);
  memcpy(u1, u0, sizeof(double)*(size_t)g.nmb*g.nvar*g.N3*g.N2*g.N1);
  return 0;
}

/* Hydro::CopyCons for integrator rk4, src/hydro/hydro_tasks.cpp:134-148 */
int akref_rk4_copy_cons(const akmi_pack *p, double delta, const double *u0, double *u1) {
  G g = mkG(p);
  for (int m = 0; m < g.nmb; ++m) for (int n = 0; n < g.nvar; ++n)
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        size_t c = ((((size_t)m*g.nvar + n)*g.N3 + k)*g.N2 + j)*g.N1 + i;
        u1[c] += delta*u0[c];
      }
  return 0;
}

/* Hydro::CalculateFluxes, src/hydro/hydro_fluxes.cpp:77-229.  ext = 1: the ranges of a run with
 * <hydro>/fofc = true (:92-101): faces and transverse cells extended by one */
static int hydro_fluxes_impl(const akmi_pack *p, int recon, int rsolver, const double *w0,
                             double *flx1, double *flx2, double *flx3, int fs, int ext) {
  if (rsolver != AKMI_RS_LLF && rsolver != AKMI_RS_HLLE && rsolver != AKMI_RS_HLLC &&
      rsolver != AKMI_RS_ROE && rsolver != AKMI_RS_ADVECT) return AKMI_FAIL;
  if (!p->is_ideal && rsolver == AKMI_RS_HLLC) return AKMI_FAIL;   /* hllc is ideal-gas only */
  const int ideal = p->is_ideal;
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  size_t ncell = (size_t)g.nmb*nv*N3*N2*N1;
  double *wl = ws_get(0, ncell), *wr = ws_get(1, ncell);
  const double gamma = p->gamma;
  for (int dir = 0; dir < 3; ++dir) {
    if (dir == 1 && !g.multi_d) continue;
    if (dir == 2 && !g.three_d) continue;
    int il = g.is, iu = g.ie, jl = g.js, ju = g.je, kl = g.ks, ku = g.ke;
    if (ext) {                       /* transverse ranges itl..itu etc., hydro_fluxes.cpp:98-100 */
      il = g.is-1; iu = g.ie+1;
      if (g.multi_d) { jl = g.js-1; ju = g.je+1; }
      if (g.three_d) { kl = g.ks-1; ku = g.ke+1; }
    }
    double *flx = flx1;
    int f3 = N3, f2 = N2, f1 = N1 + fs;
    /* face-normal ranges: [s, e+1], or [s-1, e+2] with ext (:92,95-97); cells one further left */
    if (dir == 0) { il = g.is - ext; iu = g.ie + ext;
                    recon_dir(&g, p, 1, recon, 0, nv, w0, wl, wr, kl, ku, jl, ju, il-1, iu+1); iu = iu+1; }
    if (dir == 1) { jl = g.js - ext; ju = g.je + ext;
                    recon_dir(&g, p, 1, recon, 1, nv, w0, wl, wr, kl, ku, jl-1, ju+1, il, iu); ju = ju+1;
                    flx = flx2; f1 = N1; f2 = N2 + fs; }
    if (dir == 2) { kl = g.ks - ext; ku = g.ke + ext;
                    recon_dir(&g, p, 1, recon, 2, nv, w0, wl, wr, kl-1, ku+1, jl, ju, il, iu); ku = ku+1;
                    flx = flx3; f1 = N1; f3 = N3 + fs; }
    const int ivx = IVX + dir, ivy = IVX + (dir + 1)%3, ivz = IVX + (dir + 2)%3;
#pragma omp parallel for collapse(3) schedule(static)
    for (int m = 0; m < g.nmb; ++m)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j)
          for (int i = il; i <= iu; ++i) {
            double a[5], b[5], f[5];
            a[0] = wl[ix5(nv,N3,N2,N1,m,IDN,k,j,i)]; b[0] = wr[ix5(nv,N3,N2,N1,m,IDN,k,j,i)];
            a[1] = wl[ix5(nv,N3,N2,N1,m,ivx,k,j,i)]; b[1] = wr[ix5(nv,N3,N2,N1,m,ivx,k,j,i)];
            a[2] = wl[ix5(nv,N3,N2,N1,m,ivy,k,j,i)]; b[2] = wr[ix5(nv,N3,N2,N1,m,ivy,k,j,i)];
            a[3] = wl[ix5(nv,N3,N2,N1,m,ivz,k,j,i)]; b[3] = wr[ix5(nv,N3,N2,N1,m,ivz,k,j,i)];
            if (ideal) {
              a[4] = wl[ix5(nv,N3,N2,N1,m,IEN,k,j,i)]; b[4] = wr[ix5(nv,N3,N2,N1,m,IEN,k,j,i)];
              hyd_riemann(rsolver, gamma, a, b, f);
            } else {
              hyd_riemann_iso(rsolver, p->iso_cs, a, b, f);
            }
            flx[ix5(nv,f3,f2,f1,m,IDN,k,j,i)] = f[0];
            flx[ix5(nv,f3,f2,f1,m,ivx,k,j,i)] = f[1];
            flx[ix5(nv,f3,f2,f1,m,ivy,k,j,i)] = f[2];
            flx[ix5(nv,f3,f2,f1,m,ivz,k,j,i)] = f[3];
            if (ideal) flx[ix5(nv,f3,f2,f1,m,IEN,k,j,i)] = f[4];
          }
    /* passive scalars: upwinded by the sign of the mass flux (hydro_fluxes.cpp:135-147) */
    const int nf = ideal ? 5 : 4;
    if (nv > nf) {
      const int sil = g.is, siu = g.ie + (dir == 0), sjl = g.js, sju = g.je + (dir == 1);
      const int skl = g.ks, sku = g.ke + (dir == 2);
      for (int m = 0; m < g.nmb; ++m)
        for (int k = skl; k <= sku; ++k)
          for (int j = sjl; j <= sju; ++j)
            for (int i = sil; i <= siu; ++i) {
              double fd = flx[ix5(nv,f3,f2,f1,m,IDN,k,j,i)];
              for (int n = nf; n < nv; ++n)
                flx[ix5(nv,f3,f2,f1,m,n,k,j,i)] =
                    fd*((fd >= 0.0) ? wl[ix5(nv,N3,N2,N1,m,n,k,j,i)] : wr[ix5(nv,N3,N2,N1,m,n,k,j,i)]);
            }
    }
  }
  return 0;
}

int akref_hydro_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                       double *flx1, double *flx2, double *flx3, int fs) {
  return hydro_fluxes_impl(p, recon, rsolver, w0, flx1, flx2, flx3, fs, 0);
}

int akref_hydro_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                            double *flx1, double *flx2, double *flx3, int fs) {
  if (p->nvar != (p->is_ideal ? 5 : 4)) return AKMI_FAIL;      /* FOFC + scalars: not on this path */
  return hydro_fluxes_impl(p, recon, rsolver, w0, flx1, flx2, flx3, fs, 1);
}

/* Hydro::FOFC, src/hydro/hydro_fofc.cpp:30-371 (Newtonian): trial update of the cells
 * [is-1,ie+1] x ..., flag those whose conversion to primitives needs a floor (ConsToPrim with
 * only_testfloors, src/eos/ideal_hyd.cpp:67-72, isothermal_hyd.cpp), replace the fluxes on the
 * faces of flagged cells by first-order LLF fluxes of the adjacent cell states
 * (SingleStateLLF_Hyd, src/hydro/rsolvers/llf_hyd_singlestate.hpp:27-83), reset the flags.
 * fofc: unsigned char [nmb][N3][N2][N1], all zero on entry and on exit; *nfofc += flagged cells */
int akref_hydro_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                     const double *u0, const double *u1, double *flx1, double *flx2, double *flx3,
                     int fs, unsigned char *fofc, int *nfofc) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int ideal = p->is_ideal;
  const int nhyd = ideal ? 5 : 4;
  if (nv != nhyd) return AKMI_FAIL;
  int il = g.is-1, iu = g.ie+1, jl = g.js, ju = g.je, kl = g.ks, ku = g.ke;
  if (g.multi_d) { jl = g.js-1; ju = g.je+1; }
  if (g.three_d) { kl = g.ks-1; ku = g.ke+1; }
  const double gm1 = p->gamma - 1.0;
  const double efloor = p->pfloor/gm1;
  int nflag = 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          double dtodx1 = beta_dt/p->dx[3*m];
          double dtodx2 = beta_dt/p->dx[3*m+1];
          double dtodx3 = beta_dt/p->dx[3*m+2];
          double ut[5];
          for (int n = 0; n < nhyd; ++n) {
            double divf = dtodx1*(flx1[ix5(nv,N3,N2,N1+fs,m,n,k,j,i+1)] - flx1[ix5(nv,N3,N2,N1+fs,m,n,k,j,i)]);
            if (g.multi_d)
              divf += dtodx2*(flx2[ix5(nv,N3,N2+fs,N1,m,n,k,j+1,i)] - flx2[ix5(nv,N3,N2+fs,N1,m,n,k,j,i)]);
            if (g.three_d)
              divf += dtodx3*(flx3[ix5(nv,N3+fs,N2,N1,m,n,k+1,j,i)] - flx3[ix5(nv,N3+fs,N2,N1,m,n,k,j,i)]);
            size_t c = ix5(nv,N3,N2,N1,m,n,k,j,i);
            ut[n] = gam0*u0[c] + gam1*u1[c] - divf;
          }
          int fl = 0;
          if (!ideal) {
            fl = ut[0] < p->dfloor;
          } else {          /* SingleC2P_IdealHyd, src/eos/ideal_c2p_hyd.hpp:22-66 */
            double ud = ut[0], ue = ut[4];
            if (ud < p->dfloor) { ud = p->dfloor; fl = 1; }
            double di = 1.0/ud;
            double e_k = 0.5*di*(SQR(ut[1]) + SQR(ut[2]) + SQR(ut[3]));
            double we = (ue - e_k);
            if (we < efloor) { we = efloor; fl = 1; }
            if (gm1*we*di < p->tfloor) { we = ud*p->tfloor/gm1; fl = 1; }
            double spe_over_eps = gm1/pow(ud, gm1);
            double spe = spe_over_eps*we*di;
            if (spe <= p->sfloor) fl = 1;
          }
          if (fl) { fofc[ix4(N3,N2,N1,m,k,j,i)] = 1; nflag++; }
        }
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          if (!fofc[ix4(N3,N2,N1,m,k,j,i)]) continue;
          for (int dir = 0; dir < 3; ++dir) {
            if (dir == 1 && !g.multi_d) continue;
            if (dir == 2 && !g.three_d) continue;
            const int ivx = IVX + dir, ivy = IVX + (dir + 1)%3, ivz = IVX + (dir + 2)%3;
            const int d1 = dir == 0, d2 = dir == 1, d3 = dir == 2;
            double *flx = dir == 0 ? flx1 : (dir == 1 ? flx2 : flx3);
            const int f1 = N1 + (d1 ? fs : 0), f2 = N2 + (d2 ? fs : 0), f3 = N3 + (d3 ? fs : 0);
            for (int side = 0; side < 2; ++side) {       /* face at the cell, then the next one */
              const int kf = k + side*d3, jf = j + side*d2, ifc = i + side*d1;
              double a[5], b[5], f[5];
              const int comp[5] = {IDN, ivx, ivy, ivz, IEN};
              for (int q = 0; q < nhyd; ++q) {
                a[q] = w0[ix5(nv,N3,N2,N1,m,comp[q],kf-d3,jf-d2,ifc-d1)];
                b[q] = w0[ix5(nv,N3,N2,N1,m,comp[q],kf,jf,ifc)];
              }
              if (ideal) akref_llf_hyd(p->gamma, a, b, f); else akref_llf_hyd_iso(p->iso_cs, a, b, f);
              for (int q = 0; q < nhyd; ++q) flx[ix5(nv,f3,f2,f1,m,comp[q],kf,jf,ifc)] = f[q];
            }
          }
        }
  /* "reset FOFC flag" (:364): done after all flagged cells have been processed */
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) fofc[ix4(N3,N2,N1,m,k,j,i)] = 0;
  if (nfofc) *nfofc += nflag;
  return 0;
}

/* RKUpdate, src/hydro/hydro_update.cpp:23-83 == src/mhd/mhd_update.cpp:24-84 */
int akref_rk_update(const akmi_pack *p, double gam0, double gam1, double beta_dt,
                    double *u0, const double *u1, const double *flx1, const double *flx2,
                    const double *flx3, int fs) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g.nmb; ++m)
    for (int n = 0; n < nv; ++n)
      for (int k = g.ks; k <= g.ke; ++k)
        for (int j = g.js; j <= g.je; ++j) {
          const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
          for (int i = g.is; i <= g.ie; ++i) {
            double divf = (flx1[ix5(nv,N3,N2,N1+fs,m,n,k,j,i+1)] -
                           flx1[ix5(nv,N3,N2,N1+fs,m,n,k,j,i)])/dx1;
            if (g.multi_d)
              divf += (flx2[ix5(nv,N3,N2+fs,N1,m,n,k,j+1,i)] -
                       flx2[ix5(nv,N3,N2+fs,N1,m,n,k,j,i)])/dx2;
            if (g.three_d)
              divf += (flx3[ix5(nv,N3+fs,N2,N1,m,n,k+1,j,i)] -
                       flx3[ix5(nv,N3+fs,N2,N1,m,n,k,j,i)])/dx3;
            size_t c = ix5(nv,N3,N2,N1,m,n,k,j,i);
            u0[c] = gam0*u0[c] + gam1*u1[c] - beta_dt*divf;
          }
        }
  return 0;
}

/* SingleC2P_IdealHyd, src/eos/ideal_c2p_hyd.hpp:22-66 ; wrapper src/eos/ideal_hyd.cpp:29-115 */
int akref_hydro_c2p(const akmi_pack *p, double *u0, double *w0, int il, int iu, int jl,
                    int ju, int kl, int ku, int *counters) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if (!p->is_ideal) {
    /* SingleC2P_IsothermalHyd, src/eos/isothermal_hyd.cpp:30-45,61-124 */
    int sumd_ = 0;
    for (int m = 0; m < g.nmb; ++m)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j)
          for (int i = il; i <= iu; ++i) {
            size_t cd = ix5(nv,N3,N2,N1,m,IDN,k,j,i), cx = ix5(nv,N3,N2,N1,m,IVX,k,j,i);
            size_t cy = ix5(nv,N3,N2,N1,m,IVY,k,j,i), cz = ix5(nv,N3,N2,N1,m,IVZ,k,j,i);
            double ud = u0[cd];
            if (ud < p->dfloor) { ud = p->dfloor; u0[cd] = ud; sumd_++; }
            double di = 1.0/ud;
            w0[cd] = ud; w0[cx] = di*u0[cx]; w0[cy] = di*u0[cy]; w0[cz] = di*u0[cz];
            for (int n = 4; n < nv; ++n)           /* scalars, isothermal_hyd.cpp:119-122 (no floor) */
              w0[ix5(nv,N3,N2,N1,m,n,k,j,i)] = u0[ix5(nv,N3,N2,N1,m,n,k,j,i)]/ud;
          }
    if (counters) counters[0] += sumd_;
    return 0;
  }
  const double gm1 = p->gamma - 1.0;
  const double efloor = p->pfloor/(p->gamma - 1.0);
  const double tfloor = p->tfloor, sfloor = p->sfloor, dfloor_ = p->dfloor;
  int sumd = 0, sume = 0, sumt = 0;
#pragma omp parallel for collapse(3) schedule(static) reduction(+:sumd,sume,sumt)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          size_t cd = ix5(nv,N3,N2,N1,m,IDN,k,j,i), cx = ix5(nv,N3,N2,N1,m,IVX,k,j,i);
          size_t cy = ix5(nv,N3,N2,N1,m,IVY,k,j,i), cz = ix5(nv,N3,N2,N1,m,IVZ,k,j,i);
          size_t ce = ix5(nv,N3,N2,N1,m,IEN,k,j,i);
          double ud = u0[cd], umx = u0[cx], umy = u0[cy], umz = u0[cz], ue = u0[ce];
          int dfl = 0, efl = 0, tfl = 0;
          if (ud < dfloor_) { ud = dfloor_; dfl = 1; }
          double wd = ud;
          double di = 1.0/ud;
          double wvx = di*umx, wvy = di*umy, wvz = di*umz;
          double e_k = 0.5*di*(SQR(umx) + SQR(umy) + SQR(umz));
          double we = (ue - e_k);
          if (we < efloor) { we = efloor; ue = efloor + e_k; efl = 1; }
          if (gm1*we*di < tfloor) { we = wd*tfloor/gm1; ue = we + e_k; tfl = 1; }
          double spe_over_eps = gm1/pow(wd, gm1);
          double spe = spe_over_eps*we*di;
          if (spe <= sfloor) { we = wd*sfloor/spe_over_eps; efl = 1; }
          if (dfl) { u0[cd] = ud; sumd++; }
          if (efl) { u0[ce] = ue; sume++; }
          if (tfl) { u0[ce] = ue; sumt++; }
          w0[cd] = wd; w0[cx] = wvx; w0[cy] = wvy; w0[cz] = wvz; w0[ce] = we;
          for (int n = 5; n < nv; ++n) {          /* scalars, ideal_hyd.cpp:94-101 */
            size_t cn = ix5(nv,N3,N2,N1,m,n,k,j,i);
            if (u0[cn] < 0.0) u0[cn] = 0.0;
            w0[cn] = u0[cn]/ud;
          }
        }
  if (counters) { counters[0] += sumd; counters[1] += sume; counters[2] += sumt; }
  return 0;
}

/* Hydro::NewTimeStep, src/hydro/hydro_newdt.cpp:30-139 (Newtonian ideal branch 97-118) */
int akref_hydro_newdt(const akmi_pack *p, const double *w0, double *dt3) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  double dt1 = (double)FLT_MAX, dt2 = (double)FLT_MAX, dt3_ = (double)FLT_MAX;
#pragma omp parallel for collapse(3) schedule(static) reduction(min:dt1,dt2,dt3_)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          double cs;
          if (p->is_ideal) {
            double pr = (p->gamma - 1.0)*w0[ix5(nv,N3,N2,N1,m,IEN,k,j,i)];
            cs = sqrt(p->gamma*pr/w0[ix5(nv,N3,N2,N1,m,IDN,k,j,i)]);
          } else {
            cs = p->iso_cs;                       /* hydro_newdt.cpp:109-111 */
          }
          double max_dv1 = fabs(w0[ix5(nv,N3,N2,N1,m,IVX,k,j,i)]) + cs;
          double max_dv2 = fabs(w0[ix5(nv,N3,N2,N1,m,IVY,k,j,i)]) + cs;
          double max_dv3 = fabs(w0[ix5(nv,N3,N2,N1,m,IVZ,k,j,i)]) + cs;
          dt1 = fmin(p->dx[3*m]/max_dv1, dt1);
          dt2 = fmin(p->dx[3*m+1]/max_dv2, dt2);
          dt3_ = fmin(p->dx[3*m+2]/max_dv3, dt3_);
        }
  dt3[0] = dt1; dt3[1] = dt2; dt3[2] = dt3_;
  return 0;
}

/* NewTimeStep of kinematic runs, src/hydro/hydro_newdt.cpp:55-72 == src/mhd/mhd_newdt.cpp:56-73 */
int akref_kinematic_newdt(const akmi_pack *p, const double *w0, double *dt3) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  double dt1 = (double)FLT_MAX, dt2 = (double)FLT_MAX, dt3_ = (double)FLT_MAX;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          dt1 = fmin((p->dx[3*m]/fabs(w0[ix5(nv,N3,N2,N1,m,IVX,k,j,i)])), dt1);
          dt2 = fmin((p->dx[3*m+1]/fabs(w0[ix5(nv,N3,N2,N1,m,IVY,k,j,i)])), dt2);
          dt3_ = fmin((p->dx[3*m+2]/fabs(w0[ix5(nv,N3,N2,N1,m,IVZ,k,j,i)])), dt3_);
        }
  dt3[0] = dt1; dt3[1] = dt2; dt3[2] = dt3_;
  return 0;
}

/* MHD::CalculateFluxes<hlld>, src/mhd/mhd_fluxes.cpp:84-266 */
static int mhd_fluxes_impl(const akmi_pack *p, int recon, int rsolver, const double *w0,
                     const double *bcc0, const double *bx1f, const double *bx2f,
                     const double *bx3f, double *flx1, double *flx2, double *flx3,
                     double *e3x1, double *e2x1, double *e1x2, double *e3x2, double *e2x3,
                     double *e1x3, int ext) {
  if (rsolver != AKMI_RS_LLF && rsolver != AKMI_RS_HLLE && rsolver != AKMI_RS_HLLD &&
      rsolver != AKMI_RS_ADVECT) return AKMI_FAIL;
  const int ideal = p->is_ideal;
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  size_t ncell = (size_t)g.nmb*N3*N2*N1;
  double *wl = ws_get(0, ncell*nv), *wr = ws_get(1, ncell*nv);
  double *bl = ws_get(2, ncell*3), *br = ws_get(3, ncell*3);
  const double gamma = p->gamma;
  for (int dir = 0; dir < 3; ++dir) {
    if (dir == 1 && !g.multi_d) continue;
    if (dir == 2 && !g.three_d) continue;
    int il, iu, jl, ju, kl, ku;
    const double *bx; double *flx, *ey, *ez;
    int f3 = N3, f2 = N2, f1 = N1;
    if (dir == 0) {
      jl = g.js; ju = g.je; kl = g.ks; ku = g.ke;
      if (g.multi_d) { jl = g.js-1; ju = g.je+1; }
      if (g.three_d) { kl = g.ks-1; ku = g.ke+1; }
      /* ext: face-normal range [is-1,ie+2] with <mhd>/fofc = true, src/mhd/mhd_fluxes.cpp:100-105 */
      recon_dir(&g, p, 1, recon, 0, nv, w0, wl, wr, kl, ku, jl, ju, g.is-1-ext, g.ie+1+ext);
      recon_dir(&g, p, 0, recon, 0, 3, bcc0, bl, br, kl, ku, jl, ju, g.is-1-ext, g.ie+1+ext);
      il = g.is-ext; iu = g.ie+1+ext;
      bx = bx1f; flx = flx1; ey = e3x1; ez = e2x1; f1 = N1+1;
    } else if (dir == 1) {
      kl = g.ks; ku = g.ke;
      if (g.three_d) { kl = g.ks-1; ku = g.ke+1; }
      recon_dir(&g, p, 1, recon, 1, nv, w0, wl, wr, kl, ku, g.js-1-ext, g.je+1+ext, g.is-1, g.ie+1);
      recon_dir(&g, p, 0, recon, 1, 3, bcc0, bl, br, kl, ku, g.js-1-ext, g.je+1+ext, g.is-1, g.ie+1);
      il = g.is-1; iu = g.ie+1; jl = g.js-ext; ju = g.je+1+ext;
      bx = bx2f; flx = flx2; ey = e1x2; ez = e3x2; f2 = N2+1;
    } else {
      recon_dir(&g, p, 1, recon, 2, nv, w0, wl, wr, g.ks-1-ext, g.ke+1+ext, g.js-1, g.je+1, g.is-1, g.ie+1);
      recon_dir(&g, p, 0, recon, 2, 3, bcc0, bl, br, g.ks-1-ext, g.ke+1+ext, g.js-1, g.je+1, g.is-1, g.ie+1);
      il = g.is-1; iu = g.ie+1; jl = g.js-1; ju = g.je+1; kl = g.ks-ext; ku = g.ke+1+ext;
      bx = bx3f; flx = flx3; ey = e2x3; ez = e1x3; f3 = N3+1;
    }
    const int ivx = IVX + dir, ivy = IVX + (dir + 1)%3, ivz = IVX + (dir + 2)%3;
    const int iby = (dir + 1)%3, ibz = (dir + 2)%3;
#pragma omp parallel for collapse(3) schedule(static)
    for (int m = 0; m < g.nmb; ++m)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j)
          for (int i = il; i <= iu; ++i) {
            double a[7], b[7], f[7];
            a[0] = wl[ix5(nv,N3,N2,N1,m,IDN,k,j,i)]; b[0] = wr[ix5(nv,N3,N2,N1,m,IDN,k,j,i)];
            a[1] = wl[ix5(nv,N3,N2,N1,m,ivx,k,j,i)]; b[1] = wr[ix5(nv,N3,N2,N1,m,ivx,k,j,i)];
            a[2] = wl[ix5(nv,N3,N2,N1,m,ivy,k,j,i)]; b[2] = wr[ix5(nv,N3,N2,N1,m,ivy,k,j,i)];
            a[3] = wl[ix5(nv,N3,N2,N1,m,ivz,k,j,i)]; b[3] = wr[ix5(nv,N3,N2,N1,m,ivz,k,j,i)];
            double bxi = bx[ix4(f3,f2,f1,m,k,j,i)];
            const int ob = ideal ? 5 : 4;               /* first B slot of the state vector */
            if (ideal) { a[4] = wl[ix5(nv,N3,N2,N1,m,IEN,k,j,i)]; b[4] = wr[ix5(nv,N3,N2,N1,m,IEN,k,j,i)]; }
            a[ob] = bl[ix5(3,N3,N2,N1,m,iby,k,j,i)];      b[ob] = br[ix5(3,N3,N2,N1,m,iby,k,j,i)];
            a[ob + 1] = bl[ix5(3,N3,N2,N1,m,ibz,k,j,i)];  b[ob + 1] = br[ix5(3,N3,N2,N1,m,ibz,k,j,i)];
            if (rsolver == AKMI_RS_ADVECT) akref_advect_mhd(ob, a, b, bxi, f);
            else if (ideal) mhd_riemann(rsolver, gamma, a, b, bxi, f);
            else mhd_riemann_iso(rsolver, p->iso_cs, p->dfloor, a, b, bxi, f);
            flx[ix5(nv,f3,f2,f1,m,IDN,k,j,i)] = f[0];
            flx[ix5(nv,f3,f2,f1,m,ivx,k,j,i)] = f[1];
            flx[ix5(nv,f3,f2,f1,m,ivy,k,j,i)] = f[2];
            flx[ix5(nv,f3,f2,f1,m,ivz,k,j,i)] = f[3];
            if (ideal && rsolver != AKMI_RS_ADVECT) flx[ix5(nv,f3,f2,f1,m,IEN,k,j,i)] = f[4];
            ey[ix4(N3,N2,N1,m,k,j,i)] = -f[ob];
            ez[ix4(N3,N2,N1,m,k,j,i)] = f[ob + 1];
          }
    /* passive scalars over the ACTIVE transverse range (mhd_fluxes.cpp:153-166) */
    const int nf = ideal ? 5 : 4;
    if (nv > nf) {
      const int sil = g.is, siu = g.ie + (dir == 0), sjl = g.js, sju = g.je + (dir == 1);
      const int skl = g.ks, sku = g.ke + (dir == 2);
      for (int m = 0; m < g.nmb; ++m)
        for (int k = skl; k <= sku; ++k)
          for (int j = sjl; j <= sju; ++j)
            for (int i = sil; i <= siu; ++i) {
              double fd = flx[ix5(nv,f3,f2,f1,m,IDN,k,j,i)];
              for (int n = nf; n < nv; ++n)
                flx[ix5(nv,f3,f2,f1,m,n,k,j,i)] =
                    fd*((fd >= 0.0) ? wl[ix5(nv,N3,N2,N1,m,n,k,j,i)] : wr[ix5(nv,N3,N2,N1,m,n,k,j,i)]);
            }
    }
  }
  return 0;
}

int akref_mhd_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                     const double *bcc0, const double *bx1f, const double *bx2f,
                     const double *bx3f, double *flx1, double *flx2, double *flx3,
                     double *e3x1, double *e2x1, double *e1x2, double *e3x2, double *e2x3,
                     double *e1x3) {
  return mhd_fluxes_impl(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, e3x1, e2x1,
                         e1x2, e3x2, e2x3, e1x3, 0);
}

int akref_mhd_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                          const double *bcc0, const double *bx1f, const double *bx2f,
                          const double *bx3f, double *flx1, double *flx2, double *flx3,
                          double *e3x1, double *e2x1, double *e1x2, double *e3x2, double *e2x3,
                          double *e1x3) {
  if (!p->is_ideal || p->nvar != 5) return AKMI_FAIL;     /* ideal gas, no scalars on this path */
  return mhd_fluxes_impl(p, recon, rsolver, w0, bcc0, bx1f, bx2f, bx3f, flx1, flx2, flx3, e3x1, e2x1,
                         e1x2, e3x2, e2x3, e1x3, 1);
}

/* MHD::FOFC, src/mhd/mhd_fofc.cpp:30-493 (Newtonian ideal gas): trial update of U and of the
 * cell-centred field (from the face EMFs), floor test with the trial field
 * (IdealMHD::ConsToPrim only_testfloors, src/eos/ideal_mhd.cpp:66-96), first-order LLF fluxes AND
 * face EMFs (SingleStateLLF_MHD, src/mhd/rsolvers/llf_mhd_singlestate.hpp:27-89) on the faces
 * of flagged cells, flags reset. */
int akref_mhd_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                   const double *bcc0, const double *b0x1f, const double *b0x2f, const double *b0x3f,
                   const double *b1x1f, const double *b1x2f, const double *b1x3f, const double *u0,
                   const double *u1, double *flx1, double *flx2, double *flx3, double *e3x1,
                   double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                   unsigned char *fofc, int *nfofc) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if (!p->is_ideal || nv != 5) return AKMI_FAIL;
  int il = g.is-1, iu = g.ie+1, jl = g.js, ju = g.je, kl = g.ks, ku = g.ke;
  if (g.multi_d) { jl = g.js-1; ju = g.je+1; }
  if (g.three_d) { kl = g.ks-1; ku = g.ke+1; }
  const double gm1 = p->gamma - 1.0;
  const double efloor = p->pfloor/gm1;
  int nflag = 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          double dtodx1 = beta_dt/p->dx[3*m];
          double dtodx2 = beta_dt/p->dx[3*m+1];
          double dtodx3 = beta_dt/p->dx[3*m+2];
          double ut[5];
          for (int n = 0; n < 5; ++n) {
            double divf = dtodx1*(flx1[ix5(nv,N3,N2,N1+1,m,n,k,j,i+1)] - flx1[ix5(nv,N3,N2,N1+1,m,n,k,j,i)]);
            if (g.multi_d)
              divf += dtodx2*(flx2[ix5(nv,N3,N2+1,N1,m,n,k,j+1,i)] - flx2[ix5(nv,N3,N2+1,N1,m,n,k,j,i)]);
            if (g.three_d)
              divf += dtodx3*(flx3[ix5(nv,N3+1,N2,N1,m,n,k+1,j,i)] - flx3[ix5(nv,N3+1,N2,N1,m,n,k,j,i)]);
            size_t c = ix5(nv,N3,N2,N1,m,n,k,j,i);
            ut[n] = gam0*u0[c] + gam1*u1[c] - divf;
          }
          double b1old = 0.5*(b1x1f[ix4(N3,N2,N1+1,m,k,j,i)] + b1x1f[ix4(N3,N2,N1+1,m,k,j,i+1)]);
          double b2old = 0.5*(b1x2f[ix4(N3,N2+1,N1,m,k,j,i)] + b1x2f[ix4(N3,N2+1,N1,m,k,j+1,i)]);
          double b3old = 0.5*(b1x3f[ix4(N3+1,N2,N1,m,k,j,i)] + b1x3f[ix4(N3+1,N2,N1,m,k+1,j,i)]);
          double bx = gam0*bcc0[ix5(3,N3,N2,N1,m,0,k,j,i)] + gam1*b1old;
          double by = gam0*bcc0[ix5(3,N3,N2,N1,m,1,k,j,i)] + gam1*b2old;
          double bz = gam0*bcc0[ix5(3,N3,N2,N1,m,2,k,j,i)] + gam1*b3old;
          by += dtodx1*(e3x1[ix4(N3,N2,N1,m,k,j,i+1)] - e3x1[ix4(N3,N2,N1,m,k,j,i)]);
          bz -= dtodx1*(e2x1[ix4(N3,N2,N1,m,k,j,i+1)] - e2x1[ix4(N3,N2,N1,m,k,j,i)]);
          if (g.multi_d) {
            bx -= dtodx2*(e3x2[ix4(N3,N2,N1,m,k,j+1,i)] - e3x2[ix4(N3,N2,N1,m,k,j,i)]);
            bz += dtodx2*(e1x2[ix4(N3,N2,N1,m,k,j+1,i)] - e1x2[ix4(N3,N2,N1,m,k,j,i)]);
          }
          if (g.three_d) {
            bx += dtodx3*(e2x3[ix4(N3,N2,N1,m,k+1,j,i)] - e2x3[ix4(N3,N2,N1,m,k,j,i)]);
            by -= dtodx3*(e1x3[ix4(N3,N2,N1,m,k+1,j,i)] - e1x3[ix4(N3,N2,N1,m,k,j,i)]);
          }
          /* SingleC2P_IdealMHD, src/eos/ideal_c2p_mhd.hpp:20-67 */
          int fl = 0;
          double b2 = SQR(bx) + SQR(by) + SQR(bz);
          double dfloor_ = fmax(p->dfloor, b2/p->sigma_max);
          double ud = ut[0];
          if (ud < dfloor_) { ud = dfloor_; fl = 1; }
          double di = 1.0/ud;
          double e_k = 0.5*di*(SQR(ut[1]) + SQR(ut[2]) + SQR(ut[3]));
          double e_m = 0.5*(SQR(bx) + SQR(by) + SQR(bz));
          double we = (ut[4] - e_k - e_m);
          if (we < efloor) { we = efloor; fl = 1; }
          if (gm1*we*di < p->tfloor) { we = ud*p->tfloor/gm1; fl = 1; }
          double spe_over_eps = gm1/pow(ud, gm1);
          double spe = spe_over_eps*we*di;
          if (spe <= p->sfloor) fl = 1;
          if (fl) { fofc[ix4(N3,N2,N1,m,k,j,i)] = 1; nflag++; }
        }
  const double *bf[3] = {b0x1f, b0x2f, b0x3f};
  double *fl3[3] = {flx1, flx2, flx3};
  double *eyv[3] = {e3x1, e1x2, e2x3}, *ezv[3] = {e2x1, e3x2, e1x3};
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          if (!fofc[ix4(N3,N2,N1,m,k,j,i)]) continue;
          for (int dir = 0; dir < 3; ++dir) {
            if (dir == 1 && !g.multi_d) continue;
            if (dir == 2 && !g.three_d) continue;
            const int ivx = IVX + dir, ivy = IVX + (dir + 1)%3, ivz = IVX + (dir + 2)%3;
            const int iby = (dir + 1)%3, ibz = (dir + 2)%3;
            const int d1 = dir == 0, d2 = dir == 1, d3 = dir == 2;
            const int f1 = N1 + d1, f2 = N2 + d2, f3 = N3 + d3;
            for (int side = 0; side < 2; ++side) {
              const int kf = k + side*d3, jf = j + side*d2, ifc = i + side*d1;
              double a[7], b[7], f[7];
              const int comp[5] = {IDN, ivx, ivy, ivz, IEN};
              for (int q = 0; q < 5; ++q) {
                a[q] = w0[ix5(nv,N3,N2,N1,m,comp[q],kf-d3,jf-d2,ifc-d1)];
                b[q] = w0[ix5(nv,N3,N2,N1,m,comp[q],kf,jf,ifc)];
              }
              a[5] = bcc0[ix5(3,N3,N2,N1,m,iby,kf-d3,jf-d2,ifc-d1)]; b[5] = bcc0[ix5(3,N3,N2,N1,m,iby,kf,jf,ifc)];
              a[6] = bcc0[ix5(3,N3,N2,N1,m,ibz,kf-d3,jf-d2,ifc-d1)]; b[6] = bcc0[ix5(3,N3,N2,N1,m,ibz,kf,jf,ifc)];
              akref_llf_mhd(p->gamma, a, b, bf[dir][ix4(f3,f2,f1,m,kf,jf,ifc)], f);
              for (int q = 0; q < 5; ++q) fl3[dir][ix5(nv,f3,f2,f1,m,comp[q],kf,jf,ifc)] = f[q];
              eyv[dir][ix4(N3,N2,N1,m,kf,jf,ifc)] = -f[5];     /* flux.by = -0.5*(...) (:83) */
              ezv[dir][ix4(N3,N2,N1,m,kf,jf,ifc)] = f[6];
            }
          }
        }
  memset(fofc, 0, (size_t)g.nmb*N3*N2*N1);                   /* deep_copy(fofc, false), :487-489 */
  if (nfofc) *nfofc += nflag;
  return 0;
}

/* ---- diffusion hooks of the task chain (src/hydro/hydro_tasks.cpp:183-189, src/mhd/mhd_tasks.cpp:198-206,381-383) ---- */

/* Viscosity::AddViscousFluxIso, src/diffusion/viscosity.cpp:64-229 (constant isotropic nu) */
int akref_viscous_fluxes(const akmi_pack *p, double nu_iso, const double *w0, double *flx1,
                         double *flx2, double *flx3, int fs) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int ideal = p->is_ideal;
#define W(n,k,j,i) w0[ix5(nv,N3,N2,N1,m,(n),(k),(j),(i))]
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie+1; ++i) {
        double fvx = 4.0*(W(IVX,k,j,i) - W(IVX,k,j,i-1))/(3.0*dx1);
        double fvy =     (W(IVY,k,j,i) - W(IVY,k,j,i-1))/dx1;
        double fvz =     (W(IVZ,k,j,i) - W(IVZ,k,j,i-1))/dx1;
        if (g.multi_d) {
          fvx -= ((W(IVY,k,j+1,i) + W(IVY,k,j+1,i-1)) - (W(IVY,k,j-1,i) + W(IVY,k,j-1,i-1)))/(6.0*dx2);
          fvy += ((W(IVX,k,j+1,i) + W(IVX,k,j+1,i-1)) - (W(IVX,k,j-1,i) + W(IVX,k,j-1,i-1)))/(4.0*dx2);
        }
        if (g.three_d) {
          fvx -= ((W(IVZ,k+1,j,i) + W(IVZ,k+1,j,i-1)) - (W(IVZ,k-1,j,i) + W(IVZ,k-1,j,i-1)))/(6.0*dx3);
          fvz += ((W(IVX,k+1,j,i) + W(IVX,k+1,j,i-1)) - (W(IVX,k-1,j,i) + W(IVX,k-1,j,i-1)))/(4.0*dx3);
        }
        double nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k,j,i-1));
        flx1[ix5(nv,N3,N2,N1+fs,m,IVX,k,j,i)] -= nud*fvx;
        flx1[ix5(nv,N3,N2,N1+fs,m,IVY,k,j,i)] -= nud*fvy;
        flx1[ix5(nv,N3,N2,N1+fs,m,IVZ,k,j,i)] -= nud*fvz;
        if (ideal)
          flx1[ix5(nv,N3,N2,N1+fs,m,IEN,k,j,i)] -= 0.5*nud*((W(IVX,k,j,i-1) + W(IVX,k,j,i))*fvx +
                                                           (W(IVY,k,j,i-1) + W(IVY,k,j,i))*fvy +
                                                           (W(IVZ,k,j,i-1) + W(IVZ,k,j,i))*fvz);
      }
    if (!g.multi_d) continue;
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je+1; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double fvx = (W(IVX,k,j,i) - W(IVX,k,j-1,i))/dx2 +
                     ((W(IVY,k,j,i+1) + W(IVY,k,j-1,i+1)) - (W(IVY,k,j,i-1) + W(IVY,k,j-1,i-1)))/(4.0*dx1);
        double fvy = (W(IVY,k,j,i) - W(IVY,k,j-1,i))*4.0/(3.0*dx2) -
                     ((W(IVX,k,j,i+1) + W(IVX,k,j-1,i+1)) - (W(IVX,k,j,i-1) + W(IVX,k,j-1,i-1)))/(6.0*dx1);
        double fvz = (W(IVZ,k,j,i) - W(IVZ,k,j-1,i))/dx2;
        if (g.three_d) {
          fvy -= ((W(IVZ,k+1,j,i) + W(IVZ,k+1,j-1,i)) - (W(IVZ,k-1,j,i) + W(IVZ,k-1,j-1,i)))/(6.0*dx3);
          fvz += ((W(IVY,k+1,j,i) + W(IVY,k+1,j-1,i)) - (W(IVY,k-1,j,i) + W(IVY,k-1,j-1,i)))/(4.0*dx3);
        }
        double nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k,j-1,i));
        flx2[ix5(nv,N3,N2+fs,N1,m,IVX,k,j,i)] -= nud*fvx;
        flx2[ix5(nv,N3,N2+fs,N1,m,IVY,k,j,i)] -= nud*fvy;
        flx2[ix5(nv,N3,N2+fs,N1,m,IVZ,k,j,i)] -= nud*fvz;
        if (ideal)
          flx2[ix5(nv,N3,N2+fs,N1,m,IEN,k,j,i)] -= 0.5*nud*((W(IVX,k,j-1,i) + W(IVX,k,j,i))*fvx +
                                                           (W(IVY,k,j-1,i) + W(IVY,k,j,i))*fvy +
                                                           (W(IVZ,k,j-1,i) + W(IVZ,k,j,i))*fvz);
      }
    if (!g.three_d) continue;
    for (int k = g.ks; k <= g.ke+1; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double fvx = (W(IVX,k,j,i) - W(IVX,k-1,j,i))/dx3 +
                     ((W(IVZ,k,j,i+1) + W(IVZ,k-1,j,i+1)) - (W(IVZ,k,j,i-1) + W(IVZ,k-1,j,i-1)))/(4.0*dx1);
        double fvy = (W(IVY,k,j,i) - W(IVY,k-1,j,i))/dx3 +
                     ((W(IVZ,k,j+1,i) + W(IVZ,k-1,j+1,i)) - (W(IVZ,k,j-1,i) + W(IVZ,k-1,j-1,i)))/(4.0*dx2);
        double fvz = (W(IVZ,k,j,i) - W(IVZ,k-1,j,i))*4.0/(3.0*dx3) -
                     ((W(IVX,k,j,i+1) + W(IVX,k-1,j,i+1)) - (W(IVX,k,j,i-1) + W(IVX,k-1,j,i-1)))/(6.0*dx1) -
                     ((W(IVY,k,j+1,i) + W(IVY,k-1,j+1,i)) - (W(IVY,k,j-1,i) + W(IVY,k-1,j-1,i)))/(6.0*dx2);
        double nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k-1,j,i));
        flx3[ix5(nv,N3+fs,N2,N1,m,IVX,k,j,i)] -= nud*fvx;
        flx3[ix5(nv,N3+fs,N2,N1,m,IVY,k,j,i)] -= nud*fvy;
        flx3[ix5(nv,N3+fs,N2,N1,m,IVZ,k,j,i)] -= nud*fvz;
        if (ideal)
          flx3[ix5(nv,N3+fs,N2,N1,m,IEN,k,j,i)] -= 0.5*nud*((W(IVX,k-1,j,i) + W(IVX,k,j,i))*fvx +
                                                           (W(IVY,k-1,j,i) + W(IVY,k,j,i))*fvy +
                                                           (W(IVZ,k-1,j,i) + W(IVZ,k,j,i))*fvz);
      }
  }
  return 0;
}

/* Conduction::AddHeatFluxIso, src/diffusion/conduction.cpp:106-152 (constant diffusivity) */
int akref_heat_fluxes(const akmi_pack *p, double alpha_iso, const double *w0, double *flx1,
                      double *flx2, double *flx3, int fs) {
  G g = mkG(p);
  if (!p->is_ideal) return AKMI_FAIL;
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const double gm1 = p->gamma - 1.0;
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie+1; ++i) {
        double tempr = W(IEN,k,j,i)/W(IDN,k,j,i);
        double templ = W(IEN,k,j,i-1)/W(IDN,k,j,i-1);
        double dtempdx = (tempr - templ) * gm1 / dx1;
        double densf = 0.5*(W(IDN,k,j,i) + W(IDN,k,j,i-1));
        flx1[ix5(nv,N3,N2,N1+fs,m,IEN,k,j,i)] -= alpha_iso * densf * dtempdx;
      }
    if (!g.multi_d) continue;
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je+1; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double tempr = W(IEN,k,j,i)/W(IDN,k,j,i);
        double templ = W(IEN,k,j-1,i)/W(IDN,k,j-1,i);
        double dtempdx = (tempr - templ) * gm1 / dx2;
        double densf = 0.5*(W(IDN,k,j,i) + W(IDN,k,j-1,i));
        flx2[ix5(nv,N3,N2+fs,N1,m,IEN,k,j,i)] -= alpha_iso * densf * dtempdx;
      }
    if (!g.three_d) continue;
    for (int k = g.ks; k <= g.ke+1; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double tempr = W(IEN,k,j,i)/W(IDN,k,j,i);
        double templ = W(IEN,k-1,j,i)/W(IDN,k-1,j,i);
        double dtempdx = (tempr - templ) * gm1 / dx3;
        double densf = 0.5*(W(IDN,k,j,i) + W(IDN,k-1,j,i));
        flx3[ix5(nv,N3+fs,N2,N1,m,IEN,k,j,i)] -= alpha_iso * densf * dtempdx;
      }
  }
  return 0;
}

/* Conduction::NewTimeStep, src/diffusion/conduction.cpp:314-377: the cell reduction (before *fac) */
int akref_conduction_newdt(const akmi_pack *p, double alpha_iso, const double *w0, double *dtmin) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const double gm1 = p->gamma - 1.0;
  double min_dt = (double)FLT_MAX;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        min_dt = fmin(min_dt, SQR(p->dx[3*m])/alpha_iso*W(IDN,k,j,i)/gm1);
        if (g.multi_d) min_dt = fmin(min_dt, SQR(p->dx[3*m+1])/alpha_iso*W(IDN,k,j,i)/gm1);
        if (g.three_d) min_dt = fmin(min_dt, SQR(p->dx[3*m+2])/alpha_iso*W(IDN,k,j,i)/gm1);
      }
  *dtmin = min_dt;
  return 0;
}
#undef W

/* Resistivity::AddEMFConstantResist, src/diffusion/resistivity.cpp:78-177 + CurrentDensity
 * (src/diffusion/current_density.hpp:30-57): E += eta_ohm * J on the cell edges */
int akref_resistive_emfs(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                         const double *bx3f, double *e1, double *e2, double *e3) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
#define B1(k,j,i) bx1f[ix4(N3,N2,N1+1,m,(k),(j),(i))]
#define B2(k,j,i) bx2f[ix4(N3,N2+1,N1,m,(k),(j),(i))]
#define B3(k,j,i) bx3f[ix4(N3+1,N2,N1,m,(k),(j),(i))]
#define E1(k,j,i) e1[ix4(N3+1,N2+1,N1,m,(k),(j),(i))]
#define E2(k,j,i) e2[ix4(N3+1,N2,N1+1,m,(k),(j),(i))]
#define E3(k,j,i) e3[ix4(N3,N2+1,N1+1,m,(k),(j),(i))]
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    const int kl = g.ks, ku = g.three_d ? g.ke+1 : g.ks;
    const int jl = g.js, ju = g.multi_d ? g.je+1 : g.js;
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j)
      for (int i = g.is; i <= g.ie+1; ++i) {
        double j1 = 0.0;
        double j2 = -(B3(k,j,i) - B3(k,j,i-1))/dx1;
        double j3 =  (B2(k,j,i) - B2(k,j,i-1))/dx1;
        if (g.multi_d) {
          j1 += (B3(k,j,i) - B3(k,j-1,i))/dx2;
          j3 -= (B1(k,j,i) - B1(k,j-1,i))/dx2;
        }
        if (g.three_d) {
          j1 -= (B2(k,j,i) - B2(k-1,j,i))/dx3;
          j2 += (B1(k,j,i) - B1(k-1,j,i))/dx3;
        }
        if (g.three_d) {
          E1(k,j,i) += eta_ohm*j1; E2(k,j,i) += eta_ohm*j2; E3(k,j,i) += eta_ohm*j3;
        } else if (g.multi_d) {                    /* :124-149 */
          E1(g.ks,j,i) += eta_ohm*j1; E1(g.ke+1,j,i) += eta_ohm*j1;
          E2(g.ks,j,i) += eta_ohm*j2; E2(g.ke+1,j,i) += eta_ohm*j2;
          E3(g.ks,j,i) += eta_ohm*j3;
        } else {                                   /* :93-121 */
          E2(g.ks,g.js,i) += eta_ohm*j2; E2(g.ke+1,g.js,i) += eta_ohm*j2;
          E3(g.ks,g.js,i) += eta_ohm*j3; E3(g.ks,g.je+1,i) += eta_ohm*j3;
        }
      }
  }
  return 0;
}

/* Resistivity::AddFluxConstantResist, src/diffusion/resistivity.cpp:185-272: Poynting flux of the
 * resistive field added to the (face-shaped) energy flux */
int akref_resistive_fluxes(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                           const double *bx3f, double *flx1, double *flx2, double *flx3) {
  G g = mkG(p);
  if (!p->is_ideal) return AKMI_FAIL;
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const double qa = 0.25*eta_ohm;
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie+1; ++i) {
        double j2k   = -(B3(k,j,i) - B3(k,j,i-1))/dx1;
        double j2kp1 = -(B3(k+1,j,i) - B3(k+1,j,i-1))/dx1;
        double j3j   = (B2(k,j,i) - B2(k,j,i-1))/dx1;
        double j3jp1 = (B2(k,j+1,i) - B2(k,j+1,i-1))/dx1;
        if (g.multi_d) {
          j3j   -= (B1(k,j,i) - B1(k,j-1,i))/dx2;
          j3jp1 -= (B1(k,j+1,i) - B1(k,j,i))/dx2;
        }
        if (g.three_d) {
          j2k   += (B1(k,j,i) - B1(k-1,j,i))/dx3;
          j2kp1 += (B1(k+1,j,i) - B1(k,j,i))/dx3;
        }
        flx1[ix5(nv,N3,N2,N1+1,m,IEN,k,j,i)] += qa*(j2k  *(B3(k,j,i) + B3(k,j,i-1)) +
                                                   j2kp1*(B3(k+1,j,i) + B3(k+1,j,i-1)) -
                                                   j3j  *(B2(k,j,i) + B2(k,j,i-1)) -
                                                   j3jp1*(B2(k,j+1,i) + B2(k,j+1,i-1)));
      }
    if (!g.multi_d) continue;
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je+1; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double j1k   = (B3(k,j,i) - B3(k,j-1,i))/dx2;
        double j1kp1 = (B3(k+1,j,i) - B3(k+1,j-1,i))/dx2;
        double j3i   = (B2(k,j,i) - B2(k,j,i-1))/dx1 - (B1(k,j,i) - B1(k,j-1,i))/dx2;
        double j3ip1 = (B2(k,j,i+1) - B2(k,j,i))/dx1 - (B1(k,j,i+1) - B1(k,j-1,i+1))/dx2;
        if (g.three_d) {
          j1k   -= (B2(k,j,i) - B2(k-1,j,i))/dx3;
          j1kp1 -= (B2(k+1,j,i) - B2(k,j,i))/dx3;
        }
        flx2[ix5(nv,N3,N2+1,N1,m,IEN,k,j,i)] += qa*(j3i  *(B1(k,j,i) + B1(k,j-1,i)) +
                                                   j3ip1*(B1(k,j,i+1) + B1(k,j-1,i+1)) -
                                                   j1k  *(B3(k,j,i) + B3(k,j-1,i)) -
                                                   j1kp1*(B3(k+1,j,i) + B3(k+1,j-1,i)));
      }
    if (!g.three_d) continue;
    for (int k = g.ks; k <= g.ke+1; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double j1j   = (B3(k,j,i) - B3(k,j-1,i))/dx2 - (B2(k,j,i) - B2(k-1,j,i))/dx3;
        double j1jp1 = (B3(k,j+1,i) - B3(k,j,i))/dx2 - (B2(k,j+1,i) - B2(k-1,j+1,i))/dx3;
        double j2i   = -(B3(k,j,i) - B3(k,j,i-1))/dx1 + (B1(k,j,i) - B1(k-1,j,i))/dx3;
        double j2ip1 = -(B3(k,j,i+1) - B3(k,j,i))/dx1 + (B1(k,j,i+1) - B1(k-1,j,i+1))/dx3;
        flx3[ix5(nv,N3+1,N2,N1,m,IEN,k,j,i)] += qa*(j1j  *(B2(k,j,i) + B2(k-1,j,i)) +
                                                   j1jp1*(B2(k,j+1,i) + B2(k-1,j+1,i)) -
                                                   j2i  *(B1(k,j,i) + B1(k-1,j,i)) -
                                                   j2ip1*(B1(k,j,i+1) + B1(k-1,j,i+1)));
      }
  }
  return 0;
}
#undef B1
#undef B2
#undef B3
#undef E1
#undef E2
#undef E3

/* History sums, src/outputs/history.cpp:78-160 (hydro), 272-374 (MHD); sequential (m,k,j,i) */
int akref_history_sums(const akmi_pack *p, int is_mhd, const double *u0, const double *bx1f,
                       const double *bx2f, const double *bx3f, double *out) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int ideal = p->is_ideal, o = ideal ? 5 : 4;       /* history.cpp: KE starts at nhydro */
  const int nh = (is_mhd ? 11 : 8) - (ideal ? 0 : 1);
  for (int n = 0; n < nh; ++n) out[n] = 0.0;
  for (int m = 0; m < g.nmb; ++m) {
    const double vol = p->dx[3*m]*p->dx[3*m+1]*p->dx[3*m+2];
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          double d = u0[ix5(nv,N3,N2,N1,m,IDN,k,j,i)];
          double m1 = u0[ix5(nv,N3,N2,N1,m,1,k,j,i)], m2 = u0[ix5(nv,N3,N2,N1,m,2,k,j,i)];
          double m3 = u0[ix5(nv,N3,N2,N1,m,3,k,j,i)];
          out[0] += vol*d; out[1] += vol*m1; out[2] += vol*m2; out[3] += vol*m3;
          if (ideal) out[4] += vol*u0[ix5(nv,N3,N2,N1,m,IEN,k,j,i)];
          out[o] += vol*0.5*SQR(m1)/d;
          out[o+1] += vol*0.5*SQR(m2)/d;
          out[o+2] += vol*0.5*SQR(m3)/d;
          if (is_mhd) {
            out[o+3] += vol*0.25*(SQR(bx1f[ix4(N3,N2,N1+1,m,k,j,i+1)]) + SQR(bx1f[ix4(N3,N2,N1+1,m,k,j,i)]));
            out[o+4] += vol*0.25*(SQR(bx2f[ix4(N3,N2+1,N1,m,k,j+1,i)]) + SQR(bx2f[ix4(N3,N2+1,N1,m,k,j,i)]));
            out[o+5] += vol*0.25*(SQR(bx3f[ix4(N3+1,N2,N1,m,k+1,j,i)]) + SQR(bx3f[ix4(N3+1,N2,N1,m,k,j,i)]));
          }
        }
  }
  return 0;
}

/* MHD::CornerE, src/mhd/mhd_corner_e.cpp:26-417 (Newtonian branches: 1D :39-53,
 * 2D :58-66,139-192, 3D :303-414) */
int akref_mhd_corner_e(const akmi_pack *p, const double *w0, const double *bcc0,
                       const double *e3x1, const double *e2x1, const double *e1x2,
                       const double *e3x2, const double *e2x3, const double *e1x3,
                       const double *flx1, const double *flx2, const double *flx3,
                       double *e1, double *e2, double *e3) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int is = g.is, ie = g.ie, js = g.js, je = g.je, ks = g.ks, ke = g.ke;
#define CC(a,m,k,j,i)  a[ix4(N3,N2,N1,m,k,j,i)]
#define E1(m,k,j,i) e1[ix4(N3+1,N2+1,N1,m,k,j,i)]
#define E2(m,k,j,i) e2[ix4(N3+1,N2,N1+1,m,k,j,i)]
#define E3(m,k,j,i) e3[ix4(N3,N2+1,N1+1,m,k,j,i)]
#define F1D(m,k,j,i) flx1[ix5(nv,N3,N2,N1+1,m,IDN,k,j,i)]
#define F2D(m,k,j,i) flx2[ix5(nv,N3,N2+1,N1,m,IDN,k,j,i)]
#define F3D(m,k,j,i) flx3[ix5(nv,N3+1,N2,N1,m,IDN,k,j,i)]
#define W(n,m,k,j,i) w0[ix5(nv,N3,N2,N1,m,n,k,j,i)]
#define B(n,m,k,j,i) bcc0[ix5(3,N3,N2,N1,m,n,k,j,i)]
  if (!g.multi_d) {
    for (int m = 0; m < g.nmb; ++m)
      for (int i = is; i <= ie+1; ++i) {
        E2(m,ks,js,i) = CC(e2x1,m,ks,js,i);
        E2(m,ke+1,js,i) = CC(e2x1,m,ks,js,i);
        E3(m,ks,js,i) = CC(e3x1,m,ks,js,i);
        E3(m,ks,je+1,i) = CC(e3x1,m,ks,js,i);
      }
    return 0;
  }
  size_t ncell = (size_t)g.nmb*N3*N2*N1;
  double *e1cc = ws_get(4, ncell), *e2cc = ws_get(5, ncell), *e3cc = ws_get(6, ncell);
  if (!g.three_d) {
    for (int m = 0; m < g.nmb; ++m)
      for (int j = js-1; j <= je+1; ++j)
        for (int i = is-1; i <= ie+1; ++i)
          CC(e3cc,m,ks,j,i) = W(IVY,m,ks,j,i)*B(IBX,m,ks,j,i) - W(IVX,m,ks,j,i)*B(IBY,m,ks,j,i);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < g.nmb; ++m)
      for (int j = js; j <= je+1; ++j)
        for (int i = is; i <= ie+1; ++i) {
          E2(m,ks,j,i) = CC(e2x1,m,ks,j,i);
          E2(m,ke+1,j,i) = CC(e2x1,m,ks,j,i);
          E1(m,ks,j,i) = CC(e1x2,m,ks,j,i);
          E1(m,ke+1,j,i) = CC(e1x2,m,ks,j,i);
          double e3_l2, e3_r2, e3_l1, e3_r1;
          if (F1D(m,ks,j-1,i) >= 0.0) e3_l2 = CC(e3x2,m,ks,j,i-1) - CC(e3cc,m,ks,j-1,i-1);
          else                        e3_l2 = CC(e3x2,m,ks,j,i  ) - CC(e3cc,m,ks,j-1,i  );
          if (F1D(m,ks,j,i) >= 0.0)   e3_r2 = CC(e3x2,m,ks,j,i-1) - CC(e3cc,m,ks,j  ,i-1);
          else                        e3_r2 = CC(e3x2,m,ks,j,i  ) - CC(e3cc,m,ks,j  ,i  );
          if (F2D(m,ks,j,i-1) >= 0.0) e3_l1 = CC(e3x1,m,ks,j-1,i) - CC(e3cc,m,ks,j-1,i-1);
          else                        e3_l1 = CC(e3x1,m,ks,j  ,i) - CC(e3cc,m,ks,j  ,i-1);
          if (F2D(m,ks,j,i) >= 0.0)   e3_r1 = CC(e3x1,m,ks,j-1,i) - CC(e3cc,m,ks,j-1,i  );
          else                        e3_r1 = CC(e3x1,m,ks,j  ,i) - CC(e3cc,m,ks,j  ,i  );
          E3(m,ks,j,i) = 0.25*(e3_l1 + e3_r1 + e3_l2 + e3_r2 +
              CC(e3x2,m,ks,j,i-1) + CC(e3x2,m,ks,j,i) + CC(e3x1,m,ks,j-1,i) + CC(e3x1,m,ks,j,i));
        }
    return 0;
  }
  /* 3D: e_cc_3d (src/mhd/mhd_corner_e.cpp:309-317) */
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = ks-1; k <= ke+1; ++k)
      for (int j = js-1; j <= je+1; ++j)
        for (int i = is-1; i <= ie+1; ++i) {
          CC(e1cc,m,k,j,i) = W(IVZ,m,k,j,i)*B(IBY,m,k,j,i) - W(IVY,m,k,j,i)*B(IBZ,m,k,j,i);
          CC(e2cc,m,k,j,i) = W(IVX,m,k,j,i)*B(IBZ,m,k,j,i) - W(IVZ,m,k,j,i)*B(IBX,m,k,j,i);
          CC(e3cc,m,k,j,i) = W(IVY,m,k,j,i)*B(IBX,m,k,j,i) - W(IVX,m,k,j,i)*B(IBY,m,k,j,i);
        }
  /* emf3 (src/mhd/mhd_corner_e.cpp:338-414) */
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = ks; k <= ke+1; ++k)
      for (int j = js; j <= je+1; ++j)
        for (int i = is; i <= ie+1; ++i) {
          double e1_l3, e1_r3, e1_l2, e1_r2;
          if (F2D(m,k-1,j,i) >= 0.0) e1_l3 = CC(e1x3,m,k,j-1,i) - CC(e1cc,m,k-1,j-1,i);
          else                       e1_l3 = CC(e1x3,m,k,j  ,i) - CC(e1cc,m,k-1,j  ,i);
          if (F2D(m,k,j,i) >= 0.0)   e1_r3 = CC(e1x3,m,k,j-1,i) - CC(e1cc,m,k  ,j-1,i);
          else                       e1_r3 = CC(e1x3,m,k,j  ,i) - CC(e1cc,m,k  ,j  ,i);
          if (F3D(m,k,j-1,i) >= 0.0) e1_l2 = CC(e1x2,m,k-1,j,i) - CC(e1cc,m,k-1,j-1,i);
          else                       e1_l2 = CC(e1x2,m,k  ,j,i) - CC(e1cc,m,k  ,j-1,i);
          if (F3D(m,k,j,i) >= 0.0)   e1_r2 = CC(e1x2,m,k-1,j,i) - CC(e1cc,m,k-1,j  ,i);
          else                       e1_r2 = CC(e1x2,m,k  ,j,i) - CC(e1cc,m,k  ,j  ,i);
          /* the reference writes all three components over the full (k,j,i) range */
          E1(m,k,j,i) = 0.25*(e1_l3 + e1_r3 + e1_l2 + e1_r2 +
              CC(e1x2,m,k-1,j,i) + CC(e1x2,m,k,j,i) + CC(e1x3,m,k,j-1,i) + CC(e1x3,m,k,j,i));

          double e2_l3, e2_r3, e2_l1, e2_r1;
          if (F1D(m,k-1,j,i) >= 0.0) e2_l3 = CC(e2x3,m,k,j,i-1) - CC(e2cc,m,k-1,j,i-1);
          else                       e2_l3 = CC(e2x3,m,k,j,i  ) - CC(e2cc,m,k-1,j,i  );
          if (F1D(m,k,j,i) >= 0.0)   e2_r3 = CC(e2x3,m,k,j,i-1) - CC(e2cc,m,k  ,j,i-1);
          else                       e2_r3 = CC(e2x3,m,k,j,i  ) - CC(e2cc,m,k  ,j,i  );
          if (F3D(m,k,j,i-1) >= 0.0) e2_l1 = CC(e2x1,m,k-1,j,i) - CC(e2cc,m,k-1,j,i-1);
          else                       e2_l1 = CC(e2x1,m,k  ,j,i) - CC(e2cc,m,k  ,j,i-1);
          if (F3D(m,k,j,i) >= 0.0)   e2_r1 = CC(e2x1,m,k-1,j,i) - CC(e2cc,m,k-1,j,i  );
          else                       e2_r1 = CC(e2x1,m,k  ,j,i) - CC(e2cc,m,k  ,j,i  );
          E2(m,k,j,i) = 0.25*(e2_l3 + e2_r3 + e2_l1 + e2_r1 +
              CC(e2x3,m,k,j,i-1) + CC(e2x3,m,k,j,i) + CC(e2x1,m,k-1,j,i) + CC(e2x1,m,k,j,i));

          double e3_l2, e3_r2, e3_l1, e3_r1;
          if (F1D(m,k,j-1,i) >= 0.0) e3_l2 = CC(e3x2,m,k,j,i-1) - CC(e3cc,m,k,j-1,i-1);
          else                       e3_l2 = CC(e3x2,m,k,j,i  ) - CC(e3cc,m,k,j-1,i  );
          if (F1D(m,k,j,i) >= 0.0)   e3_r2 = CC(e3x2,m,k,j,i-1) - CC(e3cc,m,k,j  ,i-1);
          else                       e3_r2 = CC(e3x2,m,k,j,i  ) - CC(e3cc,m,k,j  ,i  );
          if (F2D(m,k,j,i-1) >= 0.0) e3_l1 = CC(e3x1,m,k,j-1,i) - CC(e3cc,m,k,j-1,i-1);
          else                       e3_l1 = CC(e3x1,m,k,j  ,i) - CC(e3cc,m,k,j  ,i-1);
          if (F2D(m,k,j,i) >= 0.0)   e3_r1 = CC(e3x1,m,k,j-1,i) - CC(e3cc,m,k,j-1,i  );
          else                       e3_r1 = CC(e3x1,m,k,j  ,i) - CC(e3cc,m,k,j  ,i  );
          E3(m,k,j,i) = 0.25*(e3_l1 + e3_r1 + e3_l2 + e3_r2 +
              CC(e3x2,m,k,j,i-1) + CC(e3x2,m,k,j,i) + CC(e3x1,m,k,j-1,i) + CC(e3x1,m,k,j,i));
        }
  return 0;
}

/* MHD::CT, src/mhd/mhd_ct.cpp:23-80 */
int akref_mhd_ct(const akmi_pack *p, double gam0, double gam1, double beta_dt,
                 const double *e1, const double *e2, const double *e3, double *b0x1f,
                 double *b0x2f, double *b0x3f, const double *b1x1f, const double *b1x2f,
                 const double *b1x3f) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int is = g.is, ie = g.ie, js = g.js, je = g.je, ks = g.ks, ke = g.ke;
  if (g.multi_d) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int m = 0; m < g.nmb; ++m)
      for (int k = ks; k <= ke; ++k)
        for (int j = js; j <= je; ++j)
          for (int i = is; i <= ie+1; ++i) {
            size_t c = ix4(N3,N2,N1+1,m,k,j,i);
            b0x1f[c] = gam0*b0x1f[c] + gam1*b1x1f[c];
            b0x1f[c] -= beta_dt*(E3(m,k,j+1,i) - E3(m,k,j,i))/p->dx[3*m+1];
            if (g.three_d)
              b0x1f[c] += beta_dt*(E2(m,k+1,j,i) - E2(m,k,j,i))/p->dx[3*m+2];
          }
  }
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = ks; k <= ke; ++k)
      for (int j = js; j <= je+1; ++j)
        for (int i = is; i <= ie; ++i) {
          size_t c = ix4(N3,N2+1,N1,m,k,j,i);
          b0x2f[c] = gam0*b0x2f[c] + gam1*b1x2f[c];
          b0x2f[c] += beta_dt*(E3(m,k,j,i+1) - E3(m,k,j,i))/p->dx[3*m];
          if (g.three_d)
            b0x2f[c] -= beta_dt*(E1(m,k+1,j,i) - E1(m,k,j,i))/p->dx[3*m+2];
        }
#pragma omp parallel for collapse(3) schedule(static)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = ks; k <= ke+1; ++k)
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i) {
          size_t c = ix4(N3+1,N2,N1,m,k,j,i);
          b0x3f[c] = gam0*b0x3f[c] + gam1*b1x3f[c];
          b0x3f[c] -= beta_dt*(E2(m,k,j,i+1) - E2(m,k,j,i))/p->dx[3*m];
          if (g.multi_d)
            b0x3f[c] += beta_dt*(E1(m,k,j+1,i) - E1(m,k,j,i))/p->dx[3*m+1];
        }
  return 0;
}

/* IdealMHD::ConsToPrim, src/eos/ideal_mhd.cpp:30-134 + SingleC2P_IdealMHD,
 * src/eos/ideal_c2p_mhd.hpp:20-67 */
int akref_mhd_c2p(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                  const double *bx3f, double *w0, double *bcc0, int il, int iu, int jl,
                  int ju, int kl, int ku, int *counters) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if (!p->is_ideal) {
    /* SingleC2P_IsothermalMHD, src/eos/isothermal_mhd.cpp:32-47,68-150 */
    int sumd_ = 0;
    for (int m = 0; m < g.nmb; ++m)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j)
          for (int i = il; i <= iu; ++i) {
            size_t cd = ix5(nv,N3,N2,N1,m,IDN,k,j,i), cx = ix5(nv,N3,N2,N1,m,IVX,k,j,i);
            size_t cy = ix5(nv,N3,N2,N1,m,IVY,k,j,i), cz = ix5(nv,N3,N2,N1,m,IVZ,k,j,i);
            double ubx = 0.5*(bx1f[ix4(N3,N2,N1+1,m,k,j,i)] + bx1f[ix4(N3,N2,N1+1,m,k,j,i+1)]);
            double uby = 0.5*(bx2f[ix4(N3,N2+1,N1,m,k,j,i)] + bx2f[ix4(N3,N2+1,N1,m,k,j+1,i)]);
            double ubz = 0.5*(bx3f[ix4(N3+1,N2,N1,m,k,j,i)] + bx3f[ix4(N3+1,N2,N1,m,k+1,j,i)]);
            const double b2 = SQR(ubx) + SQR(uby) + SQR(ubz);
            const double dfloor_ = fmax(p->dfloor, b2/p->sigma_max);
            double ud = u0[cd];
            if (ud < dfloor_) { ud = dfloor_; u0[cd] = ud; sumd_++; }
            double di = 1.0/ud;
            w0[cd] = ud; w0[cx] = di*u0[cx]; w0[cy] = di*u0[cy]; w0[cz] = di*u0[cz];
            bcc0[ix5(3,N3,N2,N1,m,IBX,k,j,i)] = ubx;
            bcc0[ix5(3,N3,N2,N1,m,IBY,k,j,i)] = uby;
            bcc0[ix5(3,N3,N2,N1,m,IBZ,k,j,i)] = ubz;
            for (int n = 4; n < nv; ++n)           /* scalars, isothermal_mhd.cpp:143-146 */
              w0[ix5(nv,N3,N2,N1,m,n,k,j,i)] = u0[ix5(nv,N3,N2,N1,m,n,k,j,i)]/ud;
          }
    if (counters) counters[0] += sumd_;
    return 0;
  }
  const double gm1 = p->gamma - 1.0;
  const double efloor = p->pfloor/(p->gamma - 1.0);
  const double tfloor = p->tfloor, sfloor = p->sfloor;
  int sumd = 0, sume = 0, sumt = 0;
#pragma omp parallel for collapse(3) schedule(static) reduction(+:sumd,sume,sumt)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k)
      for (int j = jl; j <= ju; ++j)
        for (int i = il; i <= iu; ++i) {
          size_t cd = ix5(nv,N3,N2,N1,m,IDN,k,j,i), cx = ix5(nv,N3,N2,N1,m,IVX,k,j,i);
          size_t cy = ix5(nv,N3,N2,N1,m,IVY,k,j,i), cz = ix5(nv,N3,N2,N1,m,IVZ,k,j,i);
          size_t ce = ix5(nv,N3,N2,N1,m,IEN,k,j,i);
          double ud = u0[cd], umx = u0[cx], umy = u0[cy], umz = u0[cz], ue = u0[ce];
          double ubx = 0.5*(bx1f[ix4(N3,N2,N1+1,m,k,j,i)] + bx1f[ix4(N3,N2,N1+1,m,k,j,i+1)]);
          double uby = 0.5*(bx2f[ix4(N3,N2+1,N1,m,k,j,i)] + bx2f[ix4(N3,N2+1,N1,m,k,j+1,i)]);
          double ubz = 0.5*(bx3f[ix4(N3+1,N2,N1,m,k,j,i)] + bx3f[ix4(N3+1,N2,N1,m,k+1,j,i)]);
          int dfl = 0, efl = 0, tfl = 0;
          const double b2 = SQR(ubx) + SQR(uby) + SQR(ubz);
          const double dfloor_ = fmax(p->dfloor, b2/p->sigma_max);
          if (ud < dfloor_) { ud = dfloor_; dfl = 1; }
          double wd = ud;
          double di = 1.0/ud;
          double wvx = di*umx, wvy = di*umy, wvz = di*umz;
          double e_k = 0.5*di*(SQR(umx) + SQR(umy) + SQR(umz));
          double e_m = 0.5*(SQR(ubx) + SQR(uby) + SQR(ubz));
          double we = (ue - e_k - e_m);
          if (we < efloor) { we = efloor; ue = efloor + e_k + e_m; efl = 1; }
          if (gm1*we*di < tfloor) { we = wd*tfloor/gm1; ue = we + e_k + e_m; tfl = 1; }
          double spe_over_eps = gm1/pow(wd, gm1);
          double spe = spe_over_eps*we*di;
          if (spe <= sfloor) { we = wd*sfloor/spe_over_eps; efl = 1; }
          if (dfl) { u0[cd] = ud; sumd++; }
          if (efl) { u0[ce] = ue; sume++; }
          if (tfl) { u0[ce] = ue; sumt++; }
          w0[cd] = wd; w0[cx] = wvx; w0[cy] = wvy; w0[cz] = wvz; w0[ce] = we;
          bcc0[ix5(3,N3,N2,N1,m,IBX,k,j,i)] = ubx;
          bcc0[ix5(3,N3,N2,N1,m,IBY,k,j,i)] = uby;
          bcc0[ix5(3,N3,N2,N1,m,IBZ,k,j,i)] = ubz;
          for (int n = 5; n < nv; ++n) {          /* scalars, ideal_mhd.cpp:113-120 */
            size_t cn = ix5(nv,N3,N2,N1,m,n,k,j,i);
            if (u0[cn] < 0.0) u0[cn] = 0.0;
            w0[cn] = u0[cn]/ud;
          }
        }
  if (counters) { counters[0] += sumd; counters[1] += sume; counters[2] += sumt; }
  return 0;
}

/* MHD::NewTimeStep, src/mhd/mhd_newdt.cpp:31-174 (Newtonian ideal branch :123-136) */
int akref_mhd_newdt(const akmi_pack *p, const double *w0, const double *bcc0, double *dt3) {
  G g = mkG(p);
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  double dt1 = (double)FLT_MAX, dt2 = (double)FLT_MAX, dt3_ = (double)FLT_MAX;
#pragma omp parallel for collapse(3) schedule(static) reduction(min:dt1,dt2,dt3_)
  for (int m = 0; m < g.nmb; ++m)
    for (int k = g.ks; k <= g.ke; ++k)
      for (int j = g.js; j <= g.je; ++j)
        for (int i = g.is; i <= g.ie; ++i) {
          double w_d = W(IDN,m,k,j,i);
          double w_bx = B(IBX,m,k,j,i), w_by = B(IBY,m,k,j,i), w_bz = B(IBZ,m,k,j,i);
          double cf, max_dv1, max_dv2, max_dv3;
          if (p->is_ideal) {
            double pr = (p->gamma - 1.0)*W(IEN,m,k,j,i);
            cf = fast_speed(p->gamma, w_d, pr, w_bx, w_by, w_bz);
            max_dv1 = fabs(W(IVX,m,k,j,i)) + cf;
            cf = fast_speed(p->gamma, w_d, pr, w_by, w_bz, w_bx);
            max_dv2 = fabs(W(IVY,m,k,j,i)) + cf;
            cf = fast_speed(p->gamma, w_d, pr, w_bz, w_bx, w_by);
            max_dv3 = fabs(W(IVZ,m,k,j,i)) + cf;
          } else {                                  /* mhd_newdt.cpp:137-144 */
            cf = fast_speed_iso(p->iso_cs, w_d, w_bx, w_by, w_bz);
            max_dv1 = fabs(W(IVX,m,k,j,i)) + cf;
            cf = fast_speed_iso(p->iso_cs, w_d, w_by, w_bz, w_bx);
            max_dv2 = fabs(W(IVY,m,k,j,i)) + cf;
            cf = fast_speed_iso(p->iso_cs, w_d, w_bz, w_bx, w_by);
            max_dv3 = fabs(W(IVZ,m,k,j,i)) + cf;
          }
          dt1 = fmin(p->dx[3*m]/max_dv1, dt1);
          dt2 = fmin(p->dx[3*m+1]/max_dv2, dt2);
          dt3_ = fmin(p->dx[3*m+2]/max_dv3, dt3_);
        }
  dt3[0] = dt1; dt3[1] = dt2; dt3[2] = dt3_;
  return 0;
}

/* ------------------------------------------------------------------------------------
 * Same-level boundary values.  The reference packs the ng innermost active layers of the
 * sender (src/bvals/buffs_cc.cpp:36-46) and unpacks into the receiver's ghost layers
 * (:176-203); for a same-rank neighbour the pack kernel writes straight into the
 * neighbour's receive buffer (src/bvals/bvals_cc.cpp:122-135).  Net effect, restated as a
 * gather: ghost element (k,j,i) of block m in direction o=(o1,o2,o3) := element
 * (k-o3*nx3, j-o2*nx2, i-o1*nx1) of the neighbour in that direction. */
static void cell_range(int o, int s, int e, int ng, int *lo, int *hi) {
  if (o < 0) { *lo = s - ng; *hi = s - 1; }
  else if (o > 0) { *lo = e + 1; *hi = e + ng; }
  else { *lo = s; *hi = e; }
}
/* face-like index along the component's own direction: shared faces excluded
 * (src/bvals/buffs_fc.cpp:39-76): o<0 -> [s-ng,s-1], o==0 -> [s,e+1], o>0 -> [e+2,e+ng+1] */
static void face_range(int o, int s, int e, int ng, int *lo, int *hi) {
  if (o < 0) { *lo = s - ng; *hi = s - 1; }
  else if (o > 0) { *lo = e + 2; *hi = e + ng + 1; }
  else { *lo = s; *hi = e + 1; }
}
static int dir_valid(const G *g, int d, int *o1, int *o2, int *o3) {
  *o1 = d%3 - 1; *o2 = (d/3)%3 - 1; *o3 = d/9 - 1;
  if (d == 13) return 0;
  if (!g->multi_d && *o2 != 0) return 0;
  if (!g->three_d && *o3 != 0) return 0;
  return 1;
}

long long akref_bvals_cc_segsize(const akmi_pack *p, int d) {
  G g = mkG(p);
  int o1, o2, o3, il, iu, jl, ju, kl, ku;
  if (!dir_valid(&g, d, &o1, &o2, &o3)) return 0;
  cell_range(o1, g.is, g.ie, g.ng, &il, &iu);
  cell_range(o2, g.js, g.je, g.ng, &jl, &ju);
  cell_range(o3, g.ks, g.ke, g.ng, &kl, &ku);
  return (long long)(iu-il+1)*(ju-jl+1)*(ku-kl+1);
}

int akref_bvals_cc_local(const akmi_pack *p, int nvar, const int *nghbr, double *u) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  for (int m = 0; m < g.nmb; ++m)
    for (int d = 0; d < 27; ++d) {
      int o1, o2, o3, il, iu, jl, ju, kl, ku;
      if (!dir_valid(&g, d, &o1, &o2, &o3)) continue;
      int src = nghbr[m*27 + d];
      if (src < 0) continue;
      cell_range(o1, g.is, g.ie, g.ng, &il, &iu);
      cell_range(o2, g.js, g.je, g.ng, &jl, &ju);
      cell_range(o3, g.ks, g.ke, g.ng, &kl, &ku);
      for (int n = 0; n < nvar; ++n)
        for (int k = kl; k <= ku; ++k)
          for (int j = jl; j <= ju; ++j)
            for (int i = il; i <= iu; ++i)
              u[ix5(nvar,N3,N2,N1,m,n,k,j,i)] =
                  u[ix5(nvar,N3,N2,N1,src,n,k-o3*g.nx3,j-o2*g.nx2,i-o1*g.nx1)];
    }
  return 0;
}

int akref_bvals_cc_pack(const akmi_pack *p, int nvar, int nsend, const int *send_tab,
                        const long long *send_off, const double *u, double *sendbuf) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  for (int s = 0; s < nsend; ++s) {
    int m = send_tab[2*s], d = send_tab[2*s+1];
    /* receiver's ghost direction is o = -d */
    int o1, o2, o3, il, iu, jl, ju, kl, ku;
    if (!dir_valid(&g, 26 - d, &o1, &o2, &o3)) continue;
    cell_range(o1, g.is, g.ie, g.ng, &il, &iu);
    cell_range(o2, g.js, g.je, g.ng, &jl, &ju);
    cell_range(o3, g.ks, g.ke, g.ng, &kl, &ku);
    double *out = sendbuf + send_off[s];
    for (int n = 0; n < nvar; ++n)
      for (int k = kl; k <= ku; ++k)
        for (int j = jl; j <= ju; ++j)
          for (int i = il; i <= iu; ++i)
            *out++ = u[ix5(nvar,N3,N2,N1,m,n,k-o3*g.nx3,j-o2*g.nx2,i-o1*g.nx1)];
  }
  return 0;
}

int akref_bvals_cc_unpack(const akmi_pack *p, int nvar, const int *nghbr,
                          const long long *seg_off, const double *recvbuf, double *u) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  for (int m = 0; m < g.nmb; ++m)
    for (int d = 0; d < 27; ++d) {
      int o1, o2, o3, il, iu, jl, ju, kl, ku;
      if (!dir_valid(&g, d, &o1, &o2, &o3)) continue;
      int e = nghbr[m*27 + d];
      if (e > -2) continue;
      const double *in = recvbuf + seg_off[-(e + 2)];
      cell_range(o1, g.is, g.ie, g.ng, &il, &iu);
      cell_range(o2, g.js, g.je, g.ng, &jl, &ju);
      cell_range(o3, g.ks, g.ke, g.ng, &kl, &ku);
      for (int n = 0; n < nvar; ++n)
        for (int k = kl; k <= ku; ++k)
          for (int j = jl; j <= ju; ++j)
            for (int i = il; i <= iu; ++i)
              u[ix5(nvar,N3,N2,N1,m,n,k,j,i)] = *in++;
    }
  return 0;
}

/* FC component c (0:x1f 1:x2f 2:x3f) ranges for ghost direction o */
static void fc_ranges(const G *g, int c, int o1, int o2, int o3, int r[6]) {
  if (c == 0) face_range(o1, g->is, g->ie, g->ng, &r[0], &r[1]);
  else        cell_range(o1, g->is, g->ie, g->ng, &r[0], &r[1]);
  if (c == 1) face_range(o2, g->js, g->je, g->ng, &r[2], &r[3]);
  else        cell_range(o2, g->js, g->je, g->ng, &r[2], &r[3]);
  if (c == 2) face_range(o3, g->ks, g->ke, g->ng, &r[4], &r[5]);
  else        cell_range(o3, g->ks, g->ke, g->ng, &r[4], &r[5]);
  /* collapsed dimensions keep their single (or two-face) extent */
  if (!g->multi_d) { r[2] = 0; r[3] = (c == 1) ? 1 : 0; }
  if (!g->three_d) { r[4] = 0; r[5] = (c == 2) ? 1 : 0; }
}
static void fc_dims(const G *g, int c, int *n3, int *n2, int *n1) {
  *n3 = g->N3 + (c == 2); *n2 = g->N2 + (c == 1); *n1 = g->N1 + (c == 0);
}

long long akref_bvals_fc_segsize(const akmi_pack *p, int d) {
  G g = mkG(p);
  int o1, o2, o3, r[6];
  if (!dir_valid(&g, d, &o1, &o2, &o3)) return 0;
  long long tot = 0;
  for (int c = 0; c < 3; ++c) {
    fc_ranges(&g, c, o1, o2, o3, r);
    tot += (long long)(r[1]-r[0]+1)*(r[3]-r[2]+1)*(r[5]-r[4]+1);
  }
  return tot;
}

int akref_bvals_fc_local(const akmi_pack *p, const int *nghbr, double *bx1f, double *bx2f,
                         double *bx3f) {
  G g = mkG(p);
  double *b[3] = {bx1f, bx2f, bx3f};
  for (int m = 0; m < g.nmb; ++m)
    for (int d = 0; d < 27; ++d) {
      int o1, o2, o3, r[6];
      if (!dir_valid(&g, d, &o1, &o2, &o3)) continue;
      int src = nghbr[m*27 + d];
      if (src < 0) continue;
      for (int c = 0; c < 3; ++c) {
        int n3, n2, n1;
        fc_dims(&g, c, &n3, &n2, &n1);
        fc_ranges(&g, c, o1, o2, o3, r);
        for (int k = r[4]; k <= r[5]; ++k)
          for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i)
              b[c][ix4(n3,n2,n1,m,k,j,i)] =
                  b[c][ix4(n3,n2,n1,src,k-o3*(g.three_d?g.nx3:0),j-o2*(g.multi_d?g.nx2:0),i-o1*g.nx1)];
      }
    }
  return 0;
}

int akref_bvals_fc_pack(const akmi_pack *p, int nsend, const int *send_tab,
                        const long long *send_off, const double *bx1f, const double *bx2f,
                        const double *bx3f, double *sendbuf) {
  G g = mkG(p);
  const double *b[3] = {bx1f, bx2f, bx3f};
  for (int s = 0; s < nsend; ++s) {
    int m = send_tab[2*s], d = send_tab[2*s+1];
    int o1, o2, o3, r[6];
    if (!dir_valid(&g, 26 - d, &o1, &o2, &o3)) continue;
    double *out = sendbuf + send_off[s];
    for (int c = 0; c < 3; ++c) {
      int n3, n2, n1;
      fc_dims(&g, c, &n3, &n2, &n1);
      fc_ranges(&g, c, o1, o2, o3, r);
      for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
          for (int i = r[0]; i <= r[1]; ++i)
            *out++ = b[c][ix4(n3,n2,n1,m,k-o3*(g.three_d?g.nx3:0),j-o2*(g.multi_d?g.nx2:0),i-o1*g.nx1)];
    }
  }
  return 0;
}

int akref_bvals_fc_unpack(const akmi_pack *p, const int *nghbr, const long long *seg_off,
                          const double *recvbuf, double *bx1f, double *bx2f, double *bx3f) {
  G g = mkG(p);
  double *b[3] = {bx1f, bx2f, bx3f};
  for (int m = 0; m < g.nmb; ++m)
    for (int d = 0; d < 27; ++d) {
      int o1, o2, o3, r[6];
      if (!dir_valid(&g, d, &o1, &o2, &o3)) continue;
      int e = nghbr[m*27 + d];
      if (e > -2) continue;
      const double *in = recvbuf + seg_off[-(e + 2)];
      for (int c = 0; c < 3; ++c) {
        int n3, n2, n1;
        fc_dims(&g, c, &n3, &n2, &n1);
        fc_ranges(&g, c, o1, o2, o3, r);
        for (int k = r[4]; k <= r[5]; ++k)
          for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i)
              b[c][ix4(n3,n2,n1,m,k,j,i)] = *in++;
      }
    }
  return 0;
}

/* HydroBCs, src/bvals/physics/hydro_bcs.cpp:69-... (outflow, reflect); x1 over all (k,j)
 * incl. ghosts, then x2 over all (k,i), then x3 over all (j,i). */
/* u_in: [nvar][6] inflow states (MeshBoundaryValues::u_in, src/bvals/bvals.cpp:323-326), may be NULL
 * when no face is flagged inflow */
static int hydro_bcs_impl(const akmi_pack *p, int nvar, const int *bcs, const double *u_in, double *u) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3, ng = g.ng;
#define U(m,n,k,j,i) u[ix5(nvar,N3,N2,N1,m,n,k,j,i)]
  for (int m = 0; m < g.nmb; ++m)
    for (int n = 0; n < nvar; ++n)
      for (int k = 0; k < N3; ++k)
        for (int j = 0; j < N2; ++j) {
          int bi = bcs[6*m], bo = bcs[6*m+1];
          for (int i = 0; i < ng; ++i) {
            if (bi == AKMI_BC_REFLECT) U(m,n,k,j,g.is-i-1) = (n == IVX ? -1.0 : 1.0)*U(m,n,k,j,g.is+i);
            else if (bi == AKMI_BC_OUTFLOW) U(m,n,k,j,g.is-i-1) = U(m,n,k,j,g.is);
            else if (bi == AKMI_BC_INFLOW) U(m,n,k,j,g.is-i-1) = u_in[6*n + 0];
            else if (bi == AKMI_BC_DIODE) U(m,n,k,j,g.is-i-1) = (n == IVX) ? fmin(0.0, U(m,n,k,j,g.is)) : U(m,n,k,j,g.is);
            else if (bi == AKMI_BC_VACUUM) U(m,n,k,j,g.is-i-1) = 0.0;
          }
          for (int i = 0; i < ng; ++i) {
            if (bo == AKMI_BC_REFLECT) U(m,n,k,j,g.ie+i+1) = (n == IVX ? -1.0 : 1.0)*U(m,n,k,j,g.ie-i);
            else if (bo == AKMI_BC_OUTFLOW) U(m,n,k,j,g.ie+i+1) = U(m,n,k,j,g.ie);
            else if (bo == AKMI_BC_INFLOW) U(m,n,k,j,g.ie+i+1) = u_in[6*n + 1];
            else if (bo == AKMI_BC_DIODE) U(m,n,k,j,g.ie+i+1) = (n == IVX) ? fmax(0.0, U(m,n,k,j,g.ie)) : U(m,n,k,j,g.ie);
            else if (bo == AKMI_BC_VACUUM) U(m,n,k,j,g.ie+i+1) = 0.0;
          }
        }
  if (!g.multi_d) return 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int n = 0; n < nvar; ++n)
      for (int k = 0; k < N3; ++k)
        for (int i = 0; i < N1; ++i) {
          int bi = bcs[6*m+2], bo = bcs[6*m+3];
          for (int j = 0; j < ng; ++j) {
            if (bi == AKMI_BC_REFLECT) U(m,n,k,g.js-j-1,i) = (n == IVY ? -1.0 : 1.0)*U(m,n,k,g.js+j,i);
            else if (bi == AKMI_BC_OUTFLOW) U(m,n,k,g.js-j-1,i) = U(m,n,k,g.js,i);
            else if (bi == AKMI_BC_INFLOW) U(m,n,k,g.js-j-1,i) = u_in[6*n + 2];
            else if (bi == AKMI_BC_DIODE) U(m,n,k,g.js-j-1,i) = (n == IVY) ? fmin(0.0, U(m,n,k,g.js,i)) : U(m,n,k,g.js,i);
            else if (bi == AKMI_BC_VACUUM) U(m,n,k,g.js-j-1,i) = 0.0;
          }
          for (int j = 0; j < ng; ++j) {
            if (bo == AKMI_BC_REFLECT) U(m,n,k,g.je+j+1,i) = (n == IVY ? -1.0 : 1.0)*U(m,n,k,g.je-j,i);
            else if (bo == AKMI_BC_OUTFLOW) U(m,n,k,g.je+j+1,i) = U(m,n,k,g.je,i);
            else if (bo == AKMI_BC_INFLOW) U(m,n,k,g.je+j+1,i) = u_in[6*n + 3];
            else if (bo == AKMI_BC_DIODE) U(m,n,k,g.je+j+1,i) = (n == IVY) ? fmax(0.0, U(m,n,k,g.je,i)) : U(m,n,k,g.je,i);
            else if (bo == AKMI_BC_VACUUM) U(m,n,k,g.je+j+1,i) = 0.0;
          }
        }
  if (!g.three_d) return 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int n = 0; n < nvar; ++n)
      for (int j = 0; j < N2; ++j)
        for (int i = 0; i < N1; ++i) {
          int bi = bcs[6*m+4], bo = bcs[6*m+5];
          for (int k = 0; k < ng; ++k) {
            if (bi == AKMI_BC_REFLECT) U(m,n,g.ks-k-1,j,i) = (n == IVZ ? -1.0 : 1.0)*U(m,n,g.ks+k,j,i);
            else if (bi == AKMI_BC_OUTFLOW) U(m,n,g.ks-k-1,j,i) = U(m,n,g.ks,j,i);
            else if (bi == AKMI_BC_INFLOW) U(m,n,g.ks-k-1,j,i) = u_in[6*n + 4];
            else if (bi == AKMI_BC_DIODE) U(m,n,g.ks-k-1,j,i) = (n == IVZ) ? fmin(0.0, U(m,n,g.ks,j,i)) : U(m,n,g.ks,j,i);
            else if (bi == AKMI_BC_VACUUM) U(m,n,g.ks-k-1,j,i) = 0.0;
          }
          for (int k = 0; k < ng; ++k) {
            if (bo == AKMI_BC_REFLECT) U(m,n,g.ke+k+1,j,i) = (n == IVZ ? -1.0 : 1.0)*U(m,n,g.ke-k,j,i);
            else if (bo == AKMI_BC_OUTFLOW) U(m,n,g.ke+k+1,j,i) = U(m,n,g.ke,j,i);
            else if (bo == AKMI_BC_INFLOW) U(m,n,g.ke+k+1,j,i) = u_in[6*n + 5];
            else if (bo == AKMI_BC_DIODE) U(m,n,g.ke+k+1,j,i) = (n == IVZ) ? fmax(0.0, U(m,n,g.ke,j,i)) : U(m,n,g.ke,j,i);
            else if (bo == AKMI_BC_VACUUM) U(m,n,g.ke+k+1,j,i) = 0.0;
          }
        }
#undef U
  return 0;
}

int akref_hydro_bcs(const akmi_pack *p, int nvar, const int *bcs, double *u) {
  return hydro_bcs_impl(p, nvar, bcs, NULL, u);
}
int akref_hydro_bcs_inflow(const akmi_pack *p, int nvar, const int *bcs, const double *u_in, double *u) {
  return hydro_bcs_impl(p, nvar, bcs, u_in, u);
}

/* BFieldBCs, src/bvals/physics/bfield_bcs.cpp:66-... (outflow, reflect) */
/* b_in: [3][6] inflow field values (MeshBoundaryValues::b_in), may be NULL without inflow faces;
 * diode and vacuum faces treat the field like outflow (bfield_bcs.cpp:88-97) */
static int bfield_bcs_impl(const akmi_pack *p, const int *bcs, const double *b_in, double *bx1f,
                           double *bx2f, double *bx3f) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3, ng = g.ng;
  const int is = g.is, ie = g.ie, js = g.js, je = g.je, ks = g.ks, ke = g.ke;
#define B1(m,k,j,i) bx1f[ix4(N3,N2,N1+1,m,k,j,i)]
#define B2(m,k,j,i) bx2f[ix4(N3,N2+1,N1,m,k,j,i)]
#define B3(m,k,j,i) bx3f[ix4(N3+1,N2,N1,m,k,j,i)]
  for (int m = 0; m < g.nmb; ++m)
    for (int k = 0; k < N3; ++k)
      for (int j = 0; j < N2; ++j) {
        int bi = bcs[6*m], bo = bcs[6*m+1];
        for (int i = 0; i < ng; ++i) {
          if (bi == AKMI_BC_REFLECT) {
            B1(m,k,j,is-i-1) = -B1(m,k,j,is+i+1);
            B2(m,k,j,is-i-1) = B2(m,k,j,is+i);
            if (j == N2-1) B2(m,k,j+1,is-i-1) = B2(m,k,j+1,is+i);
            B3(m,k,j,is-i-1) = B3(m,k,j,is+i);
            if (k == N3-1) B3(m,k+1,j,is-i-1) = B3(m,k+1,j,is+i);
          } else if (bi == AKMI_BC_OUTFLOW || bi == AKMI_BC_DIODE || bi == AKMI_BC_VACUUM) {
            B1(m,k,j,is-i-1) = B1(m,k,j,is);
            B2(m,k,j,is-i-1) = B2(m,k,j,is);
            if (j == N2-1) B2(m,k,j+1,is-i-1) = B2(m,k,j+1,is);
            B3(m,k,j,is-i-1) = B3(m,k,j,is);
            if (k == N3-1) B3(m,k+1,j,is-i-1) = B3(m,k+1,j,is);
          } else if (bi == AKMI_BC_INFLOW) {
            B1(m,k,j,is-i-1) = b_in[6*0 + 0];
            B2(m,k,j,is-i-1) = b_in[6*1 + 0];
            if (j == N2-1) B2(m,k,j+1,is-i-1) = b_in[6*1 + 0];
            B3(m,k,j,is-i-1) = b_in[6*2 + 0];
            if (k == N3-1) B3(m,k+1,j,is-i-1) = b_in[6*2 + 0];
          }
        }
        for (int i = 0; i < ng; ++i) {
          if (bo == AKMI_BC_REFLECT) {
            B1(m,k,j,ie+i+2) = -B1(m,k,j,ie-i);
            B2(m,k,j,ie+i+1) = B2(m,k,j,ie-i);
            if (j == N2-1) B2(m,k,j+1,ie+i+1) = B2(m,k,j+1,ie-i);
            B3(m,k,j,ie+i+1) = B3(m,k,j,ie-i);
            if (k == N3-1) B3(m,k+1,j,ie+i+1) = B3(m,k+1,j,ie-i);
          } else if (bo == AKMI_BC_OUTFLOW || bo == AKMI_BC_DIODE || bo == AKMI_BC_VACUUM) {
            B1(m,k,j,ie+i+2) = B1(m,k,j,ie+1);
            B2(m,k,j,ie+i+1) = B2(m,k,j,ie);
            if (j == N2-1) B2(m,k,j+1,ie+i+1) = B2(m,k,j+1,ie);
            B3(m,k,j,ie+i+1) = B3(m,k,j,ie);
            if (k == N3-1) B3(m,k+1,j,ie+i+1) = B3(m,k+1,j,ie);
          } else if (bo == AKMI_BC_INFLOW) {
            B1(m,k,j,ie+i+2) = b_in[6*0 + 1];
            B2(m,k,j,ie+i+1) = b_in[6*1 + 1];
            if (j == N2-1) B2(m,k,j+1,ie+i+1) = b_in[6*1 + 1];
            B3(m,k,j,ie+i+1) = b_in[6*2 + 1];
            if (k == N3-1) B3(m,k+1,j,ie+i+1) = b_in[6*2 + 1];
          }
        }
      }
  if (!g.multi_d) return 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = 0; k < N3; ++k)
      for (int i = 0; i < N1; ++i) {
        int bi = bcs[6*m+2], bo = bcs[6*m+3];
        for (int j = 0; j < ng; ++j) {
          if (bi == AKMI_BC_REFLECT) {
            B1(m,k,js-j-1,i) = B1(m,k,js+j,i);
            if (i == N1-1) B1(m,k,js-j-1,i+1) = B1(m,k,js+j,i+1);
            B2(m,k,js-j-1,i) = -B2(m,k,js+j+1,i);
            B3(m,k,js-j-1,i) = B3(m,k,js+j,i);
            if (k == N3-1) B3(m,k+1,js-j-1,i) = B3(m,k+1,js+j,i);
          } else if (bi == AKMI_BC_OUTFLOW || bi == AKMI_BC_DIODE || bi == AKMI_BC_VACUUM) {
            B1(m,k,js-j-1,i) = B1(m,k,js,i);
            if (i == N1-1) B1(m,k,js-j-1,i+1) = B1(m,k,js,i+1);
            B2(m,k,js-j-1,i) = B2(m,k,js,i);
            B3(m,k,js-j-1,i) = B3(m,k,js,i);
            if (k == N3-1) B3(m,k+1,js-j-1,i) = B3(m,k+1,js,i);
          } else if (bi == AKMI_BC_INFLOW) {
            B1(m,k,js-j-1,i) = b_in[6*0 + 2];
            if (i == N1-1) B1(m,k,js-j-1,i+1) = b_in[6*0 + 2];
            B2(m,k,js-j-1,i) = b_in[6*1 + 2];
            B3(m,k,js-j-1,i) = b_in[6*2 + 2];
            if (k == N3-1) B3(m,k+1,js-j-1,i) = b_in[6*2 + 2];
          }
        }
        for (int j = 0; j < ng; ++j) {
          if (bo == AKMI_BC_REFLECT) {
            B1(m,k,je+j+1,i) = B1(m,k,je-j,i);
            if (i == N1-1) B1(m,k,je+j+1,i+1) = B1(m,k,je-j,i+1);
            B2(m,k,je+j+2,i) = -B2(m,k,je-j,i);
            B3(m,k,je+j+1,i) = B3(m,k,je-j,i);
            if (k == N3-1) B3(m,k+1,je+j+1,i) = B3(m,k+1,je-j,i);
          } else if (bo == AKMI_BC_OUTFLOW || bo == AKMI_BC_DIODE || bo == AKMI_BC_VACUUM) {
            B1(m,k,je+j+1,i) = B1(m,k,je,i);
            if (i == N1-1) B1(m,k,je+j+1,i+1) = B1(m,k,je,i+1);
            B2(m,k,je+j+2,i) = B2(m,k,je+1,i);
            B3(m,k,je+j+1,i) = B3(m,k,je,i);
            if (k == N3-1) B3(m,k+1,je+j+1,i) = B3(m,k+1,je,i);
          } else if (bo == AKMI_BC_INFLOW) {
            B1(m,k,je+j+1,i) = b_in[6*0 + 3];
            if (i == N1-1) B1(m,k,je+j+1,i+1) = b_in[6*0 + 3];
            B2(m,k,je+j+2,i) = b_in[6*1 + 3];
            B3(m,k,je+j+1,i) = b_in[6*2 + 3];
            if (k == N3-1) B3(m,k+1,je+j+1,i) = b_in[6*2 + 3];
          }
        }
      }
  if (!g.three_d) return 0;
  for (int m = 0; m < g.nmb; ++m)
    for (int j = 0; j < N2; ++j)
      for (int i = 0; i < N1; ++i) {
        int bi = bcs[6*m+4], bo = bcs[6*m+5];
        for (int k = 0; k < ng; ++k) {
          if (bi == AKMI_BC_REFLECT) {
            B1(m,ks-k-1,j,i) = B1(m,ks+k,j,i);
            if (i == N1-1) B1(m,ks-k-1,j,i+1) = B1(m,ks+k,j,i+1);
            B2(m,ks-k-1,j,i) = B2(m,ks+k,j,i);
            if (j == N2-1) B2(m,ks-k-1,j+1,i) = B2(m,ks+k,j+1,i);
            B3(m,ks-k-1,j,i) = -B3(m,ks+k+1,j,i);
          } else if (bi == AKMI_BC_OUTFLOW || bi == AKMI_BC_DIODE || bi == AKMI_BC_VACUUM) {
            B1(m,ks-k-1,j,i) = B1(m,ks,j,i);
            if (i == N1-1) B1(m,ks-k-1,j,i+1) = B1(m,ks,j,i+1);
            B2(m,ks-k-1,j,i) = B2(m,ks,j,i);
            if (j == N2-1) B2(m,ks-k-1,j+1,i) = B2(m,ks,j+1,i);
            B3(m,ks-k-1,j,i) = B3(m,ks,j,i);
          } else if (bi == AKMI_BC_INFLOW) {
            B1(m,ks-k-1,j,i) = b_in[6*0 + 4];
            if (i == N1-1) B1(m,ks-k-1,j,i+1) = b_in[6*0 + 4];
            B2(m,ks-k-1,j,i) = b_in[6*1 + 4];
            if (j == N2-1) B2(m,ks-k-1,j+1,i) = b_in[6*1 + 4];
            B3(m,ks-k-1,j,i) = b_in[6*2 + 4];
          }
        }
        for (int k = 0; k < ng; ++k) {
          if (bo == AKMI_BC_REFLECT) {
            B1(m,ke+k+1,j,i) = B1(m,ke-k,j,i);
            if (i == N1-1) B1(m,ke+k+1,j,i+1) = B1(m,ke-k,j,i+1);
            B2(m,ke+k+1,j,i) = B2(m,ke-k,j,i);
            if (j == N2-1) B2(m,ke+k+1,j+1,i) = B2(m,ke-k,j+1,i);
            B3(m,ke+k+2,j,i) = -B3(m,ke-k,j,i);
          } else if (bo == AKMI_BC_OUTFLOW || bo == AKMI_BC_DIODE || bo == AKMI_BC_VACUUM) {
            B1(m,ke+k+1,j,i) = B1(m,ke,j,i);
            if (i == N1-1) B1(m,ke+k+1,j,i+1) = B1(m,ke,j,i+1);
            B2(m,ke+k+1,j,i) = B2(m,ke,j,i);
            if (j == N2-1) B2(m,ke+k+1,j+1,i) = B2(m,ke,j+1,i);
            B3(m,ke+k+2,j,i) = B3(m,ke+1,j,i);
          } else if (bo == AKMI_BC_INFLOW) {
            B1(m,ke+k+1,j,i) = b_in[6*0 + 5];
            if (i == N1-1) B1(m,ke+k+1,j,i+1) = b_in[6*0 + 5];
            B2(m,ke+k+1,j,i) = b_in[6*1 + 5];
            if (j == N2-1) B2(m,ke+k+1,j+1,i) = b_in[6*1 + 5];
            B3(m,ke+k+2,j,i) = b_in[6*2 + 5];
          }
        }
      }
  return 0;
}

int akref_bfield_bcs(const akmi_pack *p, const int *bcs, double *bx1f, double *bx2f,
                     double *bx3f) {
  return bfield_bcs_impl(p, bcs, NULL, bx1f, bx2f, bx3f);
}
int akref_bfield_bcs_inflow(const akmi_pack *p, const int *bcs, const double *b_in, double *bx1f,
                            double *bx2f, double *bx3f) {
  return bfield_bcs_impl(p, bcs, b_in, bx1f, bx2f, bx3f);
}
/* twins of akmi_hydro_bcs_dirs / akmi_bfield_bcs_dirs (include/akmi.h): `dirs` only says which directions have a physical
 * boundary at all; the reference applies all of them (a direction without one is a no-op) */
int akref_hydro_bcs_dirs(const akmi_pack *p, int nvar, const int *bcs, int dirs, const double *u_in, double *u) {
  (void)dirs;
  return akref_hydro_bcs_inflow(p, nvar, bcs, u_in, u);
}
int akref_bfield_bcs_dirs(const akmi_pack *p, const int *bcs, int dirs, const double *b_in, double *bx1f, double *bx2f,
                          double *bx3f) {
  (void)dirs;
  return akref_bfield_bcs_inflow(p, bcs, b_in, bx1f, bx2f, bx3f);
}

/* ---- SMR/AMR operators between a MeshBlock and its coarse buffer -------------
 * Coarse arrays have cnx = nx/2 active cells and the same number of ghost cells:
 * (nmb,nvar,cN3,cN2,cN1), faces +1 in their own direction; cis = ng, cjs = ng|0, cks = ng|0
 * (src/mesh/mesh.cpp:286-330). */
typedef struct { int cN1, cN2, cN3, cis, cie, cjs, cje, cks, cke; } CG;
static CG mkCG(const G *g) {
  CG c;
  const int cnx1 = g->nx1/2, cnx2 = g->multi_d ? g->nx2/2 : 1, cnx3 = g->three_d ? g->nx3/2 : 1;
  c.cN1 = cnx1 + 2*g->ng; c.cN2 = g->multi_d ? cnx2 + 2*g->ng : 1; c.cN3 = g->three_d ? cnx3 + 2*g->ng : 1;
  c.cis = g->ng; c.cie = c.cis + cnx1 - 1;
  c.cjs = g->multi_d ? g->ng : 0; c.cje = g->multi_d ? c.cjs + cnx2 - 1 : 0;
  c.cks = g->three_d ? g->ng : 0; c.cke = g->three_d ? c.cks + cnx3 - 1 : 0;
  return c;
}

/* MeshRefinement::RestrictCC, src/mesh/mesh_refinement.cpp:1223-1277 (second-order average) */
int akref_restrict_cc(const akmi_pack *p, int nvar, const double *u, double *cu) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
#define U(n,k,j,i) u[ix5(nvar,N3,N2,N1,m,n,k,j,i)]
#define CU(n,k,j,i) cu[ix5(nvar,c.cN3,c.cN2,c.cN1,m,n,k,j,i)]
  for (int m = 0; m < g.nmb; ++m) for (int n = 0; n < nvar; ++n)
    for (int k = c.cks; k <= c.cke; ++k) for (int j = c.cjs; j <= c.cje; ++j)
      for (int i = c.cis; i <= c.cie; ++i) {
        int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
        if (!g.multi_d)
          CU(n,k,j,i) = 0.5*(U(n,k,j,fi) + U(n,k,j,fi+1));
        else if (!g.three_d)
          CU(n,k,j,i) = 0.25*(U(n,k,fj,fi) + U(n,k,fj,fi+1) + U(n,k,fj+1,fi) + U(n,k,fj+1,fi+1));
        else
          CU(n,k,j,i) = 0.125*(U(n,fk,fj,fi) + U(n,fk,fj,fi+1) + U(n,fk,fj+1,fi) + U(n,fk,fj+1,fi+1)
                             + U(n,fk+1,fj,fi) + U(n,fk+1,fj,fi+1) + U(n,fk+1,fj+1,fi) + U(n,fk+1,fj+1,fi+1));
      }
#undef U
#undef CU
  return 0;
}

/* Restricted face fluxes a fine MeshBlock hands to a coarser neighbour, in the buffer order of
 * PackAndSendFluxCC (src/bvals/flux_correct_cc.cpp:78-148): dir = face normal, box = coarse index box
 * (il,iu,jl,ju,kl,ku) whose extent along dir is one face; flx is the face-shaped flux of that direction.
 * out[m][(t1-t1l) + n1*((t2-t2l) + n2*v)], (t1,t2) = (j,k) / (i,k) / (i,j).  TEST INFRASTRUCTURE. */
int akref_restrict_flux_cc(const akmi_pack *p, int nvar, int dir, const int *box, const double *flx,
                           double *out) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int il = box[0], iu = box[1], jl = box[2], ju = box[3], kl = box[4], ku = box[5];
  const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
  const int f3 = N3 + (dir == 2), f2 = N2 + (dir == 1), f1 = N1 + (dir == 0);
  const size_t per = (size_t)nvar*ni*nj*nk;
#define FX(v,k,j,i) flx[ix5(nvar,f3,f2,f1,m,v,k,j,i)]
  for (int m = 0; m < g.nmb; ++m) for (int v = 0; v < nvar; ++v)
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
      double r;
      size_t o;
      if (dir == 0) {
        if (!g.multi_d) r = FX(v,0,0,fi);
        else if (!g.three_d) r = 0.5*(FX(v,0,fj,fi) + FX(v,0,fj+1,fi));
        else r = 0.25*(FX(v,fk,fj,fi) + FX(v,fk,fj+1,fi) + FX(v,fk+1,fj,fi) + FX(v,fk+1,fj+1,fi));
        o = (size_t)(j - jl) + (size_t)nj*((k - kl) + (size_t)nk*v);
      } else if (dir == 1) {
        if (!g.three_d) r = 0.5*(FX(v,0,fj,fi) + FX(v,0,fj,fi+1));
        else r = 0.25*(FX(v,fk,fj,fi) + FX(v,fk,fj,fi+1) + FX(v,fk+1,fj,fi) + FX(v,fk+1,fj,fi+1));
        o = (size_t)(i - il) + (size_t)ni*((k - kl) + (size_t)nk*v);
      } else {
        r = 0.25*(FX(v,fk,fj,fi) + FX(v,fk,fj,fi+1) + FX(v,fk,fj+1,fi) + FX(v,fk,fj+1,fi+1));
        o = (size_t)(i - il) + (size_t)ni*((j - jl) + (size_t)nj*v);
      }
      out[m*per + o] = r;
    }
#undef FX
  return 0;
}

/* Restricted edge EMFs for a coarser neighbour (PackAndSendFluxFC, src/bvals/flux_correct_fc.cpp:84-360):
 * the two fine edges that make up a coarse edge are averaged along the edge's own direction (no
 * averaging along a direction the mesh does not have).  comp = 0,1,2 for x1e,x2e,x3e; box in coarse
 * indices; out[m][(i-il) + ni*((j-jl) + nj*(k-kl))].  TEST INFRASTRUCTURE. */
int akref_restrict_emf(const akmi_pack *p, int comp, const int *box, const double *e, double *out) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int il = box[0], iu = box[1], jl = box[2], ju = box[3], kl = box[4], ku = box[5];
  const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
  const int e3 = N3 + (comp != 2), e2 = N2 + (comp != 1), e1 = N1 + (comp != 0);   /* efld shapes */
  const size_t per = (size_t)ni*nj*nk;
#define EE(k,j,i) e[ix4(e3,e2,e1,m,k,j,i)]
  for (int m = 0; m < g.nmb; ++m)
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const int fi = 2*i - c.cis, fj = g.multi_d ? 2*j - c.cjs : 0, fk = g.three_d ? 2*k - c.cks : 0;
      double r;
      if (comp == 0) r = g.multi_d ? 0.5*(EE(fk,fj,fi) + EE(fk,fj,fi+1)) : EE(fk,fj,fi);
      else if (comp == 1) r = g.multi_d ? 0.5*(EE(fk,fj,fi) + EE(fk,fj+1,fi)) : EE(fk,fj,fi);
      else r = g.three_d ? 0.5*(EE(fk,fj,fi) + EE(fk+1,fj,fi)) : EE(fk,fj,fi);
      out[m*per + (size_t)(i - il) + (size_t)ni*((j - jl) + (size_t)nj*(k - kl))] = r;
    }
#undef EE
  return 0;
}

/* Primitive -> conserved over a box of fine cells (il,iu,jl,ju,kl,ku): SingleP2C_IdealHyd / _IdealMHD /
 * _Isothermal* (src/eos/ideal_c2p_hyd.hpp:76-83, ideal_c2p_mhd.hpp:75-84) as MeshBoundaryValuesCC::
 * PrimToConsFineBndry applies them after prolongating primitives (src/bvals/prolong_prims.cpp:190-300,
 * 465-…); passive scalars u = d*s.  bcc == NULL: hydro.  TEST INFRASTRUCTURE. */
int akref_prim2cons(const akmi_pack *p, const int *box, const double *w, const double *bcc, double *u) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3, nv = p->nvar;
  const int nfl = p->is_ideal ? 5 : 4;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = box[4]; k <= box[5]; ++k) for (int j = box[2]; j <= box[3]; ++j)
      for (int i = box[0]; i <= box[1]; ++i) {
#define W(n) w[ix5(nv,N3,N2,N1,m,n,k,j,i)]
#define UU(n) u[ix5(nv,N3,N2,N1,m,n,k,j,i)]
        const double d = W(0), vx = W(1), vy = W(2), vz = W(3);
        UU(0) = d; UU(1) = d*vx; UU(2) = d*vy; UU(3) = d*vz;
        if (p->is_ideal) {
          if (bcc) {
            const double bx = bcc[ix5(3,N3,N2,N1,m,0,k,j,i)], by = bcc[ix5(3,N3,N2,N1,m,1,k,j,i)],
                         bz = bcc[ix5(3,N3,N2,N1,m,2,k,j,i)];
            UU(4) = W(4) + 0.5*(d*(vx*vx + vy*vy + vz*vz) + (bx*bx + by*by + bz*bz));
          } else {
            UU(4) = W(4) + 0.5*d*(vx*vx + vy*vy + vz*vz);
          }
        }
        for (int n = nfl; n < nv; ++n) UU(n) = d*W(n);
#undef W
#undef UU
      }
  return 0;
}

#define FB1(k,j,i) b1[ix4(N3,N2,N1+1,m,k,j,i)]
#define FB2(k,j,i) b2[ix4(N3,N2+1,N1,m,k,j,i)]
#define FB3(k,j,i) b3[ix4(N3+1,N2,N1,m,k,j,i)]
#define CB1(k,j,i) cb1[ix4(c.cN3,c.cN2,c.cN1+1,m,k,j,i)]
#define CB2(k,j,i) cb2[ix4(c.cN3,c.cN2+1,c.cN1,m,k,j,i)]
#define CB3(k,j,i) cb3[ix4(c.cN3+1,c.cN2,c.cN1,m,k,j,i)]

/* MeshRefinement::RestrictFC, src/mesh/mesh_refinement.cpp:1283-1382 (area averages of the faces) */
int akref_restrict_fc(const akmi_pack *p, const double *b1, const double *b2, const double *b3,
                      double *cb1, double *cb2, double *cb3) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = c.cks; k <= c.cke; ++k) for (int j = c.cjs; j <= c.cje; ++j)
      for (int i = c.cis; i <= c.cie; ++i) {
        int fi = 2*i - c.cis, fj = 2*j - c.cjs, fk = 2*k - c.cks;
        if (!g.multi_d) {
          CB1(k,j,i) = FB1(k,j,fi);
          if (i == c.cie) CB1(k,j,i+1) = FB1(k,j,fi+2);
          double b2c = 0.5*(FB2(k,j,fi) + FB2(k,j,fi+1));
          CB2(k,j,i) = b2c; CB2(k,j+1,i) = b2c;
          double b3c = 0.5*(FB3(k,j,fi) + FB3(k,j,fi+1));
          CB3(k,j,i) = b3c; CB3(k+1,j,i) = b3c;
        } else if (!g.three_d) {
          CB1(k,j,i) = 0.5*(FB1(k,fj,fi) + FB1(k,fj+1,fi));
          if (i == c.cie) CB1(k,j,i+1) = 0.5*(FB1(k,fj,fi+2) + FB1(k,fj+1,fi+2));
          CB2(k,j,i) = 0.5*(FB2(k,fj,fi) + FB2(k,fj,fi+1));
          if (j == c.cje) CB2(k,j+1,i) = 0.5*(FB2(k,fj+2,fi) + FB2(k,fj+2,fi+1));
          double b3c = 0.25*(FB3(k,fj,fi) + FB3(k,fj,fi+1) + FB3(k,fj+1,fi) + FB3(k,fj+1,fi+1));
          CB3(k,j,i) = b3c; CB3(k+1,j,i) = b3c;
        } else {
          CB1(k,j,i) = 0.25*(FB1(fk,fj,fi) + FB1(fk,fj+1,fi) + FB1(fk+1,fj,fi) + FB1(fk+1,fj+1,fi));
          if (i == c.cie)
            CB1(k,j,i+1) = 0.25*(FB1(fk,fj,fi+2) + FB1(fk,fj+1,fi+2) + FB1(fk+1,fj,fi+2) + FB1(fk+1,fj+1,fi+2));
          CB2(k,j,i) = 0.25*(FB2(fk,fj,fi) + FB2(fk,fj,fi+1) + FB2(fk+1,fj,fi) + FB2(fk+1,fj,fi+1));
          if (j == c.cje)
            CB2(k,j+1,i) = 0.25*(FB2(fk,fj+2,fi) + FB2(fk,fj+2,fi+1) + FB2(fk+1,fj+2,fi) + FB2(fk+1,fj+2,fi+1));
          CB3(k,j,i) = 0.25*(FB3(fk,fj,fi) + FB3(fk,fj,fi+1) + FB3(fk,fj+1,fi) + FB3(fk,fj+1,fi+1));
          if (k == c.cke)
            CB3(k+1,j,i) = 0.25*(FB3(fk+2,fj,fi) + FB3(fk+2,fj,fi+1) + FB3(fk+2,fj+1,fi) + FB3(fk+2,fj+1,fi+1));
        }
      }
  return 0;
}

/* twins of akmi_rk_update_oop / akmi_mhd_ct_oop (include/akmi.h): the reference's own sequence, CopyCons
 * (hydro_tasks.cpp:130-152, mhd_tasks.cpp:162-170) then the update, with the roles of the registers as the ABI
 * describes them: src stays, dst receives the new state */
int akref_rk_update_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *u0, double *u1,
                        const double *flx1, const double *flx2, const double *flx3, int face_shaped) {
  G g = mkG(p);
  memcpy(u1, u0, sizeof(double)*(size_t)p->nmb*p->nvar*g.N3*g.N2*g.N1);
  return akref_rk_update(p, gam0, gam1, beta_dt, u1, u0, flx1, flx2, flx3, face_shaped);
}
int akref_mhd_ct_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *e1, const double *e2,
                     const double *e3, const double *b0x1f, const double *b0x2f, const double *b0x3f, double *b1x1f,
                     double *b1x2f, double *b1x3f) {
  G g = mkG(p);
  memcpy(b1x1f, b0x1f, sizeof(double)*(size_t)p->nmb*g.N3*g.N2*(g.N1 + 1));
  memcpy(b1x2f, b0x2f, sizeof(double)*(size_t)p->nmb*g.N3*(g.N2 + 1)*g.N1);
  memcpy(b1x3f, b0x3f, sizeof(double)*(size_t)p->nmb*(g.N3 + 1)*g.N2*g.N1);
  return akref_mhd_ct(p, gam0, gam1, beta_dt, e1, e2, e3, b1x1f, b1x2f, b1x3f, b0x1f, b0x2f, b0x3f);
}

/* twins of akmi_restrict_cc_masked / akmi_restrict_fc_masked (include/akmi.h): RestrictCC / RestrictFC for the
 * MeshBlocks with mask[m] != 0 (NULL: all), one block at a time through the functions above */
int akref_restrict_cc_masked(const akmi_pack *p, int nvar, const unsigned char *mask, const double *u, double *cu) {
  G g = mkG(p); CG c = mkCG(&g);
  const size_t fs = (size_t)nvar*g.N3*g.N2*g.N1, cs = (size_t)nvar*c.cN3*c.cN2*c.cN1;
  for (int m = 0; m < p->nmb; ++m) {
    if (mask && !mask[m]) continue;
    akmi_pack q = *p;
    q.nmb = 1; q.dx = p->dx + 3*m;
    akref_restrict_cc(&q, nvar, u + m*fs, cu + m*cs);
  }
  return 0;
}
int akref_restrict_fc_masked(const akmi_pack *p, const unsigned char *mask, const double *b1, const double *b2,
                             const double *b3, double *cb1, double *cb2, double *cb3) {
  G g = mkG(p); CG c = mkCG(&g);
  const size_t f1 = (size_t)g.N3*g.N2*(g.N1 + 1), f2 = (size_t)g.N3*(g.N2 + 1)*g.N1, f3 = (size_t)(g.N3 + 1)*g.N2*g.N1;
  const size_t c1 = (size_t)c.cN3*c.cN2*(c.cN1 + 1), c2 = (size_t)c.cN3*(c.cN2 + 1)*c.cN1,
               c3 = (size_t)(c.cN3 + 1)*c.cN2*c.cN1;
  for (int m = 0; m < p->nmb; ++m) {
    if (mask && !mask[m]) continue;
    akmi_pack q = *p;
    q.nmb = 1; q.dx = p->dx + 3*m;
    akref_restrict_fc(&q, b1 + m*f1, b2 + m*f2, b3 + m*f3, cb1 + m*c1, cb2 + m*c2, cb3 + m*c3);
  }
  return 0;
}

static inline double sgn_(double x) { return (x < 0.0) ? -1.0 : 1.0; }     /* SIGN, src/athena.hpp:52 */
static inline double mm8(double dl, double dr) {                          /* 0.125*(SIGN+SIGN)*fmin */
  return 0.125*(sgn_(dl) + sgn_(dr))*fmin(fabs(dl), fabs(dr));
}

/* ProlongCC, src/mesh/prolongation.hpp:19-63, over the box of COARSE cells box = {il,iu,jl,ju,kl,ku};
 * fine index fi = (i - cis)*2 + is (src/bvals/prolongation.cpp:524-526) */
int akref_prolong_cc(const akmi_pack *p, int nvar, const int box[6], const double *cu, double *u) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
#define A(n,k,j,i) u[ix5(nvar,N3,N2,N1,m,n,k,j,i)]
#define CA(n,k,j,i) cu[ix5(nvar,c.cN3,c.cN2,c.cN1,m,n,k,j,i)]
  for (int m = 0; m < g.nmb; ++m) for (int v = 0; v < nvar; ++v)
    for (int k = box[4]; k <= box[5]; ++k) for (int j = box[2]; j <= box[3]; ++j)
      for (int i = box[0]; i <= box[1]; ++i) {
        int fi = (i - c.cis)*2 + g.is, fj = (j - c.cjs)*2 + g.js, fk = (k - c.cks)*2 + g.ks;
        double dvar1 = mm8(CA(v,k,j,i) - CA(v,k,j,i-1), CA(v,k,j,i+1) - CA(v,k,j,i));
        double dvar2 = 0.0, dvar3 = 0.0;
        if (g.multi_d) dvar2 = mm8(CA(v,k,j,i) - CA(v,k,j-1,i), CA(v,k,j+1,i) - CA(v,k,j,i));
        if (g.three_d) dvar3 = mm8(CA(v,k,j,i) - CA(v,k-1,j,i), CA(v,k+1,j,i) - CA(v,k,j,i));
        A(v,fk,fj,fi) = CA(v,k,j,i) - dvar1 - dvar2 - dvar3;
        A(v,fk,fj,fi+1) = CA(v,k,j,i) + dvar1 - dvar2 - dvar3;
        if (g.multi_d) {
          A(v,fk,fj+1,fi) = CA(v,k,j,i) - dvar1 + dvar2 - dvar3;
          A(v,fk,fj+1,fi+1) = CA(v,k,j,i) + dvar1 + dvar2 - dvar3;
        }
        if (g.three_d) {
          A(v,fk+1,fj,fi) = CA(v,k,j,i) - dvar1 - dvar2 + dvar3;
          A(v,fk+1,fj,fi+1) = CA(v,k,j,i) + dvar1 - dvar2 + dvar3;
          A(v,fk+1,fj+1,fi) = CA(v,k,j,i) - dvar1 + dvar2 + dvar3;
          A(v,fk+1,fj+1,fi+1) = CA(v,k,j,i) + dvar1 + dvar2 + dvar3;
        }
      }
#undef A
#undef CA
  return 0;
}

/* ProlongFCSharedX1Face/X2Face/X3Face, src/mesh/prolongation.hpp:69-160: faces of component comp
 * (0,1,2) that a fine block shares with coarse faces, over a box of coarse FACE indices */
int akref_prolong_fc_shared(const akmi_pack *p, int comp, const int box[6], const double *cb,
                            double *b) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const double *cb1 = cb, *cb2 = cb, *cb3 = cb;
  double *b1 = b, *b2 = b, *b3 = b;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = box[4]; k <= box[5]; ++k) for (int j = box[2]; j <= box[3]; ++j)
      for (int i = box[0]; i <= box[1]; ++i) {
        int fi = (i - c.cis)*2 + g.is;
        int fj = g.multi_d ? (j - c.cjs)*2 + g.js : j;
        int fk = g.three_d ? (k - c.cks)*2 + g.ks : k;
        if (comp == 0) {
          double dvar2 = 0.0, dvar3 = 0.0;
          if (g.multi_d) dvar2 = mm8(CB1(k,j,i) - CB1(k,j-1,i), CB1(k,j+1,i) - CB1(k,j,i));
          if (g.three_d) dvar3 = mm8(CB1(k,j,i) - CB1(k-1,j,i), CB1(k+1,j,i) - CB1(k,j,i));
          FB1(fk,fj,fi) = CB1(k,j,i) - dvar2 - dvar3;
          if (g.multi_d) FB1(fk,fj+1,fi) = CB1(k,j,i) + dvar2 - dvar3;
          if (g.three_d) {
            FB1(fk+1,fj,fi) = CB1(k,j,i) - dvar2 + dvar3;
            FB1(fk+1,fj+1,fi) = CB1(k,j,i) + dvar2 + dvar3;
          }
        } else if (comp == 1) {
          double dvar1 = mm8(CB2(k,j,i) - CB2(k,j,i-1), CB2(k,j,i+1) - CB2(k,j,i));
          double dvar3 = 0.0;
          if (g.three_d) dvar3 = mm8(CB2(k,j,i) - CB2(k-1,j,i), CB2(k+1,j,i) - CB2(k,j,i));
          FB2(fk,fj,fi) = CB2(k,j,i) - dvar1 - dvar3;
          FB2(fk,fj,fi+1) = CB2(k,j,i) + dvar1 - dvar3;
          if (g.three_d) {
            FB2(fk+1,fj,fi) = CB2(k,j,i) - dvar1 + dvar3;
            FB2(fk+1,fj,fi+1) = CB2(k,j,i) + dvar1 + dvar3;
          }
        } else {
          double dvar1 = mm8(CB3(k,j,i) - CB3(k,j,i-1), CB3(k,j,i+1) - CB3(k,j,i));
          double dvar2 = 0.0;
          if (g.multi_d) dvar2 = mm8(CB3(k,j,i) - CB3(k,j-1,i), CB3(k,j+1,i) - CB3(k,j,i));
          FB3(fk,fj,fi) = CB3(k,j,i) - dvar1 - dvar2;
          FB3(fk,fj,fi+1) = CB3(k,j,i) + dvar1 - dvar2;
          if (g.multi_d) {
            FB3(fk,fj+1,fi) = CB3(k,j,i) - dvar1 + dvar2;
            FB3(fk,fj+1,fi+1) = CB3(k,j,i) + dvar1 + dvar2;
          }
        }
      }
  return 0;
}

/* ProlongFCInternal, src/mesh/prolongation.hpp:166-230 (divergence-preserving interpolation of Toth &
 * Roe 2002 onto the faces inside a coarse cell) and the 1-D rule of src/bvals/prolongation.cpp:765-770,
 * over a box of coarse CELL indices; the shared faces must have been set before */
int akref_prolong_fc_internal(const akmi_pack *p, const int box[6], double *b1, double *b2, double *b3) {
  G g = mkG(p); CG c = mkCG(&g);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = box[4]; k <= box[5]; ++k) for (int j = box[2]; j <= box[3]; ++j)
      for (int i = box[0]; i <= box[1]; ++i) {
        int fi = (i - c.cis)*2 + g.is, fj = (j - c.cjs)*2 + g.js, fk = (k - c.cks)*2 + g.ks;
        if (!g.multi_d) {
          FB1(fk,fj,fi+1) = 0.5*(FB1(fk,fj,fi) + FB1(fk,fj,fi+2));
        } else if (g.three_d) {
          double Uxx = 0.0, Vyy = 0.0, Wzz = 0.0, Uxyz = 0.0, Vxyz = 0.0, Wxyz = 0.0;
          for (int jj = 0; jj < 2; jj++) {
            int jsgn = 2*jj - 1;
            int fjj = fj + jj, fjp = fj + 2*jj;
            for (int ii = 0; ii < 2; ii++) {
              int isgn = 2*ii - 1;
              int fii = fi + ii, fip = fi + 2*ii;
              Uxx += isgn*(jsgn*(FB2(fk,fjp,fii) + FB2(fk+1,fjp,fii)) + (FB3(fk+2,fjj,fii) - FB3(fk,fjj,fii)));
              Vyy += jsgn*((FB3(fk+2,fjj,fii) - FB3(fk,fjj,fii)) + isgn*(FB1(fk,fjj,fip) + FB1(fk+1,fjj,fip)));
              Wzz += isgn*(FB1(fk+1,fjj,fip) - FB1(fk,fjj,fip)) + jsgn*(FB2(fk+1,fjp,fii) - FB2(fk,fjp,fii));
              Uxyz += isgn*jsgn*(FB1(fk+1,fjj,fip) - FB1(fk,fjj,fip));
              Vxyz += isgn*jsgn*(FB2(fk+1,fjp,fii) - FB2(fk,fjp,fii));
              Wxyz += isgn*jsgn*(FB3(fk+2,fjj,fii) - FB3(fk,fjj,fii));
            }
          }
          Uxx *= 0.125; Vyy *= 0.125; Wzz *= 0.125;
          Uxyz *= 0.0625; Vxyz *= 0.0625; Wxyz *= 0.0625;
          FB1(fk,fj,fi+1) = 0.5*(FB1(fk,fj,fi) + FB1(fk,fj,fi+2)) + Uxx - Vxyz - Wxyz;
          FB1(fk,fj+1,fi+1) = 0.5*(FB1(fk,fj+1,fi) + FB1(fk,fj+1,fi+2)) + Uxx - Vxyz + Wxyz;
          FB1(fk+1,fj,fi+1) = 0.5*(FB1(fk+1,fj,fi) + FB1(fk+1,fj,fi+2)) + Uxx + Vxyz - Wxyz;
          FB1(fk+1,fj+1,fi+1) = 0.5*(FB1(fk+1,fj+1,fi) + FB1(fk+1,fj+1,fi+2)) + Uxx + Vxyz + Wxyz;
          FB2(fk,fj+1,fi) = 0.5*(FB2(fk,fj,fi) + FB2(fk,fj+2,fi)) + Vyy - Uxyz - Wxyz;
          FB2(fk,fj+1,fi+1) = 0.5*(FB2(fk,fj,fi+1) + FB2(fk,fj+2,fi+1)) + Vyy - Uxyz + Wxyz;
          FB2(fk+1,fj+1,fi) = 0.5*(FB2(fk+1,fj,fi) + FB2(fk+1,fj+2,fi)) + Vyy + Uxyz - Wxyz;
          FB2(fk+1,fj+1,fi+1) = 0.5*(FB2(fk+1,fj,fi+1) + FB2(fk+1,fj+2,fi+1)) + Vyy + Uxyz + Wxyz;
          FB3(fk+1,fj,fi) = 0.5*(FB3(fk+2,fj,fi) + FB3(fk,fj,fi)) + Wzz - Uxyz - Vxyz;
          FB3(fk+1,fj,fi+1) = 0.5*(FB3(fk+2,fj,fi+1) + FB3(fk,fj,fi+1)) + Wzz - Uxyz + Vxyz;
          FB3(fk+1,fj+1,fi) = 0.5*(FB3(fk+2,fj+1,fi) + FB3(fk,fj+1,fi)) + Wzz + Uxyz - Vxyz;
          FB3(fk+1,fj+1,fi+1) = 0.5*(FB3(fk+2,fj+1,fi+1) + FB3(fk,fj+1,fi+1)) + Wzz + Uxyz + Vxyz;
        } else {
          double tmp1 = 0.25*(FB2(fk,fj+2,fi+1) - FB2(fk,fj,fi+1) - FB2(fk,fj+2,fi) + FB2(fk,fj,fi));
          double tmp2 = 0.25*(FB1(fk,fj,fi) - FB1(fk,fj,fi+2) - FB1(fk,fj+1,fi) + FB1(fk,fj+1,fi+2));
          FB1(fk,fj,fi+1) = 0.5*(FB1(fk,fj,fi) + FB1(fk,fj,fi+2)) + tmp1;
          FB1(fk,fj+1,fi+1) = 0.5*(FB1(fk,fj+1,fi) + FB1(fk,fj+1,fi+2)) + tmp1;
          FB2(fk,fj+1,fi) = 0.5*(FB2(fk,fj,fi) + FB2(fk,fj+2,fi)) + tmp2;
          FB2(fk,fj+1,fi+1) = 0.5*(FB2(fk,fj,fi+1) + FB2(fk,fj+2,fi+1)) + tmp2;
        }
      }
  return 0;
}
#undef FB1
#undef FB2
#undef FB3
#undef CB1
#undef CB2
#undef CB3

/* ---- ambipolar diffusion, constant eta_ad (src/diffusion/ambipolar.cpp) ---------------------------- */
#define AB1(k,j,i) bx1f[ix4(N3,N2,N1+1,m,(k),(j),(i))]
#define AB2(k,j,i) bx2f[ix4(N3,N2+1,N1,m,(k),(j),(i))]
#define AB3(k,j,i) bx3f[ix4(N3+1,N2,N1,m,(k),(j),(i))]
#define ABC(n,k,j,i) bcc0[ix5(3,N3,N2,N1,m,(n),(k),(j),(i))]
#define AE1(k,j,i) e1[ix4(N3+1,N2+1,N1,m,(k),(j),(i))]
#define AE2(k,j,i) e2[ix4(N3+1,N2,N1+1,m,(k),(j),(i))]
#define AE3(k,j,i) e3[ix4(N3,N2+1,N1+1,m,(k),(j),(i))]
/* EdgeJ1/2/3, ambipolar.cpp:30-58 */
#define EJ1(k,j,i) ((g.multi_d ? (AB3(k,j,i) - AB3(k,(j)-1,i))/dx2 : 0.0) - (g.three_d ? (AB2(k,j,i) - AB2((k)-1,j,i))/dx3 : 0.0))

static inline double edge_j1(const G *g, const double *bx2f, const double *bx3f, int N1, int N2, int N3,
                             int m, int k, int j, int i, double dx2, double dx3) {
  double j1 = 0.0;
  if (g->multi_d) j1 += (AB3(k,j,i) - AB3(k,j-1,i))/dx2;
  if (g->three_d) j1 -= (AB2(k,j,i) - AB2(k-1,j,i))/dx3;
  return j1;
}
static inline double edge_j2(const G *g, const double *bx1f, const double *bx3f, int N1, int N2, int N3,
                             int m, int k, int j, int i, double dx1, double dx3) {
  double j2 = -(AB3(k,j,i) - AB3(k,j,i-1))/dx1;
  if (g->three_d) j2 += (AB1(k,j,i) - AB1(k-1,j,i))/dx3;
  return j2;
}
static inline double edge_j3(const G *g, const double *bx1f, const double *bx2f, int N1, int N2, int N3,
                             int m, int k, int j, int i, double dx1, double dx2) {
  double j3 = (AB2(k,j,i) - AB2(k,j,i-1))/dx1;
  if (g->multi_d) j3 -= (AB1(k,j,i) - AB1(k,j-1,i))/dx2;
  return j3;
}
#undef EJ1

/* Resistivity::AddEMFConstantAmbipolar, src/diffusion/ambipolar.cpp:66-246:
 * E += eta_ad*(B^2 J - (J.B) B) with J and B averaged to each edge */
int akref_ambipolar_emfs(const akmi_pack *p, double eta, const double *bcc0, const double *bx1f,
                         const double *bx2f, const double *bx3f, double *e1, double *e2, double *e3) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int is = g.is, ie = g.ie, js = g.js, je = g.je, ks = g.ks, ke = g.ke;
#define J1(k,j,i) edge_j1(&g, bx2f, bx3f, N1, N2, N3, m, k, j, i, dx2, dx3)
#define J2(k,j,i) edge_j2(&g, bx1f, bx3f, N1, N2, N3, m, k, j, i, dx1, dx3)
#define J3(k,j,i) edge_j3(&g, bx1f, bx2f, N1, N2, N3, m, k, j, i, dx1, dx2)
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    if (!g.multi_d) {
      for (int i = is; i <= ie+1; ++i) {
        double intBx = AB1(ks,js,i);
        double intBy = 0.5*(ABC(1,ks,js,i) + ABC(1,ks,js,i-1));
        double intBz = 0.5*(ABC(2,ks,js,i) + ABC(2,ks,js,i-1));
        double intJ2 = J2(ks,js,i), intJ3 = J3(ks,js,i);
        double Bsq = SQR(intBx) + SQR(intBy) + SQR(intBz);
        double JdotB = intJ2*intBy + intJ3*intBz;
        double e2_amb = eta * (Bsq*intJ2 - JdotB*intBy);
        double e3_amb = eta * (Bsq*intJ3 - JdotB*intBz);
        AE2(ks,js,i) += e2_amb; AE2(ke+1,js,i) += e2_amb;
        AE3(ks,js,i) += e3_amb; AE3(ks,je+1,i) += e3_amb;
      }
    } else if (!g.three_d) {
      for (int j = js; j <= je+1; ++j) for (int i = is; i <= ie+1; ++i) {
        double intJ1_e1 = J1(ks,j,i);
        double intJ2_e1 = 0.25*(J2(ks,j-1,i) + J2(ks,j-1,i+1) + J2(ks,j,i) + J2(ks,j,i+1));
        double intJ3_e1 = 0.5*(J3(ks,j,i) + J3(ks,j,i+1));
        double intBx_e1 = 0.5*(ABC(0,ks,j,i) + ABC(0,ks,j-1,i));
        double intBy_e1 = AB2(ks,j,i);
        double intBz_e1 = 0.5*(ABC(2,ks,j,i) + ABC(2,ks,j-1,i));
        double Bsq_e1 = SQR(intBx_e1) + SQR(intBy_e1) + SQR(intBz_e1);
        double JdotB_e1 = intJ1_e1*intBx_e1 + intJ2_e1*intBy_e1 + intJ3_e1*intBz_e1;
        double e1_amb = eta * (Bsq_e1*intJ1_e1 - JdotB_e1*intBx_e1);
        AE1(ks,j,i) += e1_amb; AE1(ke+1,j,i) += e1_amb;

        double intJ1_e2 = 0.25*(J1(ks,j,i-1) + J1(ks,j,i) + J1(ks,j+1,i-1) + J1(ks,j+1,i));
        double intJ2_e2 = J2(ks,j,i);
        double intJ3_e2 = 0.5*(J3(ks,j,i) + J3(ks,j+1,i));
        double intBx_e2 = AB1(ks,j,i);
        double intBy_e2 = 0.5*(ABC(1,ks,j,i) + ABC(1,ks,j,i-1));
        double intBz_e2 = 0.5*(ABC(2,ks,j,i) + ABC(2,ks,j,i-1));
        double Bsq_e2 = SQR(intBx_e2) + SQR(intBy_e2) + SQR(intBz_e2);
        double JdotB_e2 = intJ1_e2*intBx_e2 + intJ2_e2*intBy_e2 + intJ3_e2*intBz_e2;
        double e2_amb = eta * (Bsq_e2*intJ2_e2 - JdotB_e2*intBy_e2);
        AE2(ks,j,i) += e2_amb; AE2(ke+1,j,i) += e2_amb;

        double intJ1_e3 = 0.5*(J1(ks,j,i-1) + J1(ks,j,i));
        double intJ2_e3 = 0.5*(J2(ks,j-1,i) + J2(ks,j,i));
        double intJ3_e3 = J3(ks,j,i);
        double intBx_e3 = 0.5*(AB1(ks,j,i) + AB1(ks,j-1,i));
        double intBy_e3 = 0.5*(AB2(ks,j,i) + AB2(ks,j,i-1));
        double intBz_e3 = 0.25*(ABC(2,ks,j,i) + ABC(2,ks,j-1,i) + ABC(2,ks,j,i-1) + ABC(2,ks,j-1,i-1));
        double Bsq_e3 = SQR(intBx_e3) + SQR(intBy_e3) + SQR(intBz_e3);
        double JdotB_e3 = intJ1_e3*intBx_e3 + intJ2_e3*intBy_e3 + intJ3_e3*intBz_e3;
        double e3_amb = eta * (Bsq_e3*intJ3_e3 - JdotB_e3*intBz_e3);
        AE3(ks,j,i) += e3_amb;
      }
    } else {
      for (int k = ks; k <= ke+1; ++k) for (int j = js; j <= je+1; ++j) for (int i = is; i <= ie+1; ++i) {
        double intJ1_e1 = J1(k,j,i);
        double intJ2_e1 = 0.25*(J2(k,j-1,i) + J2(k,j-1,i+1) + J2(k,j,i) + J2(k,j,i+1));
        double intJ3_e1 = 0.25*(J3(k-1,j,i) + J3(k-1,j,i+1) + J3(k,j,i) + J3(k,j,i+1));
        double intBx_e1 = 0.25*(ABC(0,k,j,i) + ABC(0,k-1,j,i) + ABC(0,k,j-1,i) + ABC(0,k-1,j-1,i));
        double intBy_e1 = 0.5*(AB2(k,j,i) + AB2(k-1,j,i));
        double intBz_e1 = 0.5*(AB3(k,j,i) + AB3(k,j-1,i));
        double Bsq_e1 = SQR(intBx_e1) + SQR(intBy_e1) + SQR(intBz_e1);
        double JdotB_e1 = intJ1_e1*intBx_e1 + intJ2_e1*intBy_e1 + intJ3_e1*intBz_e1;
        AE1(k,j,i) += eta * (Bsq_e1*intJ1_e1 - JdotB_e1*intBx_e1);

        double intJ1_e2 = 0.25*(J1(k,j,i-1) + J1(k,j,i) + J1(k,j+1,i-1) + J1(k,j+1,i));
        double intJ2_e2 = J2(k,j,i);
        double intJ3_e2 = 0.25*(J3(k-1,j,i) + J3(k-1,j+1,i) + J3(k,j,i) + J3(k,j+1,i));
        double intBx_e2 = 0.5*(AB1(k,j,i) + AB1(k-1,j,i));
        double intBy_e2 = 0.25*(ABC(1,k,j,i) + ABC(1,k-1,j,i) + ABC(1,k,j,i-1) + ABC(1,k-1,j,i-1));
        double intBz_e2 = 0.5*(AB3(k,j,i) + AB3(k,j,i-1));
        double Bsq_e2 = SQR(intBx_e2) + SQR(intBy_e2) + SQR(intBz_e2);
        double JdotB_e2 = intJ1_e2*intBx_e2 + intJ2_e2*intBy_e2 + intJ3_e2*intBz_e2;
        AE2(k,j,i) += eta * (Bsq_e2*intJ2_e2 - JdotB_e2*intBy_e2);

        double intJ1_e3 = 0.25*(J1(k,j,i-1) + J1(k,j,i) + J1(k+1,j,i-1) + J1(k+1,j,i));
        double intJ2_e3 = 0.25*(J2(k,j-1,i) + J2(k,j,i) + J2(k+1,j-1,i) + J2(k+1,j,i));
        double intJ3_e3 = J3(k,j,i);
        double intBx_e3 = 0.5*(AB1(k,j,i) + AB1(k,j-1,i));
        double intBy_e3 = 0.5*(AB2(k,j,i) + AB2(k,j,i-1));
        double intBz_e3 = 0.25*(ABC(2,k,j,i) + ABC(2,k,j-1,i) + ABC(2,k,j,i-1) + ABC(2,k,j-1,i-1));
        double Bsq_e3 = SQR(intBx_e3) + SQR(intBy_e3) + SQR(intBz_e3);
        double JdotB_e3 = intJ1_e3*intBx_e3 + intJ2_e3*intBy_e3 + intJ3_e3*intBz_e3;
        AE3(k,j,i) += eta * (Bsq_e3*intJ3_e3 - JdotB_e3*intBz_e3);
      }
    }
  }
#undef J1
#undef J2
#undef J3
  return 0;
}

/* Resistivity::AddFluxConstantAmbipolar, src/diffusion/ambipolar.cpp:254-494: Poynting flux of the
 * ambipolar field, E_amb ~ eta_ad*B^2*J on the edges around a face, added to the energy flux */
int akref_ambipolar_fluxes(const akmi_pack *p, double eta, const double *bcc0, const double *bx1f,
                           const double *bx2f, const double *bx3f, double *flx1, double *flx2,
                           double *flx3) {
  G g = mkG(p);
  if (!p->is_ideal) return AKMI_FAIL;
  const int nv = g.nvar, N1 = g.N1, N2 = g.N2, N3 = g.N3;
  const int is = g.is, ie = g.ie, js = g.js, je = g.je, ks = g.ks, ke = g.ke;
#define J1(k,j,i) edge_j1(&g, bx2f, bx3f, N1, N2, N3, m, k, j, i, dx2, dx3)
#define J2(k,j,i) edge_j2(&g, bx1f, bx3f, N1, N2, N3, m, k, j, i, dx1, dx3)
#define J3(k,j,i) edge_j3(&g, bx1f, bx2f, N1, N2, N3, m, k, j, i, dx1, dx2)
#define F1E(k,j,i) flx1[ix5(nv,N3,N2,N1+1,m,IEN,k,j,i)]
#define F2E(k,j,i) flx2[ix5(nv,N3,N2+1,N1,m,IEN,k,j,i)]
#define F3E(k,j,i) flx3[ix5(nv,N3+1,N2,N1,m,IEN,k,j,i)]
  for (int m = 0; m < g.nmb; ++m) {
    const double dx1 = p->dx[3*m], dx2 = p->dx[3*m+1], dx3 = p->dx[3*m+2];
    double Bx, By, Bz;
    if (!g.multi_d) {
      for (int i = is; i <= ie+1; ++i) {
        Bx = AB1(ks,js,i);
        By = 0.5*(ABC(1,ks,js,i-1) + ABC(1,ks,js,i));
        Bz = 0.5*(ABC(2,ks,js,i-1) + ABC(2,ks,js,i));
        double Bsq = SQR(Bx) + SQR(By) + SQR(Bz);
        double e2_fc = eta * Bsq * J2(ks,js,i);
        double e3_fc = eta * Bsq * J3(ks,js,i);
        F1E(ks,js,i) += e2_fc*Bz - e3_fc*By;
      }
      continue;
    }
    if (!g.three_d) {
      for (int j = js; j <= je; ++j) for (int i = is; i <= ie+1; ++i) {
        Bx = AB1(ks,j,i);
        By = 0.5*(ABC(1,ks,j,i-1) + ABC(1,ks,j,i));
        Bz = 0.5*(ABC(2,ks,j,i-1) + ABC(2,ks,j,i));
        double e2_fc = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J2(ks,j,i);
        Bx = 0.5*(AB1(ks,j,i) + AB1(ks,j-1,i));
        By = 0.5*(AB2(ks,j,i) + AB2(ks,j,i-1));
        Bz = 0.25*(ABC(2,ks,j,i) + ABC(2,ks,j-1,i) + ABC(2,ks,j,i-1) + ABC(2,ks,j-1,i-1));
        double e3_j = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J3(ks,j,i);
        Bx = 0.5*(AB1(ks,j+1,i) + AB1(ks,j,i));
        By = 0.5*(AB2(ks,j+1,i) + AB2(ks,j+1,i-1));
        Bz = 0.25*(ABC(2,ks,j+1,i) + ABC(2,ks,j,i) + ABC(2,ks,j+1,i-1) + ABC(2,ks,j,i-1));
        double e3_jp1 = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J3(ks,j+1,i);
        double e3_fc = 0.5*(e3_j + e3_jp1);
        double b2_fc = 0.5*(ABC(1,ks,j,i-1) + ABC(1,ks,j,i));
        double b3_fc = 0.5*(ABC(2,ks,j,i-1) + ABC(2,ks,j,i));
        F1E(ks,j,i) += e2_fc*b3_fc - e3_fc*b2_fc;
      }
      for (int j = js; j <= je+1; ++j) for (int i = is; i <= ie; ++i) {
        Bx = 0.5*(AB1(ks,j,i) + AB1(ks,j-1,i));
        By = 0.5*(AB2(ks,j,i) + AB2(ks,j,i-1));
        Bz = 0.25*(ABC(2,ks,j,i) + ABC(2,ks,j-1,i) + ABC(2,ks,j,i-1) + ABC(2,ks,j-1,i-1));
        double e3_i = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J3(ks,j,i);
        Bx = 0.5*(AB1(ks,j,i+1) + AB1(ks,j-1,i+1));
        By = 0.5*(AB2(ks,j,i+1) + AB2(ks,j,i));
        Bz = 0.25*(ABC(2,ks,j,i+1) + ABC(2,ks,j-1,i+1) + ABC(2,ks,j,i) + ABC(2,ks,j-1,i));
        double e3_ip1 = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J3(ks,j,i+1);
        double e3_fc = 0.5*(e3_i + e3_ip1);
        Bx = 0.5*(ABC(0,ks,j,i) + ABC(0,ks,j-1,i));
        By = AB2(ks,j,i);
        Bz = 0.5*(ABC(2,ks,j,i) + ABC(2,ks,j-1,i));
        double e1_fc = eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J1(ks,j,i);
        double b1_fc = 0.5*(ABC(0,ks,j-1,i) + ABC(0,ks,j,i));
        double b3_fc = 0.5*(ABC(2,ks,j-1,i) + ABC(2,ks,j,i));
        F2E(ks,j,i) += e3_fc*b1_fc - e1_fc*b3_fc;
      }
      continue;
    }
    /* 3-D: edge fields eta*B^2*J with B averaged to the edge as in the EMF routine */
#define E1EDGE(k,j,i) (Bx = 0.25*(ABC(0,k,j,i) + ABC(0,(k)-1,j,i) + ABC(0,k,(j)-1,i) + ABC(0,(k)-1,(j)-1,i)), \
                       By = 0.5*(AB2(k,j,i) + AB2((k)-1,j,i)), Bz = 0.5*(AB3(k,j,i) + AB3(k,(j)-1,i)), \
                       eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J1(k,j,i))
#define E2EDGE(k,j,i) (Bx = 0.5*(AB1(k,j,i) + AB1((k)-1,j,i)), \
                       By = 0.25*(ABC(1,k,j,i) + ABC(1,(k)-1,j,i) + ABC(1,k,j,(i)-1) + ABC(1,(k)-1,j,(i)-1)), \
                       Bz = 0.5*(AB3(k,j,i) + AB3(k,j,(i)-1)), eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J2(k,j,i))
#define E3EDGE(k,j,i) (Bx = 0.5*(AB1(k,j,i) + AB1(k,(j)-1,i)), By = 0.5*(AB2(k,j,i) + AB2(k,j,(i)-1)), \
                       Bz = 0.25*(ABC(2,k,j,i) + ABC(2,k,(j)-1,i) + ABC(2,k,j,(i)-1) + ABC(2,k,(j)-1,(i)-1)), \
                       eta * (SQR(Bx) + SQR(By) + SQR(Bz)) * J3(k,j,i))
    for (int k = ks; k <= ke; ++k) for (int j = js; j <= je; ++j) for (int i = is; i <= ie+1; ++i) {
      double e2_k = E2EDGE(k,j,i);
      double e2_kp1 = E2EDGE(k+1,j,i);
      double e2_fc = 0.5*(e2_k + e2_kp1);
      double e3_j = E3EDGE(k,j,i);
      double e3_jp1 = E3EDGE(k,j+1,i);
      double e3_fc = 0.5*(e3_j + e3_jp1);
      double b2_fc = 0.5*(ABC(1,k,j,i-1) + ABC(1,k,j,i));
      double b3_fc = 0.5*(ABC(2,k,j,i-1) + ABC(2,k,j,i));
      F1E(k,j,i) += e2_fc*b3_fc - e3_fc*b2_fc;
    }
    for (int k = ks; k <= ke; ++k) for (int j = js; j <= je+1; ++j) for (int i = is; i <= ie; ++i) {
      double e3_i = E3EDGE(k,j,i);
      double e3_ip1 = E3EDGE(k,j,i+1);
      double e3_fc = 0.5*(e3_i + e3_ip1);
      double e1_k = E1EDGE(k,j,i);
      double e1_kp1 = E1EDGE(k+1,j,i);
      double e1_fc = 0.5*(e1_k + e1_kp1);
      double b1_fc = 0.5*(ABC(0,k,j-1,i) + ABC(0,k,j,i));
      double b3_fc = 0.5*(ABC(2,k,j-1,i) + ABC(2,k,j,i));
      F2E(k,j,i) += e3_fc*b1_fc - e1_fc*b3_fc;
    }
    for (int k = ks; k <= ke+1; ++k) for (int j = js; j <= je; ++j) for (int i = is; i <= ie; ++i) {
      double e1_j = E1EDGE(k,j,i);
      double e1_jp1 = E1EDGE(k,j+1,i);
      double e1_fc = 0.5*(e1_j + e1_jp1);
      double e2_i = E2EDGE(k,j,i);
      double e2_ip1 = E2EDGE(k,j,i+1);
      double e2_fc = 0.5*(e2_i + e2_ip1);
      double b1_fc = 0.5*(ABC(0,k-1,j,i) + ABC(0,k,j,i));
      double b2_fc = 0.5*(ABC(1,k-1,j,i) + ABC(1,k,j,i));
      F3E(k,j,i) += e1_fc*b2_fc - e2_fc*b1_fc;
    }
#undef E1EDGE
#undef E2EDGE
#undef E3EDGE
  }
#undef J1
#undef J2
#undef J3
#undef F1E
#undef F2E
#undef F3E
  return 0;
}

/* Resistivity::NewTimeStep with eta_ad != 0, src/diffusion/resistivity.cpp:313-345: the cell reduction
 * min SQR(dx)/(eta_ohm + eta_ad*B^2) (before *fac) */
int akref_resistive_newdt(const akmi_pack *p, double eta_o, double eta_a, const double *bcc0,
                          double *dtmin) {
  G g = mkG(p);
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  double min_dt = (double)FLT_MAX;
  for (int m = 0; m < g.nmb; ++m)
    for (int k = g.ks; k <= g.ke; ++k) for (int j = g.js; j <= g.je; ++j)
      for (int i = g.is; i <= g.ie; ++i) {
        double eta = eta_o + eta_a*(SQR(ABC(0,k,j,i)) + SQR(ABC(1,k,j,i)) + SQR(ABC(2,k,j,i)));
        if (eta > 0.0) {
          min_dt = fmin(min_dt, SQR(p->dx[3*m])/eta);
          if (g.multi_d) min_dt = fmin(min_dt, SQR(p->dx[3*m+1])/eta);
          if (g.three_d) min_dt = fmin(min_dt, SQR(p->dx[3*m+2])/eta);
        }
      }
  *dtmin = min_dt;
  return 0;
}
#undef AB1
#undef AB2
#undef AB3
#undef ABC
#undef AE1
#undef AE2
#undef AE3


/* twins of the ABI's akmi_hydro_c2p_newdt / akmi_mhd_c2p_newdt (the product converts all cells and scans the CFL
 * condition in one kernel): the reference's two tasks in their order -- ConToPrim over all cells incl. ghost
 * zones (hydro_tasks.cpp:404-412, mhd_tasks.cpp:559-567), then NewTimeStep (hydro_newdt.cpp:30-139,
 * mhd_newdt.cpp:31-174) when do_newdt.  TEST INFRASTRUCTURE (tests/cpu_backend.py). */
int akref_hydro_c2p_newdt(const akmi_pack *p, double *u0, double *w0, int do_newdt, int *counters, double *dt3) {
  const int n1 = p->nx1 + 2*p->ng, n2 = p->nx2 > 1 ? p->nx2 + 2*p->ng : 1, n3 = p->nx3 > 1 ? p->nx3 + 2*p->ng : 1;
  akref_hydro_c2p(p, u0, w0, 0, n1 - 1, 0, n2 - 1, 0, n3 - 1, counters);
  if (do_newdt) akref_hydro_newdt(p, w0, dt3);
  return 0;
}
int akref_mhd_c2p_newdt(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f, const double *bx3f,
                        double *w0, double *bcc0, int do_newdt, int *counters, double *dt3) {
  const int n1 = p->nx1 + 2*p->ng, n2 = p->nx2 > 1 ? p->nx2 + 2*p->ng : 1, n3 = p->nx3 > 1 ? p->nx3 + 2*p->ng : 1;
  akref_mhd_c2p(p, u0, bx1f, bx2f, bx3f, w0, bcc0, 0, n1 - 1, 0, n2 - 1, 0, n3 - 1, counters);
  if (do_newdt) akref_mhd_newdt(p, w0, bcc0, dt3);
  return 0;
}
