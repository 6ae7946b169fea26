/* akref_sim.c -- CPU ORACLE (test infrastructure, see akref.h): mesh, problem generators
 * and the RK stage driver, restating the reference's control flow for uniform
 * (single-level) meshes with any number of MeshBlocks held in ONE pack on one process.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "akref.h"

#define SQR(x) ((x)*(x))
#define SIGN(x) ((x) < 0.0 ? -1.0 : 1.0)
enum { IDN = 0, IVX = 1, IVY = 2, IVZ = 3, IEN = 4 };

struct akref_sim {
  akref_params par;
  akmi_pack pack;
  int nmb, nb1, nb2, nb3;
  int N1, N2, N3, is, ie, js, je, ks, ke, multi_d, three_d;
  int *lloc;       /* [nmb][3] logical location of block gid (Z-ordered) */
  int *nghbr;      /* [nmb][27] */
  int *bcs;        /* [nmb][6] */
  double *dx;      /* [nmb][3] */
  double *xmin;    /* [nmb][6]: x1min,x1max,x2min,x2max,x3min,x3max */
  double *u0, *w0, *u1, *flx1, *flx2, *flx3;
  double *bcc0, *b0[3], *b1[3], *efc[6], *e[3];
  size_t ncc, nf[3], ne[3], nfl[3];
  double time, dt, dtnew, tlim;
  double dt_visc, dt_cond, dt_resist;   /* Viscosity/Conduction/Resistivity::dtnew */
  int nv;                          /* nhydro|nmhd: 5 ideal gas, 4 isothermal */
  int ncycle;
  int counters[3];
  unsigned char *fofc;             /* Hydro::fofc, src/hydro/hydro.hpp:116 */
  int nfofc;                       /* EventCounters::nfofc, src/mesh/mesh.hpp:71 */
  double gam0[4], gam1[4], beta[4], delta[4];
  /* statically refined meshes (akref_smr.c): levels, coarse buffers, level-aware boundary values */
  int multilevel, root_level;
  int *lev;                        /* [nmb] logical level of each block */
  akref_smr *smr;
  double *cu0, *cw0, *cb0[3];      /* coarse_u0, coarse_w0 (prolong_prims only), coarse_b0 */
  akmi_pack cpack;                 /* the coarse arrays seen as a pack of nx/2 cells (for the BCs) */
};

/* src/coordinates/cell_locations.hpp:23-39 */
static double LeftEdgeX(int ith, int n, double xmin, double xmax) {
  double x = ((double)ith)/((double)n);
  return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax);
}
static double CellCenterX(int ith, int n, double xmin, double xmax) {
  double x = ((double)ith + 0.5)/((double)n);
  return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax);
}

void akref_params_default(akref_params *p) {
  memset(p, 0, sizeof(*p));
  p->nx1 = 64; p->nx2 = 1; p->nx3 = 1; p->mb_nx1 = 64; p->mb_nx2 = 1; p->mb_nx3 = 1;
  p->ng = 2;
  p->x1min = p->x2min = p->x3min = -0.5; p->x1max = p->x2max = p->x3max = 0.5;
  p->nstages = 2; p->cfl = 0.3; p->tlim = 1.0; p->nlim = -1;
  p->recon = AKMI_RECON_PLM; p->rsolver = AKMI_RS_HLLC;
  p->gamma = 5.0/3.0;
  p->is_ideal = 1; p->iso_cs = 1.0; p->nscalars = 0;
  p->dfloor = p->pfloor = p->tfloor = p->sfloor = (double)FLT_MIN;  /* src/eos/eos.cpp:22-25 */
  p->sigma_max = (double)FLT_MAX;                                  /* src/eos/ideal_mhd.cpp:22 */
  p->dens = 1.0; p->pgas = 0.6; p->amp = 1e-6;
  p->shock_dir = 1;
  p->pi_amb = 1.0; p->di_amb = 1.0; p->prat = 1.0; p->drat = 1.0; p->b_amb = 0.1;
}

/* Z-order (Morton) index of a block: the reference walks its octree in Z-order with x1
 * varying fastest (src/mesh/meshblock_tree.cpp CreateZOrderedLLList). */
static unsigned long long morton(int x, int y, int z) {
  unsigned long long r = 0;
  for (int b = 0; b < 20; ++b) {
    r |= ((unsigned long long)((x >> b) & 1)) << (3*b);
    r |= ((unsigned long long)((y >> b) & 1)) << (3*b + 1);
    r |= ((unsigned long long)((z >> b) & 1)) << (3*b + 2);
  }
  return r;
}
typedef struct { unsigned long long key; int l[3]; } zent;
static int zcmp(const void *a, const void *b) {
  unsigned long long ka = ((const zent *)a)->key, kb = ((const zent *)b)->key;
  return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

static double *dalloc(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  if (!p) { fprintf(stderr, "akref: out of memory\n"); abort(); }
  return p;
}

akref_sim *akref_create(const akref_params *par) {
  akref_sim *s = (akref_sim *)calloc(1, sizeof(akref_sim));
  s->par = *par;
  const akref_params *p = &s->par;
  s->nb1 = p->nx1/p->mb_nx1; s->nb2 = p->nx2/p->mb_nx2; s->nb3 = p->nx3/p->mb_nx3;
  if (s->nb1*p->mb_nx1 != p->nx1 || s->nb2*p->mb_nx2 != p->nx2 || s->nb3*p->mb_nx3 != p->nx3) {
    free(s); return NULL;
  }
  if (p->eta_ad != 0.0 && !p->is_mhd) {
    free(s); return NULL;
  }
  /* rsolver = advect only for kinematic problems and vice versa (hydro.cpp:244-278, mhd.cpp:292-326) */
  if ((p->kinematic != 0) != (p->rsolver == AKMI_RS_ADVECT)) {
    free(s); return NULL;
  }
  /* <hydro>|<mhd>/fofc: src/hydro/hydro.cpp:153-190, src/mhd/mhd.cpp:199-235 (ghost-zone checks) */
  if (p->fofc && ((p->is_mhd && !p->is_ideal) || p->nscalars > 0 || (p->recon == AKMI_RECON_PLM && p->ng < 3) ||
                  (p->recon >= AKMI_RECON_PPM4 && p->ng < 4))) {
    free(s); return NULL;
  }
  s->nmb = s->nb1*s->nb2*s->nb3;
  s->multilevel = (p->smr_nmb > 0);
  s->root_level = p->smr_root_level;
  if (s->multilevel) s->nmb = p->smr_nmb;
  s->multi_d = p->nx2 > 1; s->three_d = p->nx3 > 1;
  const int ng = p->ng;
  s->N1 = p->mb_nx1 + 2*ng;
  s->N2 = s->multi_d ? p->mb_nx2 + 2*ng : 1;
  s->N3 = s->three_d ? p->mb_nx3 + 2*ng : 1;
  s->is = ng; s->ie = ng + p->mb_nx1 - 1;
  s->js = s->multi_d ? ng : 0; s->je = s->multi_d ? ng + p->mb_nx2 - 1 : 0;
  s->ks = s->three_d ? ng : 0; s->ke = s->three_d ? ng + p->mb_nx3 - 1 : 0;
  const int nmb = s->nmb;

  /* Z-ordered block list */
  s->lloc = (int *)malloc(sizeof(int)*3*nmb);
  s->lev = (int *)calloc(nmb, sizeof(int));
  int *gid_of = NULL;
  if (s->multilevel) {
    /* leaves of the MeshBlockTree in Z-order, handed over by the test (tree walk: the product's
     * host mirror of src/mesh/meshblock_tree.cpp); lloc = (lx1, lx2, lx3, level) per block */
    for (int m = 0; m < nmb; ++m) {
      for (int q = 0; q < 3; ++q) s->lloc[3*m+q] = p->smr_lloc[4*m+q];
      s->lev[m] = p->smr_lloc[4*m+3];
    }
  } else {
    zent *z = (zent *)malloc(sizeof(zent)*nmb);
    int c = 0;
    for (int l3 = 0; l3 < s->nb3; ++l3)
      for (int l2 = 0; l2 < s->nb2; ++l2)
        for (int l1 = 0; l1 < s->nb1; ++l1) {
          z[c].key = morton(l1, l2, l3); z[c].l[0] = l1; z[c].l[1] = l2; z[c].l[2] = l3; ++c;
        }
    qsort(z, nmb, sizeof(zent), zcmp);
    gid_of = (int *)malloc(sizeof(int)*nmb);
    for (int m = 0; m < nmb; ++m) {
      for (int q = 0; q < 3; ++q) s->lloc[3*m+q] = z[m].l[q];
      gid_of[(z[m].l[2]*s->nb2 + z[m].l[1])*s->nb1 + z[m].l[0]] = m;
    }
    free(z);
  }

  /* block sizes + BCs: src/mesh/meshblock.cpp:25-131 */
  s->dx = dalloc(3*nmb); s->xmin = dalloc(6*nmb);
  s->bcs = (int *)malloc(sizeof(int)*6*nmb);
  s->nghbr = (int *)malloc(sizeof(int)*27*nmb);
  const int nbr[3] = {s->nb1, s->nb2, s->nb3};
  const double mmin[3] = {p->x1min, p->x2min, p->x3min}, mmax[3] = {p->x1max, p->x2max, p->x3max};
  const int mbn[3] = {p->mb_nx1, p->mb_nx2, p->mb_nx3};
  for (int m = 0; m < nmb; ++m) {
    /* blocks per direction at this block's level: nmb_rootx << (level - root_level), meshblock.cpp:42 */
    int nb[3];
    for (int q = 0; q < 3; ++q) nb[q] = s->multilevel ? (nbr[q] << (s->lev[m] - s->root_level)) : nbr[q];
    for (int q = 0; q < 3; ++q) {
      int l = s->lloc[3*m+q];
      int active = (q == 0) || (q == 1 && s->multi_d) || (q == 2 && s->three_d);
      double lo, hi;
      if (!active || l == 0) { lo = mmin[q]; s->bcs[6*m+2*q] = p->bcs[2*q]; }
      else { lo = LeftEdgeX(l, nb[q], mmin[q], mmax[q]); s->bcs[6*m+2*q] = AKMI_BC_BLOCK; }
      if (!active || l == nb[q]-1) { hi = mmax[q]; s->bcs[6*m+2*q+1] = p->bcs[2*q+1]; }
      else { hi = LeftEdgeX(l+1, nb[q], mmin[q], mmax[q]); s->bcs[6*m+2*q+1] = AKMI_BC_BLOCK; }
      s->xmin[6*m+2*q] = lo; s->xmin[6*m+2*q+1] = hi;
      s->dx[3*m+q] = (hi - lo)/(double)mbn[q];
    }
    /* neighbours (same level): periodic wrap where the mesh BC is periodic */
    for (int d = 0; d < 27 && !s->multilevel; ++d) {
      int o[3] = {d%3 - 1, (d/3)%3 - 1, d/9 - 1};
      int ok = (d != 13), l[3];
      if (!s->multi_d && o[1]) ok = 0;
      if (!s->three_d && o[2]) ok = 0;
      for (int q = 0; q < 3 && ok; ++q) {
        l[q] = s->lloc[3*m+q] + o[q];
        if (l[q] < 0) { if (p->bcs[2*q] == AKMI_BC_PERIODIC) l[q] += nb[q]; else ok = 0; }
        else if (l[q] >= nb[q]) { if (p->bcs[2*q+1] == AKMI_BC_PERIODIC) l[q] -= nb[q]; else ok = 0; }
      }
      s->nghbr[27*m+d] = ok ? gid_of[(l[2]*s->nb2 + l[1])*s->nb1 + l[0]] : -1;
    }
  }
  free(gid_of);   /* NULL on a multilevel mesh */

  /* arrays: src/hydro/hydro.cpp:283-298, src/mhd/mhd.cpp:148-160,335-366 */
  const int N1 = s->N1, N2 = s->N2, N3 = s->N3;
  s->ncc = (size_t)nmb*N3*N2*N1;
  const int nv = s->nv = (p->is_ideal ? 5 : 4) + p->nscalars;   /* nhydro|nmhd + nscalars */
  s->u0 = dalloc(nv*s->ncc); s->w0 = dalloc(nv*s->ncc); s->u1 = dalloc(nv*s->ncc);
  s->nf[0] = (size_t)nmb*N3*N2*(N1+1); s->nf[1] = (size_t)nmb*N3*(N2+1)*N1;
  s->nf[2] = (size_t)nmb*(N3+1)*N2*N1;
  int fs = p->is_mhd ? 1 : 0;
  s->nfl[0] = (size_t)nmb*nv*N3*N2*(N1+fs); s->nfl[1] = (size_t)nmb*nv*N3*(N2+fs)*N1;
  s->nfl[2] = (size_t)nmb*nv*(N3+fs)*N2*N1;
  s->flx1 = dalloc(s->nfl[0]); s->flx2 = dalloc(s->nfl[1]); s->flx3 = dalloc(s->nfl[2]);
  if (p->fofc) s->fofc = (unsigned char *)calloc(s->ncc, 1);
  if (p->is_mhd) {
    s->bcc0 = dalloc(3*s->ncc);
    for (int q = 0; q < 3; ++q) { s->b0[q] = dalloc(s->nf[q]); s->b1[q] = dalloc(s->nf[q]); }
    for (int q = 0; q < 6; ++q) s->efc[q] = dalloc(s->ncc);
    s->ne[0] = (size_t)nmb*(N3+1)*(N2+1)*N1; s->ne[1] = (size_t)nmb*(N3+1)*N2*(N1+1);
    s->ne[2] = (size_t)nmb*N3*(N2+1)*(N1+1);
    for (int q = 0; q < 3; ++q) s->e[q] = dalloc(s->ne[q]);
  }
  s->pack.nmb = nmb; s->pack.nvar = nv;
  s->pack.iso_cs = p->iso_cs; s->pack.is_ideal = p->is_ideal;
  s->pack.nx1 = p->mb_nx1; s->pack.nx2 = p->mb_nx2; s->pack.nx3 = p->mb_nx3; s->pack.ng = ng;
  s->pack.dx = s->dx; s->pack.gamma = p->gamma;
  s->pack.dfloor = p->dfloor; s->pack.pfloor = p->pfloor; s->pack.tfloor = p->tfloor;
  s->pack.sfloor = p->sfloor; s->pack.sigma_max = p->sigma_max;
  if (s->multilevel) {
    /* coarse buffers (hydro.cpp:300-310, mhd.cpp:368-380) + MeshBoundaryValues with levels */
    s->smr = akref_smr_create(&s->pack, nv, p->smr_nghbr, s->lev, 1);
    s->cpack = s->pack;
    s->cpack.nx1 = p->mb_nx1/2;
    s->cpack.nx2 = s->multi_d ? p->mb_nx2/2 : 1;
    s->cpack.nx3 = s->three_d ? p->mb_nx3/2 : 1;
    const int cN1 = s->cpack.nx1 + 2*ng, cN2 = s->multi_d ? s->cpack.nx2 + 2*ng : 1,
              cN3 = s->three_d ? s->cpack.nx3 + 2*ng : 1;
    s->cu0 = dalloc((size_t)nmb*nv*cN3*cN2*cN1);
    s->cw0 = p->prolong_prims ? dalloc((size_t)nmb*nv*cN3*cN2*cN1) : NULL;
    if (p->is_mhd) {
      s->cb0[0] = dalloc((size_t)nmb*cN3*cN2*(cN1+1));
      s->cb0[1] = dalloc((size_t)nmb*cN3*(cN2+1)*cN1);
      s->cb0[2] = dalloc((size_t)nmb*(cN3+1)*cN2*cN1);
    }
  }

  /* RK weights, src/driver/driver.cpp:93-130 */
  if (p->nstages == 1) { s->gam0[0] = 0.0; s->gam1[0] = 1.0; s->beta[0] = 1.0; }
  else if (p->nstages == 2) {
    s->gam0[0] = 0.0; s->gam1[0] = 1.0; s->beta[0] = 1.0;
    s->gam0[1] = 0.5; s->gam1[1] = 0.5; s->beta[1] = 0.5;
  } else if (p->nstages == 3) {
    s->gam0[0] = 0.0; s->gam1[0] = 1.0; s->beta[0] = 1.0;
    s->gam0[1] = 0.25; s->gam1[1] = 0.75; s->beta[1] = 0.25;
    s->gam0[2] = 2.0/3.0; s->gam1[2] = 1.0/3.0; s->beta[2] = 2.0/3.0;
  } else {
    /* rk4 = RK4()4[2S], src/driver/driver.cpp:131-160; the second register is advanced in
     * Hydro::CopyCons only (MHD::CopyCons has no rk4 branch, src/mhd/mhd_tasks.cpp:162-170) */
    s->gam0[0] = 0.0; s->gam1[0] = 1.0; s->beta[0] = 1.193743905974738;
    s->gam0[1] = 0.121098479554482; s->gam1[1] = 0.721781678111411; s->beta[1] = 0.099279895495783;
    s->gam0[2] = -3.843833699660025; s->gam1[2] = 2.121209265338722; s->beta[2] = 1.131678018054042;
    s->gam0[3] = 0.546370891121863; s->gam1[3] = 0.198653035682705; s->beta[3] = 0.310665766509336;
    s->delta[0] = 1.0; s->delta[1] = 0.217683334308543; s->delta[2] = 1.065841341361089;
    s->delta[3] = 0.0;
  }
  s->time = 0.0; s->ncycle = 0; s->tlim = p->tlim;
  s->dt = (double)FLT_MAX;          /* src/mesh/build_tree.cpp:301 */
  return s;
}

void akref_destroy(akref_sim *s) {
  if (!s) return;
  free(s->lloc); free(s->nghbr); free(s->bcs); free(s->dx); free(s->xmin);
  free(s->u0); free(s->w0); free(s->u1); free(s->flx1); free(s->flx2); free(s->flx3);
  free(s->bcc0); free(s->fofc); free(s->lev); free(s->cu0); free(s->cw0);
  for (int q = 0; q < 3; ++q) free(s->cb0[q]);
  akref_smr_destroy(s->smr);
  for (int q = 0; q < 3; ++q) { free(s->b0[q]); free(s->b1[q]); free(s->e[q]); }
  for (int q = 0; q < 6; ++q) free(s->efc[q]);
  free(s);
}

#define IX5(m,n,k,j,i) (((((size_t)(m)*s->nv + (n))*s->N3 + (k))*s->N2 + (j))*s->N1 + (i))
#define IB3(m,n,k,j,i) (((((size_t)(m)*3 + (n))*s->N3 + (k))*s->N2 + (j))*s->N1 + (i))
#define IF1(m,k,j,i) ((((size_t)(m)*s->N3 + (k))*s->N2 + (j))*(s->N1+1) + (i))
#define IF2(m,k,j,i) ((((size_t)(m)*s->N3 + (k))*(s->N2+1) + (j))*s->N1 + (i))
#define IF3(m,k,j,i) ((((size_t)(m)*(s->N3+1) + (k))*s->N2 + (j))*s->N1 + (i))

/* ---------------------------------------------------------------------------------- */
/* PrimToCons: src/eos/ideal_c2p_hyd.hpp:76-83, src/eos/ideal_c2p_mhd.hpp:75-84 */
static void prim_to_cons(akref_sim *s, double *u) {
  for (int m = 0; m < s->nmb; ++m)
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double d = s->w0[IX5(m,IDN,k,j,i)], vx = s->w0[IX5(m,IVX,k,j,i)];
          double vy = s->w0[IX5(m,IVY,k,j,i)], vz = s->w0[IX5(m,IVZ,k,j,i)];
          u[IX5(m,IDN,k,j,i)] = d;
          u[IX5(m,IVX,k,j,i)] = d*vx;
          u[IX5(m,IVY,k,j,i)] = d*vy;
          u[IX5(m,IVZ,k,j,i)] = d*vz;
          for (int n = (s->par.is_ideal ? 5 : 4); n < s->nv; ++n)     /* scalars: d*s */
            u[IX5(m,n,k,j,i)] = d*s->w0[IX5(m,n,k,j,i)];
          if (!s->par.is_ideal) continue;       /* SingleP2C_Isothermal*: no energy */
          double e = s->w0[IX5(m,IEN,k,j,i)];
          if (s->par.is_mhd) {
            double bx = s->bcc0[IB3(m,0,k,j,i)], by = s->bcc0[IB3(m,1,k,j,i)];
            double bz = s->bcc0[IB3(m,2,k,j,i)];
            u[IX5(m,IEN,k,j,i)] = e + 0.5*(d*(SQR(vx) + SQR(vy) + SQR(vz)) +
                                           (SQR(bx) + SQR(by) + SQR(bz)));
          } else {
            u[IX5(m,IEN,k,j,i)] = e + 0.5*d*(SQR(vx) + SQR(vy) + SQR(vz));
          }
        }
}

/* ---- linear wave: src/pgen/tests/linear_wave.cpp ---------------------------------- */
typedef struct {
  double d0, p0, vx_0, vy_0, vz_0, bx_0, by_0, bz_0, dby, dbz, k_par;
  double cos_a2, cos_a3, sin_a2, sin_a3;
  int wave_flag;
} lwvars;

/* linear_wave.cpp:83-117 */
static double lwA1(double x1, double x2, double x3, const lwvars *lw) {
  double x = x1*lw->cos_a2*lw->cos_a3 + x2*lw->cos_a2*lw->sin_a3 + x3*lw->sin_a2;
  double y = -x1*lw->sin_a3 + x2*lw->cos_a3;
  double Ay = lw->bz_0*x - (lw->dbz/lw->k_par)*cos(lw->k_par*(x));
  double Az = -lw->by_0*x + (lw->dby/lw->k_par)*cos(lw->k_par*(x)) + lw->bx_0*y;
  return -Ay*lw->sin_a3 - Az*lw->sin_a2*lw->cos_a3;
}
static double lwA2(double x1, double x2, double x3, const lwvars *lw) {
  double x = x1*lw->cos_a2*lw->cos_a3 + x2*lw->cos_a2*lw->sin_a3 + x3*lw->sin_a2;
  double y = -x1*lw->sin_a3 + x2*lw->cos_a3;
  double Ay = lw->bz_0*x - (lw->dbz/lw->k_par)*cos(lw->k_par*(x));
  double Az = -lw->by_0*x + (lw->dby/lw->k_par)*cos(lw->k_par*(x)) + lw->bx_0*y;
  return Ay*lw->cos_a3 - Az*lw->sin_a2*lw->sin_a3;
}
static double lwA3(double x1, double x2, double x3, const lwvars *lw) {
  double x = x1*lw->cos_a2*lw->cos_a3 + x2*lw->cos_a2*lw->sin_a3 + x3*lw->sin_a2;
  double y = -x1*lw->sin_a3 + x2*lw->cos_a3;
  double Az = -lw->by_0*x + (lw->dby/lw->k_par)*cos(lw->k_par*(x)) + lw->bx_0*y;
  return Az*lw->cos_a2;
}

/* HydroEigensystemPrim, linear_wave.cpp:793-867 (ideal gas) */
static void hydro_eigen(double d, double v1, double p, double gamma, double ev[5],
                        double rem[5][5]) {
  double a = sqrt(gamma*p/d);
  ev[0] = v1 - a; ev[1] = v1; ev[2] = v1; ev[3] = v1; ev[4] = v1 + a;
  memset(rem, 0, sizeof(double)*25);
  rem[0][0] = 1.0; rem[1][0] = -a/d; rem[4][0] = a*a;
  rem[0][1] = 1.0;
  rem[2][2] = 1.0;
  rem[3][3] = 1.0;
  rem[0][4] = 1.0; rem[1][4] = a/d; rem[4][4] = a*a;
}

/* isothermal hydro, linear_wave.cpp:838-866: waves 0 (v-cs), 1, 2 (shear), 3 (v+cs) */
static void hydro_eigen_iso(double d, double v1, double cs, double ev[4], double rem[4][4]) {
  ev[0] = v1 - cs; ev[1] = v1; ev[2] = v1; ev[3] = v1 + cs;
  memset(rem, 0, sizeof(double)*16);
  rem[0][0] = 1.0; rem[1][0] = -cs/d;
  rem[2][1] = 1.0;
  rem[3][2] = 1.0;
  rem[0][3] = 1.0; rem[1][3] = cs/d;
}

/* isothermal MHD, linear_wave.cpp:1005-1098: rows d,vx,vy,vz,by,bz; 6 waves */
static void mhd_eigen_iso(double d, double v1, double b1, double b2, double b3, double y,
                          double iso_cs, double ev[6], double rem[6][6]) {
  double btsq = b2*b2 + b3*b3;
  double bt = sqrt(btsq);
  double bet2, bet3;
  if (bt == 0.0) { bet2 = 1.0; bet3 = 0.0; } else { bet2 = b2/bt; bet3 = b3/bt; }
  double bt_starsq = btsq*y;
  double vaxsq = b1*b1/d;
  double iso_cs2 = (iso_cs*iso_cs);
  double ct2 = bt_starsq/d;
  double tsum = vaxsq + ct2 + iso_cs2;
  double tdif = vaxsq + ct2 - iso_cs2;
  double cf2_cs2 = sqrt(tdif*tdif + 4.0*iso_cs2*ct2);
  double cfsq = 0.5*(tsum + cf2_cs2);
  double cf = sqrt(cfsq);
  double cssq = iso_cs2*vaxsq/cfsq;
  double cs = sqrt(cssq);
  double alpha_f, alpha_s;
  if ((cfsq - cssq) == 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else if ((iso_cs2 - cssq) <= 0.0) { alpha_f = 0.0; alpha_s = 1.0; }
  else if ((cfsq - iso_cs2) <= 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else { alpha_f = sqrt((iso_cs2 - cssq)/(cfsq - cssq)); alpha_s = sqrt((cfsq - iso_cs2)/(cfsq - cssq)); }
  double sqrtd = sqrt(d);
  double sg = SIGN(b1);
  double qf = cf*alpha_f*sg;
  double qs = cs*alpha_s*sg;
  double af = (iso_cs)*alpha_f*sqrtd;
  double as = (iso_cs)*alpha_s*sqrtd;
  double vax = sqrt(vaxsq);
  ev[0] = v1 - cf; ev[1] = v1 - vax; ev[2] = v1 - cs; ev[3] = v1 + cs; ev[4] = v1 + vax; ev[5] = v1 + cf;
  rem[0][0] = d*alpha_f; rem[1][0] = -cf*alpha_f; rem[2][0] = qs*bet2; rem[3][0] = qs*bet3;
  rem[4][0] = as*bet2; rem[5][0] = as*bet3;
  rem[0][1] = 0.0; rem[1][1] = 0.0; rem[2][1] = -bet3; rem[3][1] = bet2;
  rem[4][1] = -bet3*sg*sqrtd; rem[5][1] = bet2*sg*sqrtd;
  rem[0][2] = d*alpha_s; rem[1][2] = -cs*alpha_s; rem[2][2] = -qf*bet2; rem[3][2] = -qf*bet3;
  rem[4][2] = -af*bet2; rem[5][2] = -af*bet3;
  rem[0][3] = d*alpha_s; rem[1][3] = cs*alpha_s; rem[2][3] = qf*bet2; rem[3][3] = qf*bet3;
  rem[4][3] = rem[4][2]; rem[5][3] = rem[5][2];
  rem[0][4] = 0.0; rem[1][4] = 0.0; rem[2][4] = bet3; rem[3][4] = -bet2;
  rem[4][4] = rem[4][1]; rem[5][4] = rem[5][1];
  rem[0][5] = d*alpha_f; rem[1][5] = cf*alpha_f; rem[2][5] = -qs*bet2; rem[3][5] = -qs*bet3;
  rem[4][5] = rem[4][0]; rem[5][5] = rem[5][0];
}

/* MHDEigensystemPrim, linear_wave.cpp:876-1010 (ideal gas) */
static void mhd_eigen(double d, double v1, double p, double b1, double b2, double b3,
                      double xf, double y, double gamma, double ev[7], double rem[7][7]) {
  (void)xf;
  double btsq = b2*b2 + b3*b3;
  double bt = sqrt(btsq);
  double asq = (gamma*p/d);
  double bet2, bet3;
  if (bt == 0.0) { bet2 = 1.0; bet3 = 0.0; } else { bet2 = b2/bt; bet3 = b3/bt; }
  double gm1 = gamma - 1.0;
  double bt_starsq = (gm1 - (gm1 - 1.0)*y)*btsq;
  double vaxsq = b1*b1/d;
  double ct2 = bt_starsq/d;
  double tsum = vaxsq + ct2 + asq;
  double tdif = vaxsq + ct2 - asq;
  double cf2_cs2 = sqrt(tdif*tdif + 4.0*asq*ct2);
  double cfsq = 0.5*(tsum + cf2_cs2);
  double cf = sqrt(cfsq);
  double cssq = asq*vaxsq/cfsq;
  double cs = sqrt(cssq);
  double alpha_f, alpha_s;
  if ((cfsq - cssq) == 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else if ((asq - cssq) <= 0.0) { alpha_f = 0.0; alpha_s = 1.0; }
  else if ((cfsq - asq) <= 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else { alpha_f = sqrt((asq - cssq)/(cfsq - cssq)); alpha_s = sqrt((cfsq - asq)/(cfsq - cssq)); }
  double sqrtd = sqrt(d);
  double sg = SIGN(b1);
  double a = sqrt(asq);
  double qf = cf*alpha_f*sg;
  double qs = cs*alpha_s*sg;
  double af = a*alpha_f*sqrtd;
  double as = a*alpha_s*sqrtd;
  double vax = sqrt(vaxsq);
  ev[0] = v1 - cf; ev[1] = v1 - vax; ev[2] = v1 - cs; ev[3] = v1;
  ev[4] = v1 + cs; ev[5] = v1 + vax; ev[6] = v1 + cf;
  rem[0][0] = d*alpha_f; rem[0][1] = 0.0; rem[0][2] = d*alpha_s; rem[0][3] = 1.0;
  rem[0][4] = d*alpha_s; rem[0][5] = 0.0; rem[0][6] = d*alpha_f;
  rem[1][0] = -cf*alpha_f; rem[1][1] = 0.0; rem[1][2] = -cs*alpha_s; rem[1][3] = 0.0;
  rem[1][4] = cs*alpha_s; rem[1][5] = 0.0; rem[1][6] = cf*alpha_f;
  rem[2][0] = qs*bet2; rem[2][1] = -bet3; rem[2][2] = -qf*bet2; rem[2][3] = 0.0;
  rem[2][4] = qf*bet2; rem[2][5] = bet3; rem[2][6] = -qs*bet2;
  rem[3][0] = qs*bet3; rem[3][1] = bet2; rem[3][2] = -qf*bet3; rem[3][3] = 0.0;
  rem[3][4] = qf*bet3; rem[3][5] = -bet2; rem[3][6] = -qs*bet3;
  rem[4][0] = d*asq*alpha_f; rem[4][1] = 0.0; rem[4][2] = d*asq*alpha_s; rem[4][3] = 0.0;
  rem[4][4] = d*asq*alpha_s; rem[4][5] = 0.0; rem[4][6] = d*asq*alpha_f;
  rem[5][0] = as*bet2; rem[5][1] = -bet3*sg*sqrtd; rem[5][2] = -af*bet2; rem[5][3] = 0.0;
  rem[5][4] = rem[5][2]; rem[5][5] = rem[5][1]; rem[5][6] = rem[5][0];
  rem[6][0] = as*bet3; rem[6][1] = bet2*sg*sqrtd; rem[6][2] = -af*bet3; rem[6][3] = 0.0;
  rem[6][4] = rem[6][2]; rem[6][5] = rem[6][1]; rem[6][6] = rem[6][0];
}

/* ProblemGenerator::LinearWave, linear_wave.cpp:244-783.  set_ic!=0: fill u0/b0 and
 * rescale tlim (:384-394,520-530); set_ic==0: fill u1/b1 (reference solution). */
static void pgen_linear_wave(akref_sim *s, int set_ic) {
  const akref_params *p = &s->par;
  lwvars lw;
  double x1size = p->x1max - p->x1min, x2size = p->x2max - p->x2min, x3size = p->x3max - p->x3min;
  lw.cos_a3 = 1.0; lw.sin_a3 = 0.0; lw.cos_a2 = 1.0; lw.sin_a2 = 0.0;
  if (s->multi_d && !p->along_x1) {
    double ang_3 = atan(x1size/x2size);
    lw.sin_a3 = sin(ang_3); lw.cos_a3 = cos(ang_3);
  }
  if (s->three_d && !p->along_x1) {
    double ang_2 = atan(0.5*(x1size*lw.cos_a3 + x2size*lw.sin_a3)/x3size);
    lw.sin_a2 = sin(ang_2); lw.cos_a2 = cos(ang_2);
  }
  if (p->along_x2) { lw.cos_a3 = 0.0; lw.sin_a3 = 1.0; lw.cos_a2 = 1.0; lw.sin_a2 = 0.0; }
  if (p->along_x3) { lw.cos_a3 = 0.0; lw.sin_a3 = 1.0; lw.cos_a2 = 0.0; lw.sin_a2 = 1.0; }
  double lx = (double)FLT_MAX;
  if (lw.cos_a2*lw.cos_a3 > 0.0) lx = fmin(lx, x1size*lw.cos_a2*lw.cos_a3);
  if (lw.cos_a2*lw.sin_a3 > 0.0) lx = fmin(lx, x2size*lw.cos_a2*lw.sin_a3);
  if (lw.sin_a2 > 0.0) lx = fmin(lx, x3size*lw.sin_a2);
  lw.k_par = 2.0*(M_PI)/lx;
  lw.wave_flag = p->wave_flag;
  double amp = p->amp;
  lw.d0 = p->dens; lw.p0 = p->pgas; lw.vx_0 = p->vx0; lw.vy_0 = p->vy0; lw.vz_0 = p->vz0;
  lw.bx_0 = p->bx0; lw.by_0 = p->by0; lw.bz_0 = p->bz0; lw.dby = 0.0; lw.dbz = 0.0;
  const double gm1 = p->gamma - 1.0;
  double remh[5][5], evh[5], remm[7][7], evm[7];
  double r0, r1, r2, r3, r4 = 0.0, evw;
  if (!p->is_ideal && !p->is_mhd) {
    double r[4][4], e[4];
    hydro_eigen_iso(lw.d0, lw.vx_0, p->iso_cs, e, r);
    r0 = r[0][lw.wave_flag]; r1 = r[1][lw.wave_flag]; r2 = r[2][lw.wave_flag];
    r3 = r[3][lw.wave_flag]; evw = e[lw.wave_flag];
  } else if (!p->is_ideal) {
    double r[6][6], e[6];
    mhd_eigen_iso(lw.d0, lw.vx_0, lw.bx_0, lw.by_0, lw.bz_0, 1.0, p->iso_cs, e, r);
    lw.dby = amp*r[4][lw.wave_flag];           /* rem[nmhd_][wave], nmhd_ = 4 */
    lw.dbz = amp*r[5][lw.wave_flag];
    r0 = r[0][lw.wave_flag]; r1 = r[1][lw.wave_flag]; r2 = r[2][lw.wave_flag];
    r3 = r[3][lw.wave_flag]; evw = e[lw.wave_flag];
  } else if (!p->is_mhd) {
    hydro_eigen(lw.d0, lw.vx_0, lw.p0, p->gamma, evh, remh);
    r0 = remh[0][lw.wave_flag]; r1 = remh[1][lw.wave_flag]; r2 = remh[2][lw.wave_flag];
    r3 = remh[3][lw.wave_flag]; r4 = remh[4][lw.wave_flag]; evw = evh[lw.wave_flag];
  } else {
    mhd_eigen(lw.d0, lw.vx_0, lw.p0, lw.bx_0, lw.by_0, lw.bz_0, 0.0, 1.0, p->gamma, evm, remm);
    lw.dby = amp*remm[5][lw.wave_flag];
    lw.dbz = amp*remm[6][lw.wave_flag];
    r0 = remm[0][lw.wave_flag]; r1 = remm[1][lw.wave_flag]; r2 = remm[2][lw.wave_flag];
    r3 = remm[3][lw.wave_flag]; r4 = remm[4][lw.wave_flag]; evw = evm[lw.wave_flag];
  }
  if (set_ic) {
    /* the new time limit travels through ParameterInput::SetReal, i.e. `stringstream << Real`
     * = 6 significant digits (src/parameter_input.cpp:722-731), before the Driver reads it back:
     * 2.999999999997 -> 3, 1.4999999787 -> 1.5 */
    char buf[64];
    snprintf(buf, sizeof(buf), "%g", p->tlim*(fabs(lx/evw)));
    s->tlim = strtod(buf, NULL);
  }

  const int nx1 = p->mb_nx1, nx2 = p->mb_nx2, nx3 = p->mb_nx3;
  for (int m = 0; m < s->nmb; ++m) {
    const double *xm = &s->xmin[6*m];
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double x1v = CellCenterX(i - s->is, nx1, xm[0], xm[1]);
          double x2v = CellCenterX(j - s->js, nx2, xm[2], xm[3]);
          double x3v = CellCenterX(k - s->ks, nx3, xm[4], xm[5]);
          double x = lw.cos_a2*(x1v*lw.cos_a3 + x2v*lw.sin_a3) + x3v*lw.sin_a2;
          double sn = sin(lw.k_par*x);
          double rho = lw.d0 + amp*sn*r0;
          double vx = lw.vx_0 + amp*sn*r1;
          double vy = lw.vy_0 + amp*sn*r2;
          double vz = lw.vz_0 + amp*sn*r3;
          s->w0[IX5(m,IDN,k,j,i)] = rho;
          s->w0[IX5(m,IVX,k,j,i)] = vx*lw.cos_a2*lw.cos_a3 - vy*lw.sin_a3 - vz*lw.sin_a2*lw.cos_a3;
          s->w0[IX5(m,IVY,k,j,i)] = vx*lw.cos_a2*lw.sin_a3 + vy*lw.cos_a3 - vz*lw.sin_a2*lw.sin_a3;
          s->w0[IX5(m,IVZ,k,j,i)] = vx*lw.sin_a2 + vz*lw.cos_a2;
          if (p->is_ideal) s->w0[IX5(m,IEN,k,j,i)] = (lw.p0 + amp*sn*r4)/gm1;
        }
  }
  if (p->is_mhd) {
    /* vector potential on a (ncells3|2) x (ncells2|2) x ncells1 scratch, :532-567 */
    int nc1 = s->N1, nc2 = s->multi_d ? s->N2 : 2, nc3 = s->three_d ? s->N3 : 2;
    size_t na = (size_t)nc3*nc2*nc1;
    double *a1 = dalloc(na), *a2 = dalloc(na), *a3 = dalloc(na);
#define IA(k,j,i) (((size_t)(k)*nc2 + (j))*nc1 + (i))
    double **bb = set_ic ? s->b0 : s->b1;
    for (int m = 0; m < s->nmb; ++m) {
      const double *xm = &s->xmin[6*m];
      double dx1 = s->dx[3*m], dx2 = s->dx[3*m+1], dx3 = s->dx[3*m+2];
      for (int k = s->ks; k <= s->ke+1; ++k)
        for (int j = s->js; j <= s->je+1; ++j)
          for (int i = s->is; i <= s->ie+1; ++i) {
            double x1v = CellCenterX(i - s->is, nx1, xm[0], xm[1]);
            double x1f = LeftEdgeX(i - s->is, nx1, xm[0], xm[1]);
            double x2v = CellCenterX(j - s->js, nx2, xm[2], xm[3]);
            double x2f = LeftEdgeX(j - s->js, nx2, xm[2], xm[3]);
            double x3v = CellCenterX(k - s->ks, nx3, xm[4], xm[5]);
            double x3f = LeftEdgeX(k - s->ks, nx3, xm[4], xm[5]);
            a1[IA(k,j,i)] = lwA1(x1v, x2f, x3f, &lw);
            a2[IA(k,j,i)] = lwA2(x1f, x2v, x3f, &lw);
            a3[IA(k,j,i)] = lwA3(x1f, x2f, x3v, &lw);
            if (s->multilevel) {
              /* edges shared with a finer neighbour: the potential as the mean of the two fine-edge
               * values, so that the flux through shared fine/coarse faces is identical, :569-667 */
#define FNR(n) akref_smr_finer(s->smr, m, (n))
              const int is = s->is, ie = s->ie, js = s->js, je = s->je, ks = s->ks, ke = s->ke;
              const int x1lo = (FNR(0) || FNR(1) || FNR(2) || FNR(3)) && i == is;
              const int x1hi = (FNR(4) || FNR(5) || FNR(6) || FNR(7)) && i == ie+1;
              const int x2lo = nx2 > 1 && (FNR(8) || FNR(9) || FNR(10) || FNR(11)) && j == js;
              const int x2hi = nx2 > 1 && (FNR(12) || FNR(13) || FNR(14) || FNR(15)) && j == je+1;
              const int x3lo = nx3 > 1 && (FNR(24) || FNR(25) || FNR(26) || FNR(27)) && k == ks;
              const int x3hi = nx3 > 1 && (FNR(28) || FNR(29) || FNR(30) || FNR(31)) && k == ke+1;
              const int e12 = nx2 > 1 && (((FNR(16) || FNR(17)) && i == is && j == js) || ((FNR(18) || FNR(19)) && i == ie+1 && j == js) ||
                              ((FNR(20) || FNR(21)) && i == is && j == je+1) || ((FNR(22) || FNR(23)) && i == ie+1 && j == je+1));
              const int e31 = nx3 > 1 && (((FNR(32) || FNR(33)) && i == is && k == ks) || ((FNR(34) || FNR(35)) && i == ie+1 && k == ks) ||
                              ((FNR(36) || FNR(37)) && i == is && k == ke+1) || ((FNR(38) || FNR(39)) && i == ie+1 && k == ke+1));
              const int e23 = nx3 > 1 && (((FNR(40) || FNR(41)) && j == js && k == ks) || ((FNR(42) || FNR(43)) && j == je+1 && k == ks) ||
                              ((FNR(44) || FNR(45)) && j == js && k == ke+1) || ((FNR(46) || FNR(47)) && j == je+1 && k == ke+1));
#undef FNR
              if (x2lo || x2hi || x3lo || x3hi || e23) {
                double xl = x1v + 0.25*dx1, xr = x1v - 0.25*dx1;
                a1[IA(k,j,i)] = 0.5*(lwA1(xl, x2f, x3f, &lw) + lwA1(xr, x2f, x3f, &lw));
              }
              if (x1lo || x1hi || x3lo || x3hi || e31) {
                double xl = x2v + 0.25*dx2, xr = x2v - 0.25*dx2;
                a2[IA(k,j,i)] = 0.5*(lwA2(x1f, xl, x3f, &lw) + lwA2(x1f, xr, x3f, &lw));
              }
              if (x1lo || x1hi || x2lo || x2hi || e12) {
                double xl = x3v + 0.25*dx3, xr = x3v - 0.25*dx3;
                a3[IA(k,j,i)] = 0.5*(lwA3(x1f, x2f, xl, &lw) + lwA3(x1f, x2f, xr, &lw));
              }
            }
          }
      for (int k = s->ks; k <= s->ke; ++k)
        for (int j = s->js; j <= s->je; ++j)
          for (int i = s->is; i <= s->ie; ++i) {
            bb[0][IF1(m,k,j,i)] = (a3[IA(k,j+1,i)] - a3[IA(k,j,i)])/dx2 -
                                  (a2[IA(k+1,j,i)] - a2[IA(k,j,i)])/dx3;
            bb[1][IF2(m,k,j,i)] = (a1[IA(k+1,j,i)] - a1[IA(k,j,i)])/dx3 -
                                  (a3[IA(k,j,i+1)] - a3[IA(k,j,i)])/dx1;
            bb[2][IF3(m,k,j,i)] = (a2[IA(k,j,i+1)] - a2[IA(k,j,i)])/dx1 -
                                  (a1[IA(k,j+1,i)] - a1[IA(k,j,i)])/dx2;
            if (i == s->ie)
              bb[0][IF1(m,k,j,i+1)] = (a3[IA(k,j+1,i+1)] - a3[IA(k,j,i+1)])/dx2 -
                                      (a2[IA(k+1,j,i+1)] - a2[IA(k,j,i+1)])/dx3;
            if (j == s->je)
              bb[1][IF2(m,k,j+1,i)] = (a1[IA(k+1,j+1,i)] - a1[IA(k,j+1,i)])/dx3 -
                                      (a3[IA(k,j+1,i+1)] - a3[IA(k,j+1,i)])/dx1;
            if (k == s->ke)
              bb[2][IF3(m,k+1,j,i)] = (a2[IA(k+1,j,i+1)] - a2[IA(k+1,j,i)])/dx1 -
                                      (a1[IA(k+1,j+1,i)] - a1[IA(k+1,j,i)])/dx2;
          }
      for (int k = s->ks; k <= s->ke; ++k)
        for (int j = s->js; j <= s->je; ++j)
          for (int i = s->is; i <= s->ie; ++i) {
            s->bcc0[IB3(m,0,k,j,i)] = 0.5*(bb[0][IF1(m,k,j,i)] + bb[0][IF1(m,k,j,i+1)]);
            s->bcc0[IB3(m,1,k,j,i)] = 0.5*(bb[1][IF2(m,k,j,i)] + bb[1][IF2(m,k,j+1,i)]);
            s->bcc0[IB3(m,2,k,j,i)] = 0.5*(bb[2][IF3(m,k,j,i)] + bb[2][IF3(m,k+1,j,i)]);
          }
    }
#undef IA
    free(a1); free(a2); free(a3);
  }
  prim_to_cons(s, set_ic ? s->u0 : s->u1);
}

/* ProblemGenerator::ShockTube, src/pgen/tests/shock_tube.cpp:40-330 */
static void pgen_shock_tube(akref_sim *s) {
  const akref_params *p = &s->par;
  const int sd = p->shock_dir;
  const int ivx = sd, ivy = IVX + ((ivx - IVX) + 1)%3, ivz = IVX + ((ivx - IVX) + 2)%3;
  const double gm1 = p->gamma - 1.0;
  const int nxs[3] = {p->mb_nx1, p->mb_nx2, p->mb_nx3};
  for (int m = 0; m < s->nmb; ++m) {
    const double *xm = &s->xmin[6*m];
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double x;
          double bl[3], br[3];
          if (sd == 1) { x = CellCenterX(i - s->is, nxs[0], xm[0], xm[1]);
            bl[0] = p->wl[5]; bl[1] = p->wl[6]; bl[2] = p->wl[7];
            br[0] = p->wr[5]; br[1] = p->wr[6]; br[2] = p->wr[7]; }
          else if (sd == 2) { x = CellCenterX(j - s->js, nxs[1], xm[2], xm[3]);
            bl[0] = p->wl[7]; bl[1] = p->wl[5]; bl[2] = p->wl[6];
            br[0] = p->wr[7]; br[1] = p->wr[5]; br[2] = p->wr[6]; }
          else { x = CellCenterX(k - s->ks, nxs[2], xm[4], xm[5]);
            bl[0] = p->wl[6]; bl[1] = p->wl[7]; bl[2] = p->wl[5];
            br[0] = p->wr[6]; br[1] = p->wr[7]; br[2] = p->wr[5]; }
          const double *w = (x < p->xshock) ? p->wl : p->wr;
          const double *b = (x < p->xshock) ? bl : br;
          s->w0[IX5(m,IDN,k,j,i)] = w[0];
          s->w0[IX5(m,ivx,k,j,i)] = w[1]*1.0;
          s->w0[IX5(m,ivy,k,j,i)] = w[2]*1.0;
          s->w0[IX5(m,ivz,k,j,i)] = w[3]*1.0;
          if (p->is_ideal) s->w0[IX5(m,IEN,k,j,i)] = w[4]/gm1;
          if (p->is_mhd) {
            s->b0[0][IF1(m,k,j,i)] = b[0];
            s->b0[1][IF2(m,k,j,i)] = b[1];
            s->b0[2][IF3(m,k,j,i)] = b[2];
            if (i == s->ie) s->b0[0][IF1(m,k,j,i+1)] = b[0];
            if (j == s->je) s->b0[1][IF2(m,k,j+1,i)] = b[1];
            if (k == s->ke) s->b0[2][IF3(m,k+1,j,i)] = b[2];
            s->bcc0[IB3(m,0,k,j,i)] = b[0];
            s->bcc0[IB3(m,1,k,j,i)] = b[1];
            s->bcc0[IB3(m,2,k,j,i)] = b[2];
          }
        }
  }
  prim_to_cons(s, s->u0);
}

/* ProblemGenerator::OrszagTang, src/pgen/tests/orszag_tang.cpp:42-123 */
static double otA3(double x1, double x2, double B0) {
  return (B0/(4.0*M_PI))*(cos(4.0*M_PI*x1) - 2.0*cos(2.0*M_PI*x2));
}
static void pgen_orszag_tang(akref_sim *s) {
  const akref_params *p = &s->par;
  double B0 = 1.0/sqrt(4.0*M_PI);
  double d0 = 25.0/(36.0*M_PI);
  double v0 = 1.0;
  double p0 = 5.0/(12.0*M_PI);
  const double gm1 = p->gamma - 1.0;
  const int nx1 = p->mb_nx1, nx2 = p->mb_nx2;
  for (int m = 0; m < s->nmb; ++m) {
    const double *xm = &s->xmin[6*m];
    double dx1 = s->dx[3*m], dx2 = s->dx[3*m+1];
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double x1v = CellCenterX(i - s->is, nx1, xm[0], xm[1]);
          double x2v = CellCenterX(j - s->js, nx2, xm[2], xm[3]);
          s->u0[IX5(m,IDN,k,j,i)] = d0;
          s->u0[IX5(m,IVX,k,j,i)] = d0*v0*sin(2.0*M_PI*x2v);
          s->u0[IX5(m,IVY,k,j,i)] = -d0*v0*sin(2.0*M_PI*x1v);
          s->u0[IX5(m,IVZ,k,j,i)] = 0.0;
          double x1f = LeftEdgeX(i - s->is, nx1, xm[0], xm[1]);
          double x1fp1 = LeftEdgeX(i + 1 - s->is, nx1, xm[0], xm[1]);
          double x2f = LeftEdgeX(j - s->js, nx2, xm[2], xm[3]);
          double x2fp1 = LeftEdgeX(j + 1 - s->js, nx2, xm[2], xm[3]);
          s->b0[0][IF1(m,k,j,i)] = (otA3(x1f, x2fp1, B0) - otA3(x1f, x2f, B0))/dx2;
          s->b0[1][IF2(m,k,j,i)] = -(otA3(x1fp1, x2f, B0) - otA3(x1f, x2f, B0))/dx1;
          s->b0[2][IF3(m,k,j,i)] = 0.0;
          if (i == s->ie)
            s->b0[0][IF1(m,k,j,i+1)] = (otA3(x1fp1, x2fp1, B0) - otA3(x1fp1, x2f, B0))/dx2;
          if (j == s->je)
            s->b0[1][IF2(m,k,j+1,i)] = -(otA3(x1fp1, x2fp1, B0) - otA3(x1f, x2fp1, B0))/dx1;
          if (k == s->ke) s->b0[2][IF3(m,k+1,j,i)] = 0.0;
        }
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          s->u0[IX5(m,IEN,k,j,i)] = p0/gm1 + (0.5/s->u0[IX5(m,IDN,k,j,i)])*
              (SQR(s->u0[IX5(m,IVX,k,j,i)]) + SQR(s->u0[IX5(m,IVY,k,j,i)]) +
               SQR(s->u0[IX5(m,IVZ,k,j,i)])) +
              0.5*(SQR(0.5*(s->b0[0][IF1(m,k,j,i)] + s->b0[0][IF1(m,k,j,i+1)])) +
                   SQR(0.5*(s->b0[1][IF2(m,k,j,i)] + s->b0[1][IF2(m,k,j+1,i)])) +
                   SQR(0.5*(s->b0[2][IF3(m,k,j,i)] + s->b0[2][IF3(m,k+1,j,i)])));
        }
  }
}

/* blast: src/pgen/fluids/blast.cpp:134-392 (Cartesian, uniform-level mesh) */
static void pgen_blast(akref_sim *s) {
  const akref_params *p = &s->par;
  const double gm1 = p->gamma - 1.0;
  double rout = p->outer_radius;
  double rin = rout - p->inner_radius;
  const int nx1 = p->mb_nx1, nx2 = p->mb_nx2, nx3 = p->mb_nx3;
  for (int m = 0; m < s->nmb; ++m) {
    const double *xm = &s->xmin[6*m];
    double dx1 = s->dx[3*m], dx2 = s->dx[3*m+1];
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double x1v = CellCenterX(i - s->is, nx1, xm[0], xm[1]);
          double x2v = CellCenterX(j - s->js, nx2, xm[2], xm[3]);
          double x3v = CellCenterX(k - s->ks, nx3, xm[4], xm[5]);
          double rad = sqrt(SQR(x1v) + SQR(x2v) + SQR(x3v));
          double den = p->di_amb, pres = p->pi_amb;
          if (rad < rout) {
            if (rad < rin) { den *= p->drat; pres *= p->prat; }
            else {
              double f = (rad - rin)/(rout - rin);
              double log_den = (1.0 - f)*log(p->drat*p->di_amb) + f*log(p->di_amb);
              den = exp(log_den);
              double log_pres = (1.0 - f)*log(p->prat*p->pi_amb) + f*log(p->pi_amb);
              pres = exp(log_pres);
            }
          }
          s->w0[IX5(m,IDN,k,j,i)] = den;
          s->w0[IX5(m,IVX,k,j,i)] = 0.0;
          s->w0[IX5(m,IVY,k,j,i)] = 0.0;
          s->w0[IX5(m,IVZ,k,j,i)] = 0.0;
          s->w0[IX5(m,IEN,k,j,i)] = pres/gm1;
        }
    if (p->is_mhd) {
      /* a3 = b_amb*x2f (blast.cpp:355); faces from curl (:362-375) */
      for (int k = s->ks; k <= s->ke; ++k)
        for (int j = s->js; j <= s->je; ++j)
          for (int i = s->is; i <= s->ie; ++i) {
            double a3_j = p->b_amb*LeftEdgeX(j - s->js, nx2, xm[2], xm[3]);
            double a3_jp = p->b_amb*LeftEdgeX(j + 1 - s->js, nx2, xm[2], xm[3]);
            s->b0[0][IF1(m,k,j,i)] = (a3_jp - a3_j)/dx2;
            s->b0[1][IF2(m,k,j,i)] = -(a3_j - a3_j)/dx1;
            s->b0[2][IF3(m,k,j,i)] = 0.0;
            if (i == s->ie) s->b0[0][IF1(m,k,j,i+1)] = (a3_jp - a3_j)/dx2;
            if (j == s->je) s->b0[1][IF2(m,k,j+1,i)] = -(a3_jp - a3_jp)/dx1;
            if (k == s->ke) s->b0[2][IF3(m,k+1,j,i)] = 0.0;
          }
      for (int k = s->ks; k <= s->ke; ++k)
        for (int j = s->js; j <= s->je; ++j)
          for (int i = s->is; i <= s->ie; ++i) {
            s->bcc0[IB3(m,0,k,j,i)] = 0.5*(s->b0[0][IF1(m,k,j,i)] + s->b0[0][IF1(m,k,j,i+1)]);
            s->bcc0[IB3(m,1,k,j,i)] = 0.5*(s->b0[1][IF2(m,k,j,i)] + s->b0[1][IF2(m,k,j+1,i)]);
            s->bcc0[IB3(m,2,k,j,i)] = 0.5*(s->b0[2][IF3(m,k,j,i)] + s->b0[2][IF3(m,k+1,j,i)]);
          }
    }
  }
  prim_to_cons(s, s->u0);
}

/* ---------------------------------------------------------------------------------- */
static int strictly_periodic(const akref_sim *s) {
  const akref_params *p = &s->par;
  if (p->bcs[0] != AKMI_BC_PERIODIC || p->bcs[1] != AKMI_BC_PERIODIC) return 0;
  if (s->multi_d && (p->bcs[2] != AKMI_BC_PERIODIC || p->bcs[3] != AKMI_BC_PERIODIC)) return 0;
  if (s->three_d && (p->bcs[4] != AKMI_BC_PERIODIC || p->bcs[5] != AKMI_BC_PERIODIC)) return 0;
  return 1;
}

/* SendU/RecvU, SendB/RecvB, ApplyPhysicalBCs, ConToPrim over all cells incl. ghosts
 * (src/hydro/hydro_tasks.cpp:308-320,357-412; src/mhd/mhd_tasks.cpp:478-520) */
/* the same chain on a multilevel mesh: RestrictU, SendU, RecvU, (RestrictB, SendB, RecvB,) Prolongate,
 * ApplyPhysicalBCs (hydro_tasks.cpp:64-70,291-400; mhd_tasks.cpp:59-72,312-552).  with_u = 0: only the
 * field part (the U exchange of an MHD stage happens before CornerE, mhd_tasks.cpp:59-61) */
static void smr_exchange_u(akref_sim *s) {
  akref_restrict_cc(&s->pack, s->nv, s->u0, s->cu0);
  akref_smr_send_cc(s->smr, s->u0, s->cu0);
  akref_smr_recv_cc(s->smr, s->u0, s->cu0);
}
static void smr_exchange_b(akref_sim *s) {
  akref_restrict_fc(&s->pack, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
  akref_smr_send_fc(s->smr, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
  akref_smr_recv_fc(s->smr, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
}
static void smr_prolongate(akref_sim *s) {
  const int mhd = s->par.is_mhd;
  akref_smr_fill_coarse_cc(s->smr, s->u0, s->cu0);
  if (mhd) akref_smr_fill_coarse_fc(s->smr, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
  if (!strictly_periodic(s)) {            /* HydroBCsCoarse / BFieldBCsCoarse: the BC helpers on coarse indices */
    akref_hydro_bcs(&s->cpack, s->nv, s->bcs, s->cu0);
    if (mhd) akref_bfield_bcs(&s->cpack, s->bcs, s->cb0[0], s->cb0[1], s->cb0[2]);
  }
  if (s->par.prolong_prims) {             /* mhd_tasks.cpp:539-544, hydro_tasks.cpp:388-392 */
    akref_smr_c2p_coarse(s->smr, &s->pack, s->cu0, mhd ? s->cb0[0] : NULL, mhd ? s->cb0[1] : NULL,
                         mhd ? s->cb0[2] : NULL, s->cw0);
    akref_smr_prolong_cc(s->smr, s->w0, s->cw0);
    if (mhd) akref_smr_prolong_fc(s->smr, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
    akref_smr_p2c_fine(s->smr, s->w0, mhd ? s->b0[0] : NULL, mhd ? s->b0[1] : NULL, mhd ? s->b0[2] : NULL, s->u0);
    return;
  }
  akref_smr_prolong_cc(s->smr, s->u0, s->cu0);
  if (mhd) akref_smr_prolong_fc(s->smr, s->b0[0], s->b0[1], s->b0[2], s->cb0[0], s->cb0[1], s->cb0[2]);
}

static void halo_bcs_c2p(akref_sim *s) {
  const akmi_pack *pk = &s->pack;
  if (s->multilevel) {
    /* Driver::InitBoundaryValuesAndPrimitives order (driver.cpp:586-630); inside a stage the U part
     * has already run when u_done is set by akref_step */
    smr_exchange_u(s);
    if (s->par.is_mhd) smr_exchange_b(s);
    smr_prolongate(s);
  } else {
    akref_bvals_cc_local(pk, s->nv, s->nghbr, s->u0);
    if (s->par.is_mhd) akref_bvals_fc_local(pk, s->nghbr, s->b0[0], s->b0[1], s->b0[2]);
  }
  if (!strictly_periodic(s)) {
    akref_hydro_bcs(pk, s->nv, s->bcs, s->u0);
    if (s->par.is_mhd) akref_bfield_bcs(pk, s->bcs, s->b0[0], s->b0[1], s->b0[2]);
  }
  if (s->par.is_mhd)
    akref_mhd_c2p(pk, s->u0, s->b0[0], s->b0[1], s->b0[2], s->w0, s->bcc0, 0, s->N1-1, 0,
                  s->N2-1, 0, s->N3-1, s->counters);
  else
    akref_hydro_c2p(pk, s->u0, s->w0, 0, s->N1-1, 0, s->N2-1, 0, s->N3-1, s->counters);
}

static void new_dt_task(akref_sim *s) {
  double d3[3];
  if (s->par.kinematic) akref_kinematic_newdt(&s->pack, s->w0, d3);
  else if (s->par.is_mhd) akref_mhd_newdt(&s->pack, s->w0, s->bcc0, d3);
  else akref_hydro_newdt(&s->pack, s->w0, d3);
  double dtnew = d3[0];
  if (s->multi_d) dtnew = fmin(dtnew, d3[1]);
  if (s->three_d) dtnew = fmin(dtnew, d3[2]);
  s->dtnew = dtnew;
  /* diffusive limits: viscosity.cpp:232-251, conduction.cpp:314-377, resistivity.cpp:291-311 */
  const akref_params *p = &s->par;
  const double fac = s->three_d ? 1.0/6.0 : (s->multi_d ? 0.25 : 0.5);
  s->dt_visc = s->dt_cond = s->dt_resist = (double)FLT_MAX;
  for (int m = 0; m < s->nmb; ++m) {
    const double *dx = s->dx + 3*m;
    if (p->nu_iso != 0.0) {
      s->dt_visc = fmin(s->dt_visc, fac*(dx[0]*dx[0])/p->nu_iso);
      if (s->multi_d) s->dt_visc = fmin(s->dt_visc, fac*(dx[1]*dx[1])/p->nu_iso);
      if (s->three_d) s->dt_visc = fmin(s->dt_visc, fac*(dx[2]*dx[2])/p->nu_iso);
    }
    if (p->is_mhd && p->eta_ad == 0.0 && p->eta_ohm > 0.0) {          /* resistivity.cpp:299-311 */
      s->dt_resist = fmin(s->dt_resist, fac*(dx[0]*dx[0])/p->eta_ohm);
      if (s->multi_d) s->dt_resist = fmin(s->dt_resist, fac*(dx[1]*dx[1])/p->eta_ohm);
      if (s->three_d) s->dt_resist = fmin(s->dt_resist, fac*(dx[2]*dx[2])/p->eta_ohm);
    }
  }
  if (p->is_mhd && p->eta_ad != 0.0) {                                 /* resistivity.cpp:313-345 */
    akref_resistive_newdt(&s->pack, p->eta_ohm, p->eta_ad, s->bcc0, &s->dt_resist);
    s->dt_resist *= fac;
  }
  if (p->alpha_iso != 0.0) {
    akref_conduction_newdt(&s->pack, p->alpha_iso, s->w0, &s->dt_cond);
    s->dt_cond *= fac;
  }
}

/* Mesh::NewTimeStep, src/mesh/mesh.cpp:573-643 */
static void mesh_new_dt(akref_sim *s) {
  s->dt = 2.0*s->dt;
  s->dt = fmin(s->dt, s->par.cfl*s->dtnew);
  if (s->par.nu_iso != 0.0) s->dt = fmin(s->dt, s->par.cfl*s->dt_visc);
  if (s->par.is_mhd && (s->par.eta_ohm != 0.0 || s->par.eta_ad != 0.0)) s->dt = fmin(s->dt, s->par.cfl*s->dt_resist);
  if (s->par.alpha_iso != 0.0) s->dt = fmin(s->dt, s->par.cfl*s->dt_cond);
  if ((s->time < s->tlim) && ((s->time + s->dt) > s->tlim)) s->dt = s->tlim - s->time;
}

void akref_initialize(akref_sim *s) {
  switch (s->par.pgen) {
    case AKREF_PGEN_LINEAR_WAVE: pgen_linear_wave(s, 1); break;
    case AKREF_PGEN_SHOCK_TUBE: pgen_shock_tube(s); break;
    case AKREF_PGEN_ORSZAG_TANG: pgen_orszag_tang(s); break;
    case AKREF_PGEN_BLAST: pgen_blast(s); break;
  }
  /* Driver::Initialize -> InitBoundaryValuesAndPrimitives + NewTimeStep
   * (src/driver/driver.cpp:314-371,569-653) */
  halo_bcs_c2p(s);
  new_dt_task(s);
  mesh_new_dt(s);
}

/* Driver::Initialize again on state a test wrote into u0/b0 after akref_initialize (e.g. passive
 * scalars, which none of the problem generators of this path sets) */
void akref_reinitialize(akref_sim *s) {
  s->dt = (double)FLT_MAX;
  halo_bcs_c2p(s);
  new_dt_task(s);
  mesh_new_dt(s);
}

int akref_step(akref_sim *s) {
  const akref_params *p = &s->par;
  if (!(s->time < s->tlim && (s->ncycle < p->nlim || p->nlim < 0))) return 0;
  const akmi_pack *pk = &s->pack;
  for (int stage = 1; stage <= p->nstages; ++stage) {
    double gam0 = s->gam0[stage-1], gam1 = s->gam1[stage-1];
    double beta_dt = s->beta[stage-1]*s->dt;
    if (p->is_mhd) {
      /* stagen chain, src/mhd/mhd_tasks.cpp:48-75 */
      if (stage == 1) {
        akref_copy_cons(pk, s->u0, s->u1);
        for (int q = 0; q < 3; ++q) memcpy(s->b1[q], s->b0[q], sizeof(double)*s->nf[q]);
      }
      if (p->fofc)
        akref_mhd_fluxes_fofc(pk, p->recon, p->rsolver, s->w0, s->bcc0, s->b0[0], s->b0[1], s->b0[2],
                              s->flx1, s->flx2, s->flx3, s->efc[0], s->efc[1], s->efc[2], s->efc[3],
                              s->efc[4], s->efc[5]);
      else
        akref_mhd_fluxes(pk, p->recon, p->rsolver, s->w0, s->bcc0, s->b0[0], s->b0[1], s->b0[2],
                         s->flx1, s->flx2, s->flx3, s->efc[0], s->efc[1], s->efc[2], s->efc[3],
                         s->efc[4], s->efc[5]);
      /* diffusion fluxes, then FOFC: mhd_tasks.cpp:198-211 */
      if (p->alpha_iso != 0.0) akref_heat_fluxes(pk, p->alpha_iso, s->w0, s->flx1, s->flx2, s->flx3, 1);
      if (p->nu_iso != 0.0) akref_viscous_fluxes(pk, p->nu_iso, s->w0, s->flx1, s->flx2, s->flx3, 1);
      if (p->eta_ohm != 0.0 && p->is_ideal)
        akref_resistive_fluxes(pk, p->eta_ohm, s->b0[0], s->b0[1], s->b0[2], s->flx1, s->flx2, s->flx3);
      if (p->eta_ad != 0.0 && p->is_ideal)                       /* resistivity.cpp:67-69 */
        akref_ambipolar_fluxes(pk, p->eta_ad, s->bcc0, s->b0[0], s->b0[1], s->b0[2], s->flx1, s->flx2, s->flx3);
      if (p->fofc) {               /* mhd_tasks.cpp:209-211 */
        akref_mhd_fofc(pk, gam0, gam1, beta_dt, s->w0, s->bcc0, s->b0[0], s->b0[1], s->b0[2],
                       s->b1[0], s->b1[1], s->b1[2], s->u0, s->u1, s->flx1, s->flx2, s->flx3,
                       s->efc[0], s->efc[1], s->efc[2], s->efc[3], s->efc[4], s->efc[5], s->fofc,
                       &s->nfofc);
      }
      if (s->multilevel) akref_smr_flux_cc(s->smr, s->flx1, s->flx2, s->flx3, 1);   /* SendFlux/RecvFlux */
      akref_rk_update(pk, gam0, gam1, beta_dt, s->u0, s->u1, s->flx1, s->flx2, s->flx3, 1);
      akref_mhd_corner_e(pk, s->w0, s->bcc0, s->efc[0], s->efc[1], s->efc[2], s->efc[3],
                         s->efc[4], s->efc[5], s->flx1, s->flx2, s->flx3, s->e[0], s->e[1],
                         s->e[2]);
      if (p->eta_ohm != 0.0)         /* MHD::EField, mhd_tasks.cpp:381-383 */
        akref_resistive_emfs(pk, p->eta_ohm, s->b0[0], s->b0[1], s->b0[2], s->e[0], s->e[1], s->e[2]);
      if (p->eta_ad != 0.0)          /* resistivity.cpp:52-54 */
        akref_ambipolar_emfs(pk, p->eta_ad, s->bcc0, s->b0[0], s->b0[1], s->b0[2], s->e[0], s->e[1], s->e[2]);
      /* SendE/RecvE: on a uniform mesh every shared edge EMF is computed identically by
       * both owners; sum and average return the value itself (2a/2, ((2a+a)+a)/4 are exact), so the
       * exchange is skipped there.  With levels it is the flux correction of the field. */
      if (s->multilevel) akref_smr_flux_fc(s->smr, s->e[0], s->e[1], s->e[2]);
      akref_mhd_ct(pk, gam0, gam1, beta_dt, s->e[0], s->e[1], s->e[2], s->b0[0], s->b0[1],
                   s->b0[2], s->b1[0], s->b1[1], s->b1[2]);
    } else {
      /* stagen chain, src/hydro/hydro_tasks.cpp:55-71 */
      if (stage == 1) akref_copy_cons(pk, s->u0, s->u1);
      else if (p->nstages == 4) akref_rk4_copy_cons(pk, s->delta[stage-1], s->u0, s->u1);
      if (p->fofc) akref_hydro_fluxes_fofc(pk, p->recon, p->rsolver, s->w0, s->flx1, s->flx2, s->flx3, 0);
      else akref_hydro_fluxes(pk, p->recon, p->rsolver, s->w0, s->flx1, s->flx2, s->flx3, 0);
      /* diffusion fluxes, then FOFC: hydro_tasks.cpp:183-199 */
      if (p->alpha_iso != 0.0) akref_heat_fluxes(pk, p->alpha_iso, s->w0, s->flx1, s->flx2, s->flx3, 0);
      if (p->nu_iso != 0.0) akref_viscous_fluxes(pk, p->nu_iso, s->w0, s->flx1, s->flx2, s->flx3, 0);
      if (p->fofc)
        akref_hydro_fofc(pk, gam0, gam1, beta_dt, s->w0, s->u0, s->u1, s->flx1, s->flx2, s->flx3, 0,
                         s->fofc, &s->nfofc);
      if (s->multilevel) akref_smr_flux_cc(s->smr, s->flx1, s->flx2, s->flx3, 0);   /* SendFlux/RecvFlux */
      akref_rk_update(pk, gam0, gam1, beta_dt, s->u0, s->u1, s->flx1, s->flx2, s->flx3, 0);
    }
    halo_bcs_c2p(s);
    if (stage == p->nstages) new_dt_task(s);
  }
  s->time = s->time + s->dt;
  s->ncycle++;
  mesh_new_dt(s);
  return 1;
}

int akref_run(akref_sim *s) {
  int n = 0;
  while (akref_step(s)) ++n;
  return n;
}

double akref_time(const akref_sim *s) { return s->time; }
double akref_dt(const akref_sim *s) { return s->dt; }
double akref_tlim(const akref_sim *s) { return s->tlim; }
int akref_ncycle(const akref_sim *s) { return s->ncycle; }
int akref_nmb(const akref_sim *s) { return s->nmb; }
int akref_nfofc(const akref_sim *s) { return s->nfofc; }
void akref_pack(const akref_sim *s, akmi_pack *out) { *out = s->pack; }

void *akref_array(akref_sim *s, const char *name, long long *count) {
  struct { const char *n; void *p; size_t c; } tab[] = {
    {"u0", s->u0, s->nv*s->ncc}, {"w0", s->w0, s->nv*s->ncc}, {"u1", s->u1, s->nv*s->ncc},
    {"bcc0", s->bcc0, 3*s->ncc},
    {"b0x1f", s->b0[0], s->nf[0]}, {"b0x2f", s->b0[1], s->nf[1]}, {"b0x3f", s->b0[2], s->nf[2]},
    {"b1x1f", s->b1[0], s->nf[0]}, {"b1x2f", s->b1[1], s->nf[1]}, {"b1x3f", s->b1[2], s->nf[2]},
    {"flx1", s->flx1, s->nfl[0]}, {"flx2", s->flx2, s->nfl[1]}, {"flx3", s->flx3, s->nfl[2]},
    {"e3x1", s->efc[0], s->ncc}, {"e2x1", s->efc[1], s->ncc}, {"e1x2", s->efc[2], s->ncc},
    {"e3x2", s->efc[3], s->ncc}, {"e2x3", s->efc[4], s->ncc}, {"e1x3", s->efc[5], s->ncc},
    {"e1", s->e[0], s->ne[0]}, {"e2", s->e[1], s->ne[1]}, {"e3", s->e[2], s->ne[2]},
    {"dx", s->dx, (size_t)3*s->nmb}, {"xminmax", s->xmin, (size_t)6*s->nmb},
    {"nghbr", s->nghbr, (size_t)27*s->nmb}, {"bcs", s->bcs, (size_t)6*s->nmb},
    {"lloc", s->lloc, (size_t)3*s->nmb}, {"counters", s->counters, 3},
  };
  for (size_t t = 0; t < sizeof(tab)/sizeof(tab[0]); ++t)
    if (!strcmp(tab[t].n, name)) { if (count) *count = (long long)tab[t].c; return tab[t].p; }
  if (count) *count = 0;
  return NULL;
}

/* LinearWaveErrors + OutputErrors: src/pgen/tests/linear_wave.cpp:1430-1437,
 * src/pgen/pgen.cpp:680-900 */
int akref_linear_wave_errors(akref_sim *s, double *out) {
  pgen_linear_wave(s, 0);
  const int nf = s->par.is_ideal ? 5 : 4;     /* pgen.cpp:756-766: bindx = nmhd */
  int nvars = s->par.is_mhd ? nf + 3 : nf;
  double l1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double linf = 0.0;
  for (int m = 0; m < s->nmb; ++m) {
    double vol = s->dx[3*m]*s->dx[3*m+1]*s->dx[3*m+2];
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double ev[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          for (int n = 0; n < nf; ++n) {
            ev[n] = vol*fabs(s->u0[IX5(m,n,k,j,i)] - s->u1[IX5(m,n,k,j,i)]);
            linf = fmax(linf, ev[n]);
          }
          if (s->par.is_mhd) {
            double a = 0.5*(s->b0[0][IF1(m,k,j,i)] + s->b0[0][IF1(m,k,j,i+1)]);
            double b = 0.5*(s->b1[0][IF1(m,k,j,i)] + s->b1[0][IF1(m,k,j,i+1)]);
            ev[nf] = vol*fabs(a - b);
            a = 0.5*(s->b0[1][IF2(m,k,j,i)] + s->b0[1][IF2(m,k,j+1,i)]);
            b = 0.5*(s->b1[1][IF2(m,k,j,i)] + s->b1[1][IF2(m,k,j+1,i)]);
            ev[nf+1] = vol*fabs(a - b);
            a = 0.5*(s->b0[2][IF3(m,k,j,i)] + s->b0[2][IF3(m,k+1,j,i)]);
            b = 0.5*(s->b1[2][IF3(m,k,j,i)] + s->b1[2][IF3(m,k+1,j,i)]);
            ev[nf+2] = vol*fabs(a - b);
            /* pgen.cpp:793-805 takes the maxima from slots IEN+1..IEN+3 = 5,6,7 whatever bindx is */
            linf = fmax(linf, fmax(ev[5], fmax(ev[6], ev[7])));
          }
          for (int n = 0; n < nvars; ++n) l1[n] += ev[n];
        }
  }
  const akref_params *p = &s->par;
  double vol = (p->x1max - p->x1min)*(p->x2max - p->x2min)*(p->x3max - p->x3min);
  double rms = 0.0;
  for (int n = 0; n < nvars; ++n) { l1[n] = l1[n]/vol; rms += SQR(l1[n]); }
  out[0] = sqrt(rms);
  out[1] = linf/vol;
  for (int n = 0; n < nvars; ++n) out[2+n] = l1[n];
  return 2 + nvars;
}

void akref_divb(akref_sim *s, double *out) {
  double mx = 0.0, sum = 0.0; size_t cnt = 0;
  out[0] = out[1] = 0.0;
  if (!s->par.is_mhd) return;
  for (int m = 0; m < s->nmb; ++m)
    for (int k = s->ks; k <= s->ke; ++k)
      for (int j = s->js; j <= s->je; ++j)
        for (int i = s->is; i <= s->ie; ++i) {
          double d = (s->b0[0][IF1(m,k,j,i+1)] - s->b0[0][IF1(m,k,j,i)])/s->dx[3*m];
          if (s->multi_d) d += (s->b0[1][IF2(m,k,j+1,i)] - s->b0[1][IF2(m,k,j,i)])/s->dx[3*m+1];
          if (s->three_d) d += (s->b0[2][IF3(m,k+1,j,i)] - s->b0[2][IF3(m,k,j,i)])/s->dx[3*m+2];
          mx = fmax(mx, fabs(d)); sum += fabs(d); ++cnt;
        }
  out[0] = mx; out[1] = sum/(double)cnt;
}

void akref_totals(akref_sim *s, double *out) {
  const int nt = s->nv < 5 ? s->nv : 5;
  for (int n = 0; n < 5; ++n) out[n] = 0.0;
  for (int m = 0; m < s->nmb; ++m) {
    double vol = s->dx[3*m]*s->dx[3*m+1]*s->dx[3*m+2];
    for (int n = 0; n < nt; ++n)
      for (int k = s->ks; k <= s->ke; ++k)
        for (int j = s->js; j <= s->je; ++j)
          for (int i = s->is; i <= s->ie; ++i) out[n] += vol*s->u0[IX5(m,n,k,j,i)];
  }
}
