/* akref_smr.c -- CPU ORACLE (test infrastructure, see akref.h): boundary values of a statically
 * refined mesh.  Restates, buffer by buffer and in the reference's own order,
 *   src/bvals/buffs_cc.cpp, buffs_fc.cpp   index ranges of the 56 send/recv buffers
 *   src/bvals/bvals_cc.cpp, bvals_fc.cpp   PackAndSend / RecvAndUnpack with same/coarser/finer levels
 *   src/bvals/prolongation.cpp             FillCoarseInBndry, Prolongate (incl. the "owned face" rule)
 *   src/bvals/flux_correct_cc.cpp          restricted fluxes of the conserved variables
 *   src/bvals/flux_correct_fc.cpp          edge EMFs: sum / zero at finer / sum / average
 * for all MeshBlocks of ONE pack on one process (every neighbour is "on this rank", so a sender
 * writes straight into the receiver's buffer, bvals_cc.cpp:122-135).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "akref.h"

typedef struct { int bis, bie, bjs, bje, bks, bke; } bi_t;          /* MeshBufferIndcs, bvals.hpp:51-56 */
typedef struct {                                                    /* MeshBoundaryBuffer, bvals.hpp:62-107 */
  bi_t isame[3], icoar[3], ifine[3], iprol[3], iflux_same[3], iflux_coar[3];
  int isame_ndat, icoar_ndat, ifine_ndat, iflxs_ndat, iflxc_ndat;
  int ox1, ox2, ox3, f1, f2, used;
  double *vars, *flux;       /* (nmb, nvar*nmax) */
  size_t vstride, fstride;
} bb_t;

typedef struct {
  int fc, nvar;
  bb_t sendbuf[56], recvbuf[56];
} bv_t;

struct akref_smr {
  int nmb, nnghbr, ng, nx1, nx2, nx3, one_d, two_d, multi_d, three_d, multilevel;
  int is, ie, js, je, ks, ke, cis, cie, cjs, cje, cks, cke, cnx1, cnx2, cnx3;
  int N1, N2, N3, cN1, cN2, cN3;
  int *gid, *lev, *dest;     /* [nmb][56] NeighborBlock, mesh.hpp:47-52 (rank: all on this process) */
  int *mblev;
  bv_t cc, fc;
  /* twins of the ABI's pack / unpack halves: segments live in the CALLER's buffer, at soff/roff
   * ([4][nmb][56]) or, without them, at layout[cls][slot] {offset, per-block stride} */
  double *xbuf;
  const long long *xlay, *soff, *roff;
};

/* NeighborIndex, src/mesh/nghbr_index.hpp:28-54 */
static int nidx(int ix, int iy, int iz, int n1, int n2) {
  if ((abs(ix) + abs(iy) + abs(iz)) == 0) return -1;
  if (abs(ix*iy*iz) > 1) return -1;
  if (iz == 0) {
    if (ix*iy == 0) return abs(ix)*2*(ix + 1) + abs(iy)*2*(iy + 5) + n1 + 2*n2;
    return 16 + (ix + 1) + 2*(iy + 1) + n1;
  }
  if (ix*iy == 0) return 24 + abs(ix)*(ix + 9) + abs(iy)*(iy + 17) + 2*(iz + 1) + n1 + 2*n2;
  return 48 + (ix + 1)/2 + (iy + 1) + 2*(iz + 1);
}

/* where block m writes the segment of slot n (receiver (dm, dn)) / reads the one it receives; cls 0 cc
 * vars, 1 cc flux, 2 fc vars, 3 fc flux */
static double *wseg(akref_smr *s, int cls, int m, int n, int dm, int dn, double *own, size_t stride) {
  if (!s->xbuf) return own + (size_t)dm*stride;
  if (s->soff) return s->xbuf + s->soff[((size_t)cls*s->nmb + m)*56 + n];
  return s->xbuf + s->xlay[(cls*56 + dn)*2] + (size_t)dm*s->xlay[(cls*56 + dn)*2 + 1];
}
static const double *rseg(const akref_smr *s, int cls, int m, int n, const double *own, size_t stride) {
  if (!s->xbuf) return own + (size_t)m*stride;
  if (s->roff) return s->xbuf + s->roff[((size_t)cls*s->nmb + m)*56 + n];
  return s->xbuf + s->xlay[(cls*56 + n)*2] + (size_t)m*s->xlay[(cls*56 + n)*2 + 1];
}

static int bsz(const bi_t *b) {
  return (b->bie - b->bis + 1)*(b->bje - b->bjs + 1)*(b->bke - b->bks + 1);
}
static int imax(int a, int b) { return a > b ? a : b; }

/* one direction of the index formulas.  s,e: fine active range; cs,ce: coarse; cnx: coarse cells;
 * o: offset of the buffer in this direction; f: the sub-block flag that applies to this direction
 * (buffs_cc.cpp:90-122: x1 takes f1; x2 takes f1 on x1-faces/edges else f2; x3 takes f1 on x1x2
 * edges else f2); act: direction exists (nx>1; x1 always) */
typedef struct { int s, e, cs, ce, cnx, ng, act; } dir_t;

/* ---- cell-centred, buffs_cc.cpp:28-150 (send) and :164-483 (recv) -------------------------------- */
static void cc_send_same(const dir_t *d, int o, int *lo, int *hi) {      /* :36-46 */
  const int ng1 = d->ng - 1;
  *lo = (o > 0) ? (d->e - ng1) : d->s;
  *hi = (o < 0) ? (d->s + ng1) : d->e;
}
static void cc_send_coar(const dir_t *d, int o, int *lo, int *hi) {      /* :63-72 */
  const int ng1 = d->ng - 1;
  *lo = (o > 0) ? (d->ce - ng1) : d->cs;
  *hi = (o < 0) ? (d->cs + ng1) : d->ce;
}
static void cc_send_fine(const dir_t *d, int o, int f, int *lo, int *hi) {   /* :77-122 */
  cc_send_same(d, o, lo, hi);
  if (o == 0 && d->act) {
    if (f == 1) *lo += d->cnx - d->ng; else *hi -= d->cnx - d->ng;
  }
}
static void cc_send_flux(const dir_t *d, int o, int *lo, int *hi) {      /* :126-149 */
  if (o == 0) { *lo = d->cs; *hi = d->ce; }
  else if (o > 0) { *lo = d->ce + 1; *hi = d->ce + 1; }
  else { *lo = d->cs; *hi = d->cs; }
}
static void cc_recv_same(const dir_t *d, int o, int *lo, int *hi) {      /* :172-203 */
  if (o == 0) { *lo = d->s; *hi = d->e; }
  else if (o > 0) { *lo = d->e + 1; *hi = d->e + d->ng; }
  else { *lo = d->s - d->ng; *hi = d->s - 1; }
}
static void cc_recv_coar(const dir_t *d, int o, int f, int n, int *lo, int *hi) {   /* :241-299; n = ng (or ng/2: iprol :366-423) */
  if (o == 0) {
    *lo = d->cs; *hi = d->ce;
    if (d->act) { if (f == 0) *hi += n; else *lo -= n; }
  } else if (o > 0) { *lo = d->ce + 1; *hi = d->ce + n; }
  else { *lo = d->cs - n; *hi = d->cs - 1; }
}
static void cc_recv_fine(const dir_t *d, int o, int f, int *lo, int *hi) {   /* :303-361 */
  if (o == 0) {
    *lo = d->s; *hi = d->e;
    if (d->act) { if (f == 1) *lo += d->cnx; else *hi -= d->cnx; }
  } else if (o > 0) { *lo = d->e + 1; *hi = d->e + d->ng; }
  else { *lo = d->s - d->ng; *hi = d->s - 1; }
}
static void cc_recv_flux(const dir_t *d, int o, int f, int *lo, int *hi) {   /* :427-482 */
  if (o == 0) {
    *lo = d->s; *hi = d->e;
    if (d->act) { if (f == 1) *lo += d->cnx; else *hi -= d->cnx; }
  } else if (o > 0) { *lo = d->e + 1; *hi = d->e + 1; }
  else { *lo = d->s; *hi = d->s; }
}

/* ---- face-centred, buffs_fc.cpp:29-386 (send) and :388-937 (recv).  a = 1 when this is the
 * direction the field component is normal to (its array has one more face there); oth = 1 when the
 * buffer has a non-zero offset in another direction (the multilevel +-1 of :73-84, :128-139, ...) -- */
static void fc_send_same(const dir_t *d, int o, int a, int ml_oth, int *lo, int *hi) {   /* :39-84 */
  const int ng1 = d->ng - 1;
  if (o == 0) { *lo = d->s; *hi = d->e + a; }
  else if (o > 0) { *lo = d->e - ng1; *hi = d->e; }
  else { *lo = d->s + a; *hi = d->s + ng1 + a; }
  if (a && ml_oth) { if (o > 0) (*hi)++; if (o < 0) (*lo)--; }
}
static void fc_send_coar(const dir_t *d, int o, int a, int ml_oth, int *lo, int *hi) {   /* :94-139 */
  const int ng1 = d->ng - 1;
  if (o == 0) { *lo = d->cs; *hi = d->ce + a; }
  else if (o > 0) { *lo = d->ce - ng1; *hi = d->ce; }
  else { *lo = d->cs + a; *hi = d->cs + ng1 + a; }
  if (a && ml_oth) { if (o > 0) (*hi)++; if (o < 0) (*lo)--; }
}
static void fc_send_fine(const dir_t *d, int o, int f, int a, int *lo, int *hi) {        /* :149-252 */
  const int ng1 = d->ng - 1;
  if (o == 0) {
    *lo = d->s; *hi = d->e + a;
    if (d->act) { if (f == 1) *lo += d->cnx - d->ng; else *hi -= d->cnx - d->ng; }
  } else if (o > 0) { *lo = d->e - ng1; *hi = d->e + a; }
  else { *lo = d->s; *hi = d->s + ng1 + a; }
}
/* EMF (edge) fluxes: component v is staggered in the two directions other than v: st = (dir != v) */
static void fc_send_flxc(const dir_t *d, int o, int st, int *lo, int *hi) {              /* :262-305 */
  if (o == 0) { *lo = d->cs; *hi = d->ce + st; }
  else if (o > 0) { *lo = d->ce + 1; *hi = d->ce + 1; }
  else { *lo = d->cs; *hi = d->cs; }
}
static void fc_flxs(const dir_t *d, int o, int st, int *lo, int *hi) {                   /* :308-351 = :890-929 */
  if (o == 0) { *lo = d->s; *hi = d->e + st; }
  else if (o > 0) { *lo = d->e + 1; *hi = d->e + 1; }
  else { *lo = d->s; *hi = d->s; }
}
static void fc_recv_same(const dir_t *d, int o, int a, int ml_oth, int *lo, int *hi) {   /* :396-447 */
  if (o == 0) { *lo = d->s; *hi = d->e + a; }
  else if (o > 0) { *lo = d->e + 1 + a; *hi = d->e + d->ng + a; }
  else { *lo = d->s - d->ng; *hi = d->s - 1; }
  if (a && ml_oth) { if (o > 0) (*lo)--; if (o < 0) (*hi)++; }
}
static void fc_recv_coar(const dir_t *d, int o, int f, int a, int *lo, int *hi) {        /* :456-551 */
  if (o == 0) {
    *lo = d->cs; *hi = d->ce + a;
    if (d->act) { if (f == 0) *hi += d->ng; else *lo -= d->ng; }
  } else if (o > 0) { *lo = d->ce + 1; *hi = d->ce + d->ng + a; }
  else { *lo = d->cs - d->ng; *hi = d->cs - 1 + a; }
}
static void fc_recv_fine(const dir_t *d, int o, int f, int a, int ml_oth, int *lo, int *hi) {   /* :560-669 */
  if (o == 0) {
    *lo = d->s; *hi = d->e + a;
    if (d->act) { if (f == 1) *lo += d->cnx; else *hi -= d->cnx; }
  } else if (o > 0) { *lo = d->e + 1 + a; *hi = d->e + d->ng + a; }
  else { *lo = d->s - d->ng; *hi = d->s - 1; }
  if (a && ml_oth) { if (o > 0) (*lo)--; if (o < 0) (*hi)++; }
}
static void fc_recv_prol(const dir_t *d, int o, int f, int a, int *lo, int *hi) {        /* :678-773 */
  const int cn = d->ng/2;
  if (o == 0) {
    *lo = d->cs; *hi = d->ce + a;
    if (d->act) { if (f == 0) *hi += cn; else *lo -= cn; }
  } else if (o > 0) { *lo = d->ce + 1 + a; *hi = d->ce + cn + a; }
  else { *lo = d->cs - cn; *hi = d->cs - 1; }
}
static void fc_recv_flxc(const dir_t *d, int o, int f, int st, int *lo, int *hi) {       /* :776-885 */
  if (o == 0) {
    *lo = d->s; *hi = d->e + st;
    if (d->act) { if (f == 1) *lo += d->cnx; else *hi -= d->cnx; }
  } else if (o > 0) { *lo = d->e + 1; *hi = d->e + 1; }
  else { *lo = d->s; *hi = d->s; }
}

static void set3(bi_t *b, int dir, int lo, int hi) {
  if (dir == 0) { b->bis = lo; b->bie = hi; }
  else if (dir == 1) { b->bjs = lo; b->bje = hi; }
  else { b->bks = lo; b->bke = hi; }
}

static void init_indices(akref_smr *s, bv_t *bv, int n, int ox1, int ox2, int ox3, int f1, int f2) {
  bb_t *sb = &bv->sendbuf[n], *rb = &bv->recvbuf[n];
  sb->used = rb->used = 1;
  sb->ox1 = rb->ox1 = ox1; sb->ox2 = rb->ox2 = ox2; sb->ox3 = rb->ox3 = ox3;
  sb->f1 = rb->f1 = f1; sb->f2 = rb->f2 = f2;
  const dir_t D[3] = {{s->is, s->ie, s->cis, s->cie, s->cnx1, s->ng, 1},
                      {s->js, s->je, s->cjs, s->cje, s->cnx2, s->ng, s->nx2 > 1},
                      {s->ks, s->ke, s->cks, s->cke, s->cnx3, s->ng, s->nx3 > 1}};
  const int o[3] = {ox1, ox2, ox3};
  const int fl[3] = {f1, (ox1 != 0) ? f1 : f2, (ox1 != 0 && ox2 != 0) ? f1 : f2};
  const int same = (f1 == 0) && (f2 == 0);
  int lo, hi;
  if (!bv->fc) {
    for (int d = 0; d < 3; ++d) {
      if (same) {
        cc_send_same(&D[d], o[d], &lo, &hi); set3(&sb->isame[0], d, lo, hi);
        cc_recv_same(&D[d], o[d], &lo, &hi); set3(&rb->isame[0], d, lo, hi);
      }
      cc_send_coar(&D[d], o[d], &lo, &hi); set3(&sb->icoar[0], d, lo, hi);
      cc_send_fine(&D[d], o[d], fl[d], &lo, &hi); set3(&sb->ifine[0], d, lo, hi);
      cc_send_flux(&D[d], o[d], &lo, &hi); set3(&sb->iflux_coar[0], d, lo, hi);
      cc_recv_coar(&D[d], o[d], fl[d], s->ng, &lo, &hi); set3(&rb->icoar[0], d, lo, hi);
      cc_recv_fine(&D[d], o[d], fl[d], &lo, &hi); set3(&rb->ifine[0], d, lo, hi);
      cc_recv_coar(&D[d], o[d], fl[d], s->ng/2, &lo, &hi); set3(&rb->iprol[0], d, lo, hi);
      cc_recv_flux(&D[d], o[d], fl[d], &lo, &hi); set3(&rb->iflux_coar[0], d, lo, hi);
    }
    if (same) { sb->isame_ndat = bsz(&sb->isame[0]); rb->isame_ndat = bsz(&rb->isame[0]); }
    sb->icoar_ndat = bsz(&sb->icoar[0]); rb->icoar_ndat = bsz(&rb->icoar[0]);
    sb->ifine_ndat = bsz(&sb->ifine[0]); rb->ifine_ndat = bsz(&rb->ifine[0]);
    sb->iflxc_ndat = bsz(&sb->iflux_coar[0]); rb->iflxc_ndat = bsz(&rb->iflux_coar[0]);
  } else {
    const int ml = s->multilevel;
    for (int v = 0; v < 3; ++v) {
      for (int d = 0; d < 3; ++d) {
        const int a = (v == d), st = (v != d);
        const int oth = ml && ((d != 0 && o[0] != 0) || (d != 1 && o[1] != 0) || (d != 2 && o[2] != 0));
        if (same) {
          fc_send_same(&D[d], o[d], a, oth, &lo, &hi); set3(&sb->isame[v], d, lo, hi);
          fc_recv_same(&D[d], o[d], a, oth, &lo, &hi); set3(&rb->isame[v], d, lo, hi);
        }
        fc_send_coar(&D[d], o[d], a, oth, &lo, &hi); set3(&sb->icoar[v], d, lo, hi);
        fc_send_fine(&D[d], o[d], fl[d], a, &lo, &hi); set3(&sb->ifine[v], d, lo, hi);
        fc_send_flxc(&D[d], o[d], st, &lo, &hi); set3(&sb->iflux_coar[v], d, lo, hi);
        fc_flxs(&D[d], o[d], st, &lo, &hi); set3(&sb->iflux_same[v], d, lo, hi);
        fc_recv_coar(&D[d], o[d], fl[d], a, &lo, &hi); set3(&rb->icoar[v], d, lo, hi);
        fc_recv_fine(&D[d], o[d], fl[d], a, oth, &lo, &hi); set3(&rb->ifine[v], d, lo, hi);
        fc_recv_prol(&D[d], o[d], fl[d], a, &lo, &hi); set3(&rb->iprol[v], d, lo, hi);
        fc_recv_flxc(&D[d], o[d], fl[d], st, &lo, &hi); set3(&rb->iflux_coar[v], d, lo, hi);
        fc_flxs(&D[d], o[d], st, &lo, &hi); set3(&rb->iflux_same[v], d, lo, hi);
      }
      if (same) {
        sb->isame_ndat = imax(sb->isame_ndat, bsz(&sb->isame[v]));
        rb->isame_ndat = imax(rb->isame_ndat, bsz(&rb->isame[v]));
      }
      sb->icoar_ndat = imax(sb->icoar_ndat, bsz(&sb->icoar[v]));
      rb->icoar_ndat = imax(rb->icoar_ndat, bsz(&rb->icoar[v]));
      sb->ifine_ndat = imax(sb->ifine_ndat, bsz(&sb->ifine[v]));
      rb->ifine_ndat = imax(rb->ifine_ndat, bsz(&rb->ifine[v]));
      sb->iflxc_ndat = imax(sb->iflxc_ndat, bsz(&sb->iflux_coar[v]));
      rb->iflxc_ndat = imax(rb->iflxc_ndat, bsz(&rb->iflux_coar[v]));
      sb->iflxs_ndat = imax(sb->iflxs_ndat, bsz(&sb->iflux_same[v]));
      rb->iflxs_ndat = imax(rb->iflxs_ndat, bsz(&rb->iflux_same[v]));
    }
  }
  /* AllocateBuffers, bvals.hpp:93-106 (only the receive side is used on one process) */
  const int nv = bv->nvar;
  int nmax = imax(rb->isame_ndat, imax(rb->icoar_ndat, rb->ifine_ndat));
  int smax = imax(sb->isame_ndat, imax(sb->icoar_ndat, sb->ifine_ndat));
  if (smax > nmax) nmax = smax;
  rb->vstride = (size_t)nv*nmax;
  rb->vars = (double *)calloc((size_t)s->nmb*rb->vstride + 1, sizeof(double));
  int fmax = imax(imax(rb->iflxs_ndat, rb->iflxc_ndat), imax(sb->iflxs_ndat, sb->iflxc_ndat));
  rb->fstride = (size_t)nv*fmax;
  rb->flux = (double *)calloc((size_t)s->nmb*rb->fstride + 1, sizeof(double));
}

/* MeshBoundaryValues::InitializeBuffers, src/bvals/bvals.cpp:322-439 */
static void initialize_buffers(akref_smr *s, bv_t *bv) {
  int nfx = 1, nfy = 1, nfz = 1;
  if (s->multilevel) { nfx = 2; if (s->multi_d) nfy = 2; if (s->three_d) nfz = 2; }
  for (int n = -1; n <= 1; n += 2)
    for (int fz = 0; fz < nfz; fz++) for (int fy = 0; fy < nfy; fy++)
      init_indices(s, bv, nidx(n, 0, 0, fy, fz), n, 0, 0, fy, fz);
  if (s->multi_d) {
    for (int m = -1; m <= 1; m += 2)
      for (int fz = 0; fz < nfz; fz++) for (int fx = 0; fx < nfx; fx++)
        init_indices(s, bv, nidx(0, m, 0, fx, fz), 0, m, 0, fx, fz);
    for (int m = -1; m <= 1; m += 2) for (int n = -1; n <= 1; n += 2)
      for (int fz = 0; fz < nfz; fz++)
        init_indices(s, bv, nidx(n, m, 0, fz, 0), n, m, 0, fz, 0);
  }
  if (s->three_d) {
    for (int l = -1; l <= 1; l += 2)
      for (int fy = 0; fy < nfy; fy++) for (int fx = 0; fx < nfx; fx++)
        init_indices(s, bv, nidx(0, 0, l, fx, fy), 0, 0, l, fx, fy);
    for (int l = -1; l <= 1; l += 2) for (int n = -1; n <= 1; n += 2)
      for (int fy = 0; fy < nfy; fy++)
        init_indices(s, bv, nidx(n, 0, l, fy, 0), n, 0, l, fy, 0);
    for (int l = -1; l <= 1; l += 2) for (int m = -1; m <= 1; m += 2)
      for (int fx = 0; fx < nfx; fx++)
        init_indices(s, bv, nidx(0, m, l, fx, 0), 0, m, l, fx, 0);
    for (int l = -1; l <= 1; l += 2) for (int m = -1; m <= 1; m += 2) for (int n = -1; n <= 1; n += 2)
      init_indices(s, bv, nidx(n, m, l, 0, 0), n, m, l, 0, 0);
  }
}

akref_smr *akref_smr_create(const akmi_pack *p, int nvar, const int *nghbr, const int *mblev,
                            int multilevel) {
  akref_smr *s = (akref_smr *)calloc(1, sizeof(akref_smr));
  s->nmb = p->nmb; s->ng = p->ng; s->nx1 = p->nx1; s->nx2 = p->nx2; s->nx3 = p->nx3;
  s->multi_d = p->nx2 > 1; s->three_d = p->nx3 > 1;
  s->one_d = !s->multi_d; s->two_d = s->multi_d && !s->three_d;
  s->multilevel = multilevel;
  s->nnghbr = s->three_d ? 56 : (s->multi_d ? 24 : 8);            /* meshblock.cpp:145-147 */
  const int ng = p->ng;
  /* RegionIndcs incl. the coarse ones, src/mesh/mesh.cpp:285-330 */
  s->is = ng; s->ie = ng + p->nx1 - 1;
  s->js = s->multi_d ? ng : 0; s->je = s->multi_d ? ng + p->nx2 - 1 : 0;
  s->ks = s->three_d ? ng : 0; s->ke = s->three_d ? ng + p->nx3 - 1 : 0;
  s->cnx1 = p->nx1/2; s->cnx2 = s->multi_d ? p->nx2/2 : 1; s->cnx3 = s->three_d ? p->nx3/2 : 1;
  s->cis = ng; s->cie = ng + s->cnx1 - 1;
  s->cjs = s->multi_d ? ng : 0; s->cje = s->multi_d ? ng + s->cnx2 - 1 : 0;
  s->cks = s->three_d ? ng : 0; s->cke = s->three_d ? ng + s->cnx3 - 1 : 0;
  s->N1 = p->nx1 + 2*ng; s->N2 = s->multi_d ? p->nx2 + 2*ng : 1; s->N3 = s->three_d ? p->nx3 + 2*ng : 1;
  s->cN1 = s->cnx1 + 2*ng; s->cN2 = s->multi_d ? s->cnx2 + 2*ng : 1; s->cN3 = s->three_d ? s->cnx3 + 2*ng : 1;
  const size_t nn = (size_t)s->nmb*56;
  s->gid = (int *)malloc(nn*sizeof(int)); s->lev = (int *)malloc(nn*sizeof(int));
  s->dest = (int *)malloc(nn*sizeof(int));
  for (size_t q = 0; q < nn; ++q) { s->gid[q] = nghbr[3*q]; s->lev[q] = nghbr[3*q + 1]; s->dest[q] = nghbr[3*q + 2]; }
  s->mblev = (int *)malloc(s->nmb*sizeof(int));
  memcpy(s->mblev, mblev, s->nmb*sizeof(int));
  s->cc.fc = 0; s->cc.nvar = nvar;
  s->fc.fc = 1; s->fc.nvar = 3;
  initialize_buffers(s, &s->cc);
  initialize_buffers(s, &s->fc);
  return s;
}

void akref_smr_destroy(akref_smr *s) {
  if (!s) return;
  for (int n = 0; n < 56; ++n) {
    free(s->cc.recvbuf[n].vars); free(s->cc.recvbuf[n].flux);
    free(s->fc.recvbuf[n].vars); free(s->fc.recvbuf[n].flux);
  }
  free(s->gid); free(s->lev); free(s->dest); free(s->mblev);
  free(s);
}

/* 1 when slot n of block m holds a neighbour at a finer level (nghbr.lev > mblev) */
int akref_smr_finer(const akref_smr *s, int m, int n) {
  return s->gid[(size_t)m*56 + n] >= 0 && s->lev[(size_t)m*56 + n] > s->mblev[m];
}

/* index boxes of buffer n for table-parity tests: out[kind][v][6], kind = same, coar, fine, prol,
 * flux_same, flux_coar; returns 0 when the buffer is not used in this dimensionality */
int akref_smr_indices(const akref_smr *s, int fc, int send, int n, int *out) {
  const bv_t *bv = fc ? &s->fc : &s->cc;
  const bb_t *b = send ? &bv->sendbuf[n] : &bv->recvbuf[n];
  if (!b->used) return 0;
  const bi_t *k[6] = {b->isame, b->icoar, b->ifine, b->iprol, b->iflux_same, b->iflux_coar};
  for (int q = 0; q < 6; ++q) for (int v = 0; v < 3; ++v) {
    const bi_t *x = &k[q][v];
    int *o = out + (q*3 + v)*6;
    o[0] = x->bis; o[1] = x->bie; o[2] = x->bjs; o[3] = x->bje; o[4] = x->bks; o[5] = x->bke;
  }
  return 1;
}

#define NG(m,n) ((size_t)(m)*56 + (n))
#define A5(a,nv,m,v,k,j,i) (a)[(((((size_t)(m)*(nv) + (v))*s->N3 + (k))*s->N2 + (j))*s->N1 + (i))]
#define C5(a,nv,m,v,k,j,i) (a)[(((((size_t)(m)*(nv) + (v))*s->cN3 + (k))*s->cN2 + (j))*s->cN1 + (i))]
/* face fields, fine (F*) and coarse (G*) */
#define F1(a,m,k,j,i) (a)[((((size_t)(m)*s->N3 + (k))*s->N2 + (j))*(s->N1+1) + (i))]
#define F2(a,m,k,j,i) (a)[((((size_t)(m)*s->N3 + (k))*(s->N2+1) + (j))*s->N1 + (i))]
#define F3(a,m,k,j,i) (a)[((((size_t)(m)*(s->N3+1) + (k))*s->N2 + (j))*s->N1 + (i))]
#define G1(a,m,k,j,i) (a)[((((size_t)(m)*s->cN3 + (k))*s->cN2 + (j))*(s->cN1+1) + (i))]
#define G2(a,m,k,j,i) (a)[((((size_t)(m)*s->cN3 + (k))*(s->cN2+1) + (j))*s->cN1 + (i))]
#define G3(a,m,k,j,i) (a)[((((size_t)(m)*(s->cN3+1) + (k))*s->cN2 + (j))*s->cN1 + (i))]
/* edge fields (fine) */
#define E1(a,m,k,j,i) (a)[((((size_t)(m)*(s->N3+1) + (k))*(s->N2+1) + (j))*s->N1 + (i))]
#define E2(a,m,k,j,i) (a)[((((size_t)(m)*(s->N3+1) + (k))*s->N2 + (j))*(s->N1+1) + (i))]
#define E3(a,m,k,j,i) (a)[((((size_t)(m)*s->N3 + (k))*(s->N2+1) + (j))*(s->N1+1) + (i))]

static double *fptr(double *b1, double *b2, double *b3, int v) { return v == 0 ? b1 : (v == 1 ? b2 : b3); }

static double fget(const akref_smr *s, const double *a, int v, int coarse, int m, int k, int j, int i) {
  if (!coarse) return v == 0 ? F1(a,m,k,j,i) : (v == 1 ? F2(a,m,k,j,i) : F3(a,m,k,j,i));
  return v == 0 ? G1(a,m,k,j,i) : (v == 1 ? G2(a,m,k,j,i) : G3(a,m,k,j,i));
}
static void fset(const akref_smr *s, double *a, int v, int coarse, int m, int k, int j, int i, double x) {
  if (!coarse) { if (v == 0) F1(a,m,k,j,i) = x; else if (v == 1) F2(a,m,k,j,i) = x; else F3(a,m,k,j,i) = x; }
  else { if (v == 0) G1(a,m,k,j,i) = x; else if (v == 1) G2(a,m,k,j,i) = x; else G3(a,m,k,j,i) = x; }
}

/* ---- PackAndSendCC, src/bvals/bvals_cc.cpp:42-267 --------------------------------------------- */
int akref_smr_send_cc(akref_smr *s, const double *a, const double *ca) {
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (s->gid[NG(m,n)] < 0) continue;
    const bb_t *sb = &s->cc.sendbuf[n];
    const bi_t *x;
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    if (nl < ml) x = &sb->icoar[0]; else if (nl == ml) x = &sb->isame[0]; else x = &sb->ifine[0];
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
    const int dm = s->gid[NG(m,n)], dn = s->dest[NG(m,n)];
    bb_t *rb = &s->cc.recvbuf[dn];
    for (int v = 0; v < nvar; ++v) for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j)
      for (int i = il; i <= iu; ++i) {
        const size_t bi = (size_t)(i-il + ni*(j-jl + nj*(k-kl + nk*v)));
        wseg(s, 0, m, n, dm, dn, rb->vars, rb->vstride)[bi] = (nl >= ml) ? A5(a,nvar,m,v,k,j,i) : C5(ca,nvar,m,v,k,j,i);
      }
  }
  return 0;
}

/* ---- RecvAndUnpackCC, src/bvals/bvals_cc.cpp:273-447 ------------------------------------------ */
int akref_smr_recv_cc(akref_smr *s, double *a, double *ca) {
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (s->gid[NG(m,n)] < 0) continue;
    const bb_t *rb = &s->cc.recvbuf[n];
    const bi_t *x;
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    if (nl < ml) x = &rb->icoar[0]; else if (nl == ml) x = &rb->isame[0]; else x = &rb->ifine[0];
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
    for (int v = 0; v < nvar; ++v) for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j)
      for (int i = il; i <= iu; ++i) {
        const size_t bi = (size_t)(i-il + ni*(j-jl + nj*(k-kl + nk*v)));
        const double val = rseg(s, 0, m, n, rb->vars, rb->vstride)[bi];
        if (nl >= ml) A5(a,nvar,m,v,k,j,i) = val; else C5(ca,nvar,m,v,k,j,i) = val;
      }
  }
  return 0;
}

/* ---- PackAndSendFC, src/bvals/bvals_fc.cpp:63-283 --------------------------------------------- */
int akref_smr_send_fc(akref_smr *s, const double *b1, const double *b2, const double *b3,
                      const double *cb1, const double *cb2, const double *cb3) {
  for (int m = 0; m < s->nmb; ++m) for (int v = 0; v < 3; ++v) for (int n = 0; n < s->nnghbr; ++n) {
    if (s->gid[NG(m,n)] < 0) continue;
    const bb_t *sb = &s->fc.sendbuf[n];
    const bi_t *x; int ndat;
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    if (nl < ml) { x = &sb->icoar[v]; ndat = sb->icoar_ndat; }
    else if (nl == ml) { x = &sb->isame[v]; ndat = sb->isame_ndat; }
    else { x = &sb->ifine[v]; ndat = sb->ifine_ndat; }
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1;
    const int dm = s->gid[NG(m,n)], dn = s->dest[NG(m,n)];
    bb_t *rb = &s->fc.recvbuf[dn];
    const double *src = (nl >= ml) ? (v == 0 ? b1 : (v == 1 ? b2 : b3)) : (v == 0 ? cb1 : (v == 1 ? cb2 : cb3));
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i)
      wseg(s, 2, m, n, dm, dn, rb->vars, rb->vstride)[(size_t)ndat*v + (i-il + ni*(j-jl + nj*(k-kl)))] =
          fget(s, src, v, nl < ml, m, k, j, i);
  }
  return 0;
}

/* IsActiveFCFace, src/bvals/bvals_fc.cpp:32-49 = prolongation.cpp:70-88 */
static int is_active_fc_face(const akref_smr *s, int v, int k, int j, int i) {
  if (v == 0) return (i >= s->is) && (i <= s->ie + 1) && (j >= s->js) && (j <= s->je) && (k >= s->ks) && (k <= s->ke);
  if (v == 1) return (i >= s->is) && (i <= s->ie) && (j >= s->js) && (j <= s->je + 1) && (k >= s->ks) && (k <= s->ke);
  return (i >= s->is) && (i <= s->ie) && (j >= s->js) && (j <= s->je) && (k >= s->ks) && (k <= s->ke + 1);
}

/* ---- RecvAndUnpackFC, src/bvals/bvals_fc.cpp:289-436: buffers of a (block, component) are
 * unpacked one after the other in slot order; data of same-level and finer neighbours never
 * overwrites an active face ---------------------------------------------------------------- */
int akref_smr_recv_fc(akref_smr *s, double *b1, double *b2, double *b3, double *cb1, double *cb2,
                      double *cb3) {
  for (int m = 0; m < s->nmb; ++m) for (int v = 0; v < 3; ++v) for (int n = 0; n < s->nnghbr; ++n) {
    if (s->gid[NG(m,n)] < 0) continue;
    const bb_t *rb = &s->fc.recvbuf[n];
    const bi_t *x; int ndat;
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    if (nl < ml) { x = &rb->icoar[v]; ndat = rb->icoar_ndat; }
    else if (nl == ml) { x = &rb->isame[v]; ndat = rb->isame_ndat; }
    else { x = &rb->ifine[v]; ndat = rb->ifine_ndat; }
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1;
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const double val = rseg(s, 2, m, n, rb->vars, rb->vstride)[(size_t)ndat*v + (i-il + ni*(j-jl + nj*(k-kl)))];
      if (nl >= ml) {
        if (is_active_fc_face(s, v, k, j, i)) continue;
        fset(s, fptr(b1, b2, b3, v), v, 0, m, k, j, i, val);
      } else {
        fset(s, fptr(cb1, cb2, cb3, v), v, 1, m, k, j, i, val);
      }
    }
  }
  return 0;
}

/* ---- FillCoarseInBndryCC, src/bvals/prolongation.cpp:366-462 --------------------------------- */
int akref_smr_fill_coarse_cc(akref_smr *s, const double *a, double *ca) {
  if (!s->multi_d) return 0;
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] == s->mblev[m])) continue;
    const bi_t *x = &s->cc.recvbuf[n].isame[0];
    const int il = (x->bis + s->cis)/2, iu = (x->bie + s->cis)/2;
    const int jl = (x->bjs + s->cjs)/2, ju = (x->bje + s->cjs)/2;
    const int kl = (x->bks + s->cks)/2, ku = (x->bke + s->cks)/2;
    for (int v = 0; v < nvar; ++v) for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j)
      for (int i = il; i <= iu; ++i) {
        const int finei = (i - s->cis)*2 + s->is, finej = (j - s->cjs)*2 + s->js, finek = (k - s->cks)*2 + s->ks;
        if (!s->three_d) {
          C5(ca,nvar,m,v,kl,j,i) = 0.25*(A5(a,nvar,m,v,kl,finej,finei) + A5(a,nvar,m,v,kl,finej,finei+1)
                                       + A5(a,nvar,m,v,kl,finej+1,finei) + A5(a,nvar,m,v,kl,finej+1,finei+1));
        } else {
          C5(ca,nvar,m,v,k,j,i) = 0.125*(
              A5(a,nvar,m,v,finek,finej,finei) + A5(a,nvar,m,v,finek,finej,finei+1)
            + A5(a,nvar,m,v,finek,finej+1,finei) + A5(a,nvar,m,v,finek,finej+1,finei+1)
            + A5(a,nvar,m,v,finek+1,finej,finei) + A5(a,nvar,m,v,finek+1,finej,finei+1)
            + A5(a,nvar,m,v,finek+1,finej+1,finei) + A5(a,nvar,m,v,finek+1,finej+1,finei+1));
        }
      }
  }
  return 0;
}

static double sgn_(double x) { return (x < 0.0) ? -1.0 : 1.0; }           /* SIGN, src/athena.hpp:52 */
static double mm8(double dl, double dr) { return 0.125*(sgn_(dl) + sgn_(dr))*fmin(fabs(dl), fabs(dr)); }

/* ---- ProlongateCC, src/bvals/prolongation.cpp:470-546 with ProlongCC, src/mesh/prolongation.hpp:19-63 */
int akref_smr_prolong_cc(akref_smr *s, double *a, const double *ca) {
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bi_t *x = &s->cc.recvbuf[n].iprol[0];
    for (int v = 0; v < nvar; ++v) for (int k = x->bks; k <= x->bke; ++k) for (int j = x->bjs; j <= x->bje; ++j)
      for (int i = x->bis; i <= x->bie; ++i) {
        const int fi = (i - s->cis)*2 + s->is, fj = (j - s->cjs)*2 + s->js, fk = (k - s->cks)*2 + s->ks;
        const double q = C5(ca,nvar,m,v,k,j,i);
        const double dvar1 = mm8(q - C5(ca,nvar,m,v,k,j,i-1), C5(ca,nvar,m,v,k,j,i+1) - q);
        double dvar2 = 0.0, dvar3 = 0.0;
        if (s->multi_d) dvar2 = mm8(q - C5(ca,nvar,m,v,k,j-1,i), C5(ca,nvar,m,v,k,j+1,i) - q);
        if (s->three_d) dvar3 = mm8(q - C5(ca,nvar,m,v,k-1,j,i), C5(ca,nvar,m,v,k+1,j,i) - q);
        A5(a,nvar,m,v,fk,fj,fi) = q - dvar1 - dvar2 - dvar3;
        A5(a,nvar,m,v,fk,fj,fi+1) = q + dvar1 - dvar2 - dvar3;
        if (s->multi_d) {
          A5(a,nvar,m,v,fk,fj+1,fi) = q - dvar1 + dvar2 - dvar3;
          A5(a,nvar,m,v,fk,fj+1,fi+1) = q + dvar1 + dvar2 - dvar3;
        }
        if (s->three_d) {
          A5(a,nvar,m,v,fk+1,fj,fi) = q - dvar1 - dvar2 + dvar3;
          A5(a,nvar,m,v,fk+1,fj,fi+1) = q + dvar1 - dvar2 + dvar3;
          A5(a,nvar,m,v,fk+1,fj+1,fi) = q - dvar1 + dvar2 + dvar3;
          A5(a,nvar,m,v,fk+1,fj+1,fi+1) = q + dvar1 + dvar2 + dvar3;
        }
      }
  }
  return 0;
}

/* ---- FillCoarseInBndryFC, src/bvals/prolongation.cpp:556-644 ---------------------------------- */
int akref_smr_fill_coarse_fc(akref_smr *s, const double *b1, const double *b2, const double *b3,
                             double *cb1, double *cb2, double *cb3) {
  if (!s->multi_d) return 0;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < 3; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] == s->mblev[m])) continue;
    const bi_t *x = &s->fc.recvbuf[n].isame[v];
    const int il = (x->bis + s->cis)/2, iu = (x->bie + s->cis)/2;
    const int jl = (x->bjs + s->cjs)/2, ju = (x->bje + s->cjs)/2;
    const int kl = (x->bks + s->cks)/2, ku = (x->bke + s->cks)/2;
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const int fk = (k - s->cks)*2 + s->ks, fj = (j - s->cjs)*2 + s->js, fi = (i - s->cis)*2 + s->is;
      if (!s->three_d) {
        if (v == 0) G1(cb1,m,kl,j,i) = 0.5*(F1(b1,m,kl,fj,fi) + F1(b1,m,kl,fj+1,fi));
        else if (v == 1) G2(cb2,m,kl,j,i) = 0.5*(F2(b2,m,kl,fj,fi) + F2(b2,m,kl,fj,fi+1));
        else {
          const double b3c = 0.25*(F3(b3,m,kl,fj,fi) + F3(b3,m,kl,fj,fi+1) + F3(b3,m,kl,fj+1,fi) + F3(b3,m,kl,fj+1,fi+1));
          G3(cb3,m,kl,j,i) = b3c; G3(cb3,m,kl+1,j,i) = b3c;
        }
      } else {
        if (v == 0) G1(cb1,m,k,j,i) = 0.25*(F1(b1,m,fk,fj,fi) + F1(b1,m,fk,fj+1,fi) + F1(b1,m,fk+1,fj,fi) + F1(b1,m,fk+1,fj+1,fi));
        else if (v == 1) G2(cb2,m,k,j,i) = 0.25*(F2(b2,m,fk,fj,fi) + F2(b2,m,fk,fj,fi+1) + F2(b2,m,fk+1,fj,fi) + F2(b2,m,fk+1,fj,fi+1));
        else G3(cb3,m,k,j,i) = 0.25*(F3(b3,m,fk,fj,fi) + F3(b3,m,fk,fj,fi+1) + F3(b3,m,fk,fj+1,fi) + F3(b3,m,fk,fj+1,fi+1));
      }
    }
  }
  return 0;
}

/* NeighborOffsetFromIndex / MaxNeighborLevelAtOffset / CanProlongateFCFace, prolongation.cpp:28-133 */
static int max_nghbr_level_at(const akref_smr *s, int m, int ox1, int ox2, int ox3) {
  int max_lev = -1;
  for (int n1 = 0; n1 <= 1; ++n1) for (int n2 = 0; n2 <= 1; ++n2) {
    const int idx = nidx(ox1, ox2, ox3, n1, n2);
    if (idx >= 0 && idx < s->nnghbr && s->gid[NG(m,idx)] >= 0)
      max_lev = s->lev[NG(m,idx)] > max_lev ? s->lev[NG(m,idx)] : max_lev;
  }
  return max_lev;
}
static int can_prolongate(const akref_smr *s, int m, int v, int k, int j, int i, int ox1, int ox2, int ox3) {
  if (!is_active_fc_face(s, v, k, j, i)) return 1;
  const int my_lev = s->mblev[m];
  int normal_ox = 0;
  if (v == 0) {
    if (i == s->is) normal_ox = -1; else if (i == s->ie + 1) normal_ox = 1; else return 0;
    return (ox1 == normal_ox) && (ox2 == 0) && (ox3 == 0) && (max_nghbr_level_at(s, m, normal_ox, 0, 0) < my_lev);
  } else if (v == 1) {
    if (j == s->js) normal_ox = -1; else if (j == s->je + 1) normal_ox = 1; else return 0;
    return (ox1 == 0) && (ox2 == normal_ox) && (ox3 == 0) && (max_nghbr_level_at(s, m, 0, normal_ox, 0) < my_lev);
  }
  if (k == s->ks) normal_ox = -1; else if (k == s->ke + 1) normal_ox = 1; else return 0;
  return (ox1 == 0) && (ox2 == 0) && (ox3 == normal_ox) && (max_nghbr_level_at(s, m, 0, 0, normal_ox) < my_lev);
}
#define STORE(arr, v, k, j, i, val) \
  do { if (can_prolongate(s, m, v, k, j, i, ox1, ox2, ox3)) fset(s, arr, v, 0, m, k, j, i, (val)); } while (0)

/* ---- ProlongateFC, src/bvals/prolongation.cpp:650-785 ---------------------------------------- */
int akref_smr_prolong_fc(akref_smr *s, double *b1, double *b2, double *b3, const double *cb1,
                         const double *cb2, const double *cb3) {
  /* shared faces, :664-722 with ProlongFCSharedX1/2/3FaceOwned :149-258 */
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < 3; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bb_t *rb = &s->fc.recvbuf[n];
    const int ox1 = rb->ox1, ox2 = rb->ox2, ox3 = rb->ox3;
    const bi_t *x = &rb->iprol[v];
    for (int k = x->bks; k <= x->bke; ++k) for (int j = x->bjs; j <= x->bje; ++j) for (int i = x->bis; i <= x->bie; ++i) {
      const int fi = (i - s->cis)*2 + s->is;
      const int fj = s->multi_d ? ((j - s->cjs)*2 + s->js) : j;
      const int fk = s->three_d ? ((k - s->cks)*2 + s->ks) : k;
      if (v == 0) {
        double dvar2 = 0.0, dvar3 = 0.0;
        if (s->multi_d) dvar2 = mm8(G1(cb1,m,k,j,i) - G1(cb1,m,k,j-1,i), G1(cb1,m,k,j+1,i) - G1(cb1,m,k,j,i));
        if (s->three_d) dvar3 = mm8(G1(cb1,m,k,j,i) - G1(cb1,m,k-1,j,i), G1(cb1,m,k+1,j,i) - G1(cb1,m,k,j,i));
        STORE(b1, 0, fk, fj, fi, G1(cb1,m,k,j,i) - dvar2 - dvar3);
        if (s->multi_d) STORE(b1, 0, fk, fj+1, fi, G1(cb1,m,k,j,i) + dvar2 - dvar3);
        if (s->three_d) {
          STORE(b1, 0, fk+1, fj, fi, G1(cb1,m,k,j,i) - dvar2 + dvar3);
          STORE(b1, 0, fk+1, fj+1, fi, G1(cb1,m,k,j,i) + dvar2 + dvar3);
        }
      } else if (v == 1) {
        const double dvar1 = mm8(G2(cb2,m,k,j,i) - G2(cb2,m,k,j,i-1), G2(cb2,m,k,j,i+1) - G2(cb2,m,k,j,i));
        double dvar3 = 0.0;
        if (s->three_d) dvar3 = mm8(G2(cb2,m,k,j,i) - G2(cb2,m,k-1,j,i), G2(cb2,m,k+1,j,i) - G2(cb2,m,k,j,i));
        STORE(b2, 1, fk, fj, fi, G2(cb2,m,k,j,i) - dvar1 - dvar3);
        STORE(b2, 1, fk, fj, fi+1, G2(cb2,m,k,j,i) + dvar1 - dvar3);
        if (s->three_d) {
          STORE(b2, 1, fk+1, fj, fi, G2(cb2,m,k,j,i) - dvar1 + dvar3);
          STORE(b2, 1, fk+1, fj, fi+1, G2(cb2,m,k,j,i) + dvar1 + dvar3);
        }
      } else {
        const double dvar1 = mm8(G3(cb3,m,k,j,i) - G3(cb3,m,k,j,i-1), G3(cb3,m,k,j,i+1) - G3(cb3,m,k,j,i));
        double dvar2 = 0.0;
        if (s->multi_d) dvar2 = mm8(G3(cb3,m,k,j,i) - G3(cb3,m,k,j-1,i), G3(cb3,m,k,j+1,i) - G3(cb3,m,k,j,i));
        STORE(b3, 2, fk, fj, fi, G3(cb3,m,k,j,i) - dvar1 - dvar2);
        STORE(b3, 2, fk, fj, fi+1, G3(cb3,m,k,j,i) + dvar1 - dvar2);
        if (s->multi_d) {
          STORE(b3, 2, fk, fj+1, fi, G3(cb3,m,k,j,i) - dvar1 + dvar2);
          STORE(b3, 2, fk, fj+1, fi+1, G3(cb3,m,k,j,i) + dvar1 + dvar2);
        }
      }
    }
  }
  /* interior faces, :730-782 with ProlongFCInternalOwned :260-359 */
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bb_t *rb = &s->fc.recvbuf[n];
    const int ox1 = rb->ox1, ox2 = rb->ox2, ox3 = rb->ox3;
    const int il = rb->iprol[2].bis, iu = rb->iprol[2].bie, jl = rb->iprol[0].bjs, ju = rb->iprol[0].bje;
    const int kl = rb->iprol[1].bks, ku = rb->iprol[1].bke;
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const int fi = (i - s->cis)*2 + s->is, fj = (j - s->cjs)*2 + s->js, fk = (k - s->cks)*2 + s->ks;
      if (s->one_d) {
        STORE(b1, 0, fk, fj, fi+1, 0.5*(F1(b1,m,fk,fj,fi) + F1(b1,m,fk,fj,fi+2)));
      } else if (s->three_d) {
        double Uxx = 0.0, Vyy = 0.0, Wzz = 0.0, Uxyz = 0.0, Vxyz = 0.0, Wxyz = 0.0;
        for (int jj = 0; jj < 2; jj++) {
          const int jsgn = 2*jj - 1, fjj = fj + jj, fjp = fj + 2*jj;
          for (int ii = 0; ii < 2; ii++) {
            const int isgn = 2*ii - 1, fii = fi + ii, fip = fi + 2*ii;
            Uxx += isgn*(jsgn*(F2(b2,m,fk,fjp,fii) + F2(b2,m,fk+1,fjp,fii)) + (F3(b3,m,fk+2,fjj,fii) - F3(b3,m,fk,fjj,fii)));
            Vyy += jsgn*((F3(b3,m,fk+2,fjj,fii) - F3(b3,m,fk,fjj,fii)) + isgn*(F1(b1,m,fk,fjj,fip) + F1(b1,m,fk+1,fjj,fip)));
            Wzz += isgn*(F1(b1,m,fk+1,fjj,fip) - F1(b1,m,fk,fjj,fip)) + jsgn*(F2(b2,m,fk+1,fjp,fii) - F2(b2,m,fk,fjp,fii));
            Uxyz += isgn*jsgn*(F1(b1,m,fk+1,fjj,fip) - F1(b1,m,fk,fjj,fip));
            Vxyz += isgn*jsgn*(F2(b2,m,fk+1,fjp,fii) - F2(b2,m,fk,fjp,fii));
            Wxyz += isgn*jsgn*(F3(b3,m,fk+2,fjj,fii) - F3(b3,m,fk,fjj,fii));
          }
        }
        Uxx *= 0.125; Vyy *= 0.125; Wzz *= 0.125;
        Uxyz *= 0.0625; Vxyz *= 0.0625; Wxyz *= 0.0625;
        STORE(b1, 0, fk, fj, fi+1, 0.5*(F1(b1,m,fk,fj,fi) + F1(b1,m,fk,fj,fi+2)) + Uxx - Vxyz - Wxyz);
        STORE(b1, 0, fk, fj+1, fi+1, 0.5*(F1(b1,m,fk,fj+1,fi) + F1(b1,m,fk,fj+1,fi+2)) + Uxx - Vxyz + Wxyz);
        STORE(b1, 0, fk+1, fj, fi+1, 0.5*(F1(b1,m,fk+1,fj,fi) + F1(b1,m,fk+1,fj,fi+2)) + Uxx + Vxyz - Wxyz);
        STORE(b1, 0, fk+1, fj+1, fi+1, 0.5*(F1(b1,m,fk+1,fj+1,fi) + F1(b1,m,fk+1,fj+1,fi+2)) + Uxx + Vxyz + Wxyz);
        STORE(b2, 1, fk, fj+1, fi, 0.5*(F2(b2,m,fk,fj,fi) + F2(b2,m,fk,fj+2,fi)) + Vyy - Uxyz - Wxyz);
        STORE(b2, 1, fk, fj+1, fi+1, 0.5*(F2(b2,m,fk,fj,fi+1) + F2(b2,m,fk,fj+2,fi+1)) + Vyy - Uxyz + Wxyz);
        STORE(b2, 1, fk+1, fj+1, fi, 0.5*(F2(b2,m,fk+1,fj,fi) + F2(b2,m,fk+1,fj+2,fi)) + Vyy + Uxyz - Wxyz);
        STORE(b2, 1, fk+1, fj+1, fi+1, 0.5*(F2(b2,m,fk+1,fj,fi+1) + F2(b2,m,fk+1,fj+2,fi+1)) + Vyy + Uxyz + Wxyz);
        STORE(b3, 2, fk+1, fj, fi, 0.5*(F3(b3,m,fk+2,fj,fi) + F3(b3,m,fk,fj,fi)) + Wzz - Uxyz - Vxyz);
        STORE(b3, 2, fk+1, fj, fi+1, 0.5*(F3(b3,m,fk+2,fj,fi+1) + F3(b3,m,fk,fj,fi+1)) + Wzz - Uxyz + Vxyz);
        STORE(b3, 2, fk+1, fj+1, fi, 0.5*(F3(b3,m,fk+2,fj+1,fi) + F3(b3,m,fk,fj+1,fi)) + Wzz + Uxyz - Vxyz);
        STORE(b3, 2, fk+1, fj+1, fi+1, 0.5*(F3(b3,m,fk+2,fj+1,fi+1) + F3(b3,m,fk,fj+1,fi+1)) + Wzz + Uxyz + Vxyz);
      } else {
        const double tmp1 = 0.25*(F2(b2,m,fk,fj+2,fi+1) - F2(b2,m,fk,fj,fi+1) - F2(b2,m,fk,fj+2,fi) + F2(b2,m,fk,fj,fi));
        const double tmp2 = 0.25*(F1(b1,m,fk,fj,fi) - F1(b1,m,fk,fj,fi+2) - F1(b1,m,fk,fj+1,fi) + F1(b1,m,fk,fj+1,fi+2));
        STORE(b1, 0, fk, fj, fi+1, 0.5*(F1(b1,m,fk,fj,fi) + F1(b1,m,fk,fj,fi+2)) + tmp1);
        STORE(b1, 0, fk, fj+1, fi+1, 0.5*(F1(b1,m,fk,fj+1,fi) + F1(b1,m,fk,fj+1,fi+2)) + tmp1);
        STORE(b2, 1, fk, fj+1, fi, 0.5*(F2(b2,m,fk,fj,fi) + F2(b2,m,fk,fj+2,fi)) + tmp2);
        STORE(b2, 1, fk, fj+1, fi+1, 0.5*(F2(b2,m,fk,fj,fi+1) + F2(b2,m,fk,fj+2,fi+1)) + tmp2);
      }
    }
  }
  return 0;
}

/* ---- PackAndSendFluxCC + RecvAndUnpackFluxCC, src/bvals/flux_correct_cc.cpp:29-197,199-304.
 * Flux arrays flx1 (nmb,nvar,N3,N2,N1+fs), flx2, flx3: fs = 1 face-shaped (MHD, mhd.cpp:153-160),
 * fs = 0 cell-shaped (hydro, hydro.cpp:289-298). ------------------------------------------------ */
static int flux_cc_phase(akref_smr *s, double *flx1, double *flx2, double *flx3, int fs, int phase) {
  const int nvar = s->cc.nvar;
  const int N1 = s->N1, N2 = s->N2, N3 = s->N3;
#define X1(m,v,k,j,i) flx1[(((((size_t)(m)*nvar + (v))*N3 + (k))*N2 + (j))*(N1+fs) + (i))]
#define X2(m,v,k,j,i) flx2[(((((size_t)(m)*nvar + (v))*N3 + (k))*(N2+fs) + (j))*N1 + (i))]
#define X3(m,v,k,j,i) flx3[(((((size_t)(m)*nvar + (v))*(N3+fs) + (k))*N2 + (j))*N1 + (i))]
  if (phase & 1)
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < nvar; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bi_t *x = &s->cc.sendbuf[n].iflux_coar[0];
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
    const int dm = s->gid[NG(m,n)], dn = s->dest[NG(m,n)];
    bb_t *rb = &s->cc.recvbuf[dn];
    double *out = wseg(s, 1, m, n, dm, dn, rb->flux, rb->fstride);
    if (n < 8) {
      const int fi = 2*il - s->cis;
      for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) {
        const int fj = 2*j - s->cjs, fk = 2*k - s->cks;
        double rflx;
        if (s->one_d) rflx = X1(m,v,0,0,fi);
        else if (s->two_d) rflx = 0.5*(X1(m,v,0,fj,fi) + X1(m,v,0,fj+1,fi));
        else rflx = 0.25*(X1(m,v,fk,fj,fi) + X1(m,v,fk,fj+1,fi) + X1(m,v,fk+1,fj,fi) + X1(m,v,fk+1,fj+1,fi));
        out[(j-jl + nj*(k-kl + nk*v))] = rflx;
      }
    } else if (n < 16) {
      const int fj = 2*jl - s->cjs;
      for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) {
        const int fi = 2*i - s->cis, fk = 2*k - s->cks;
        double rflx;
        if (s->two_d) rflx = 0.5*(X2(m,v,0,fj,fi) + X2(m,v,0,fj,fi+1));
        else rflx = 0.25*(X2(m,v,fk,fj,fi) + X2(m,v,fk,fj,fi+1) + X2(m,v,fk+1,fj,fi) + X2(m,v,fk+1,fj,fi+1));
        out[(i-il + ni*(k-kl + nk*v))] = rflx;
      }
    } else if (n >= 24 && n < 32) {
      const int fk = 2*kl - s->cks;
      for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
        const int fi = 2*i - s->cis, fj = 2*j - s->cjs;
        out[(i-il + ni*(j-jl + nj*v))] = 0.25*(X3(m,v,fk,fj,fi) + X3(m,v,fk,fj,fi+1) + X3(m,v,fk,fj+1,fi) + X3(m,v,fk,fj+1,fi+1));
      }
    }
  }
  if (phase & 2)
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < nvar; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] > s->mblev[m])) continue;
    const bb_t *rb = &s->cc.recvbuf[n];
    const bi_t *x = &rb->iflux_coar[0];
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1, nk = ku - kl + 1;
    const double *in = rseg(s, 1, m, n, rb->flux, rb->fstride);
    if (n < 8) {
      for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) X1(m,v,k,j,il) = in[(j-jl + nj*(k-kl + nk*v))];
    } else if (n < 16) {
      for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) X2(m,v,k,jl,i) = in[(i-il + ni*(k-kl + nk*v))];
    } else if (n >= 24 && n < 32) {
      for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) X3(m,v,kl,j,i) = in[(i-il + ni*(j-jl + nj*v))];
    }
  }
#undef X1
#undef X2
#undef X3
  return 0;
}
int akref_smr_flux_cc(akref_smr *s, double *flx1, double *flx2, double *flx3, int fs) {
  return flux_cc_phase(s, flx1, flx2, flx3, fs, 3);
}

/* ---- PackAndSendFluxFC, src/bvals/flux_correct_fc.cpp:29-372 ---------------------------------- */
static void send_flux_fc(akref_smr *s, const double *e1, const double *e2, const double *e3) {
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < 3; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] <= s->mblev[m])) continue;
    const bb_t *sb = &s->fc.sendbuf[n];
    const int same = (s->lev[NG(m,n)] == s->mblev[m]);
    const bi_t *x = same ? &sb->iflux_same[v] : &sb->iflux_coar[v];
    const int ndat = same ? sb->iflxs_ndat : sb->iflxc_ndat;
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1;
    const int dm = s->gid[NG(m,n)], dn = s->dest[NG(m,n)];
    bb_t *rb = &s->fc.recvbuf[dn];
    double *out = wseg(s, 3, m, n, dm, dn, rb->flux, rb->fstride) + (size_t)ndat*v;
    const int cis = s->cis, cjs = s->cjs, cks = s->cks;
    if (n < 8) {                                              /* x1 faces :78-123 */
      const int fi = 2*il - cis;
      for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) {
        const int fj = 2*j - cjs, fk = 2*k - cks;
        double rflx;
        if (v == 1) {
          if (same) rflx = E2(e2,m,k,j,il);
          else if (s->one_d) rflx = E2(e2,m,0,0,fi);
          else if (s->two_d) rflx = 0.5*(E2(e2,m,0,fj,fi) + E2(e2,m,0,fj+1,fi));
          else rflx = 0.5*(E2(e2,m,fk,fj,fi) + E2(e2,m,fk,fj+1,fi));
          out[(j-jl + nj*(k-kl))] = rflx;
        } else if (v == 2) {
          if (same) rflx = E3(e3,m,k,j,il);
          else if (s->one_d) rflx = E3(e3,m,0,0,fi);
          else if (s->two_d) rflx = E3(e3,m,0,fj,fi);
          else rflx = 0.5*(E3(e3,m,fk,fj,fi) + E3(e3,m,fk+1,fj,fi));
          out[(j-jl + nj*(k-kl))] = rflx;
        }
      }
    } else if (n < 16) {                                      /* x2 faces :126-170 */
      const int j = jl, fj = 2*jl - cjs;
      for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) {
        const int fk = 2*k - cks, fi = 2*i - cis;
        double rflx;
        if (v == 0) {
          if (same) rflx = E1(e1,m,k,j,i);
          else if (s->two_d) rflx = 0.5*(E1(e1,m,0,fj,fi) + E1(e1,m,0,fj,fi+1));
          else rflx = 0.5*(E1(e1,m,fk,fj,fi) + E1(e1,m,fk,fj,fi+1));
          out[i-il + ni*(k-kl)] = rflx;
        } else if (v == 2) {
          if (same) rflx = E3(e3,m,k,j,i);
          else if (s->two_d) rflx = E3(e3,m,0,fj,fi);
          else rflx = 0.5*(E3(e3,m,fk,fj,fi) + E3(e3,m,fk+1,fj,fi));
          out[i-il + ni*(k-kl)] = rflx;
        }
      }
    } else if (n < 24) {                                      /* x1x2 edges :173-198 */
      const int i = il, j = jl, fi = 2*il - cis, fj = 2*jl - cjs;
      if (v == 2) for (int k = kl; k <= ku; ++k) {
        const int fk = 2*k - cks;
        double rflx;
        if (same) rflx = E3(e3,m,k,j,i);
        else if (s->two_d) rflx = E3(e3,m,0,fj,fi);
        else rflx = 0.5*(E3(e3,m,fk,fj,fi) + E3(e3,m,fk+1,fj,fi));
        out[(k-kl)] = rflx;
      }
    } else if (n < 32) {                                      /* x3 faces :201-236 */
      const int k = kl, fk = 2*kl - cks;
      for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
        const int fi = 2*i - cis, fj = 2*j - cjs;
        if (v == 0) out[i-il + ni*(j-jl)] = same ? E1(e1,m,k,j,i) : 0.5*(E1(e1,m,fk,fj,fi) + E1(e1,m,fk,fj,fi+1));
        else if (v == 1) out[i-il + ni*(j-jl)] = same ? E2(e2,m,k,j,i) : 0.5*(E2(e2,m,fk,fj,fi) + E2(e2,m,fk,fj+1,fi));
      }
    } else if (n < 40) {                                      /* x3x1 edges :239-260 */
      const int i = il, k = kl, fi = 2*il - cis, fk = 2*kl - cks;
      if (v == 1) for (int j = jl; j <= ju; ++j) {
        const int fj = 2*j - cjs;
        out[(j-jl)] = same ? E2(e2,m,k,j,i) : 0.5*(E2(e2,m,fk,fj,fi) + E2(e2,m,fk,fj+1,fi));
      }
    } else if (n < 48) {                                      /* x2x3 edges :263-284 */
      const int j = jl, k = kl, fj = 2*jl - cjs, fk = 2*kl - cks;
      if (v == 0) for (int i = il; i <= iu; ++i) {
        const int fi = 2*i - cis;
        out[i-il] = same ? E1(e1,m,k,j,i) : 0.5*(E1(e1,m,fk,fj,fi) + E1(e1,m,fk,fj,fi+1));
      }
    }
  }
}

/* SumBoundaryFluxes, flux_correct_fc.cpp:445-644 (no shearing box on this path) */
static void sum_boundary_fluxes(akref_smr *s, double *e1, double *e2, double *e3, int same_level, int *nflx) {
  for (int m = 0; m < s->nmb; ++m) for (int v = 0; v < 3; ++v) for (int n = 0; n < s->nnghbr; ++n) {
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    if (!(s->gid[NG(m,n)] >= 0 && ((same_level && nl == ml) || (!same_level && nl > ml)))) continue;
    const bb_t *rb = &s->fc.recvbuf[n];
    const bi_t *x = same_level ? &rb->iflux_same[v] : &rb->iflux_coar[v];
    const int ndat = same_level ? rb->iflxs_ndat : rb->iflxc_ndat;
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int ni = iu - il + 1, nj = ju - jl + 1;
    const double *in = rseg(s, 3, m, n, rb->flux, rb->fstride) + (size_t)ndat*v;
    int *nf = nflx + 48*m;
    if (n < 8) {
      if (v == 0) {
        if (n == 0) { nf[16] += 1; nf[20] += 1; nf[32] += 1; nf[36] += 1; }
        if (n == 4) { nf[18] += 1; nf[22] += 1; nf[34] += 1; nf[38] += 1; }
      } else {
        for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) {
          if (v == 1) E2(e2,m,k,j,il) += in[(j-jl + nj*(k-kl))];
          else E3(e3,m,k,j,il) += in[(j-jl + nj*(k-kl))];
        }
      }
    } else if (n < 16) {
      if (v == 0) {
        if (n == 8) { nf[16] += 1; nf[18] += 1; nf[40] += 1; nf[44] += 1; }
        if (n == 12) { nf[20] += 1; nf[22] += 1; nf[42] += 1; nf[46] += 1; }
      }
      for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) {
        if (v == 0) E1(e1,m,k,jl,i) += in[i-il + ni*(k-kl)];
        else if (v == 2) E3(e3,m,k,jl,i) += in[i-il + ni*(k-kl)];
      }
    } else if (n < 24) {
      if (v == 0) nf[n] += 1;
      else if (v == 2) for (int k = kl; k <= ku; ++k) E3(e3,m,k,jl,il) += in[(k-kl)];
    } else if (n < 32) {
      if (v == 0) {
        if (n == 24) { nf[32] += 1; nf[34] += 1; nf[40] += 1; nf[42] += 1; }
        if (n == 28) { nf[36] += 1; nf[38] += 1; nf[44] += 1; nf[46] += 1; }
      }
      for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
        if (v == 0) E1(e1,m,kl,j,i) += in[i-il + ni*(j-jl)];
        else if (v == 1) E2(e2,m,kl,j,i) += in[i-il + ni*(j-jl)];
      }
    } else if (n < 40) {
      if (v == 0) nf[n] += 1;
      else if (v == 1) for (int j = jl; j <= ju; ++j) E2(e2,m,kl,j,il) += in[(j-jl)];
    } else if (n < 48) {
      if (v == 0) {
        nf[n] += 1;
        for (int i = il; i <= iu; ++i) E1(e1,m,kl,jl,i) += in[i-il];
      }
    }
  }
}

/* ZeroFluxesAtBoundaryWithFiner, flux_correct_fc.cpp:655-790 */
static void zero_fluxes_at_finer(akref_smr *s, double *e1, double *e2, double *e3, int *nflx) {
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < 3; ++v) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] > s->mblev[m])) continue;
    const bi_t *x = &s->fc.recvbuf[n].iflux_coar[v];
    const int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    int *nf = nflx + 48*m;
    if (n < 8) {
      if (v == 0) {
        if (n == 0) { nf[16] = 0; nf[20] = 0; nf[32] = 0; nf[36] = 0; }
        if (n == 4) { nf[18] = 0; nf[22] = 0; nf[34] = 0; nf[38] = 0; }
      } else {
        for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) {
          if (v == 1) E2(e2,m,k,j,il) = 0.0; else E3(e3,m,k,j,il) = 0.0;
        }
      }
    } else if (n < 16) {
      if (v == 1) {
        if (n == 8) { nf[16] = 0; nf[18] = 0; nf[40] = 0; nf[44] = 0; }
        if (n == 12) { nf[20] = 0; nf[22] = 0; nf[42] = 0; nf[46] = 0; }
      } else {
        for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) {
          if (v == 0) E1(e1,m,k,jl,i) = 0.0; else E3(e3,m,k,jl,i) = 0.0;
        }
      }
    } else if (n < 24) {
      if (v == 2) { nf[n] = 0; for (int k = kl; k <= ku; ++k) E3(e3,m,k,jl,il) = 0.0; }
    } else if (n < 32) {
      if (v == 2) {
        if (n == 24) { nf[32] = 0; nf[34] = 0; nf[40] = 0; nf[42] = 0; }
        if (n == 28) { nf[36] = 0; nf[38] = 0; nf[44] = 0; nf[46] = 0; }
      } else {
        for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
          if (v == 0) E1(e1,m,kl,j,i) = 0.0; else E2(e2,m,kl,j,i) = 0.0;
        }
      }
    } else if (n < 40) {
      if (v == 1) { nf[n] = 0; for (int j = jl; j <= ju; ++j) E2(e2,m,kl,j,il) = 0.0; }
    } else if (n < 48) {
      if (v == 0) { nf[n] = 0; for (int i = il; i <= iu; ++i) E1(e1,m,kl,jl,i) = 0.0; }
    }
  }
}

/* AverageBoundaryFluxes, flux_correct_fc.cpp:801-1034 */
static void average_boundary_fluxes(akref_smr *s, double *e1, double *e2, double *e3, const int *nflx) {
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) for (int v = 0; v < 3; ++v) {
    if (!s->fc.recvbuf[n].used) continue;
    const bi_t *x = &s->fc.recvbuf[n].iflux_same[v];
    int il = x->bis, iu = x->bie, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    const int nl = s->lev[NG(m,n)], ml = s->mblev[m];
    const int *nf = nflx + 48*m;
    if (n == 0 || n == 4) {
      if (v == 1) {
        if (nl == ml) {
          if (s->three_d) { kl += 1; ku -= 1; }
          for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) E2(e2,m,k,j,il) *= 0.5;
        } else if (nl >= ml) {
          if (s->three_d) { const int k = kl + (ku - kl + 1)/2; for (int j = jl; j <= ju; ++j) E2(e2,m,k,j,il) *= 0.5; }
        }
      } else if (v == 2) {
        if (nl == ml) {
          if (s->multi_d) { jl += 1; ju -= 1; }
          for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) E3(e3,m,k,j,il) *= 0.5;
        } else if (nl >= ml) {
          if (s->multi_d) { const int j = jl + (ju - jl + 1)/2; for (int k = kl; k <= ku; ++k) E3(e3,m,k,j,il) *= 0.5; }
        }
      }
    } else if (s->multi_d && (n == 8 || n == 12)) {
      if (v == 0) {
        if (nl == ml) {
          if (s->three_d) { kl += 1; ku -= 1; }
          for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) E1(e1,m,k,jl,i) *= 0.5;
        } else if (nl >= ml) {
          if (s->three_d) { const int k = kl + (ku - kl + 1)/2; for (int i = il; i <= iu; ++i) E1(e1,m,k,jl,i) *= 0.5; }
        }
      } else if (v == 2) {
        if (nl == ml) {
          il += 1; iu -= 1;
          for (int k = kl; k <= ku; ++k) for (int i = il; i <= iu; ++i) E3(e3,m,k,jl,i) *= 0.5;
        } else if (nl >= ml) {
          const int i = il + (iu - il + 1)/2;
          for (int k = kl; k <= ku; ++k) E3(e3,m,k,jl,i) *= 0.5;
        }
      }
    } else if (s->multi_d && (n == 16 || n == 18 || n == 20 || n == 22)) {
      if (v == 2) for (int k = kl; k <= ku; ++k) E3(e3,m,k,jl,il) /= (double)nf[n];
    } else if (s->three_d && (n == 24 || n == 28)) {
      if (v == 0) {
        if (nl == ml) {
          jl += 1; ju -= 1;
          for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) E1(e1,m,kl,j,i) *= 0.5;
        } else if (nl >= ml) {
          const int j = jl + (ju - jl + 1)/2;
          for (int i = il; i <= iu; ++i) E1(e1,m,kl,j,i) *= 0.5;
        }
      } else if (v == 1) {
        if (nl == ml) {
          il += 1; iu -= 1;
          for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) E2(e2,m,kl,j,i) *= 0.5;
        } else if (nl >= ml) {
          const int i = il + (iu - il + 1)/2;
          for (int j = jl; j <= ju; ++j) E2(e2,m,kl,j,i) *= 0.5;
        }
      }
    } else if (s->three_d && (n == 32 || n == 34 || n == 36 || n == 38)) {
      if (v == 1) for (int j = jl; j <= ju; ++j) E2(e2,m,kl,j,il) /= (double)nf[n];
    } else if (s->three_d && (n == 40 || n == 42 || n == 44 || n == 46)) {
      if (v == 0) for (int i = il; i <= iu; ++i) E1(e1,m,kl,jl,i) /= (double)nf[n];
    }
  }
}

/* SendE + RecvE: PackAndSendFluxFC, then RecvAndUnpackFluxFC (flux_correct_fc.cpp:374-434) */
static int flux_fc_phase(akref_smr *s, double *e1, double *e2, double *e3, int phase) {
  if (phase & 1) send_flux_fc(s, e1, e2, e3);
  if (!(phase & 2)) return 0;
  int *nflx = (int *)malloc(sizeof(int)*48*s->nmb);
  for (int q = 0; q < 48*s->nmb; ++q) nflx[q] = 1;
  sum_boundary_fluxes(s, e1, e2, e3, 1, nflx);
  if (s->multilevel) {
    zero_fluxes_at_finer(s, e1, e2, e3, nflx);
    sum_boundary_fluxes(s, e1, e2, e3, 0, nflx);
  }
  average_boundary_fluxes(s, e1, e2, e3, nflx);
  free(nflx);
  return 0;
}
int akref_smr_flux_fc(akref_smr *s, double *e1, double *e2, double *e3) { return flux_fc_phase(s, e1, e2, e3, 3); }

/* ---- twins of the product's akmi_smr_* entry points (same arguments; the tables of the descriptor
 * other than the neighbour table and the levels are NOT read: the oracle builds its own index
 * ranges).  Used by the CPU stand-in backend of the host-logic tests (tests/cpu_backend.py). -------- */
static akref_smr *from_desc(const akmi_pack *p, const akmi_smr *t, int nvar) {
  return akref_smr_create(p, nvar, t->nghbr, t->mblev, t->multilevel);
}
int akref_smr_exchange_cc(const akmi_pack *p, const akmi_smr *t, int nvar, double *u, double *cu, double *buf) {
  (void)buf;
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_send_cc(s, u, cu);
  akref_smr_recv_cc(s, u, cu);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_exchange_fc(const akmi_pack *p, const akmi_smr *t, double *b1, double *b2, double *b3,
                          double *cb1, double *cb2, double *cb3, double *buf) {
  (void)buf;
  akref_smr *s = from_desc(p, t, 1);
  akref_smr_send_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_recv_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_fill_coarse_cc_t(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, double *cu) {
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_fill_coarse_cc(s, u, cu);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_fill_coarse_fc_t(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2,
                               const double *b3, double *cb1, double *cb2, double *cb3) {
  akref_smr *s = from_desc(p, t, 1);
  akref_smr_fill_coarse_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_destroy(s);
  return 0;
}
/* ---- <mesh_refinement>/prolong_primitives = true, src/bvals/prolong_prims.cpp ----------------------------- *
 * SingleC2P_IdealHyd (src/eos/ideal_c2p_hyd.hpp:22-66) / SingleC2P_IdealMHD (ideal_c2p_mhd.hpp:20-67) on one
 * state; u[0..4] = d,mx,my,mz,E (d is floored in place, as the reference's by-reference argument is),
 * bcc = NULL: hydro */
static void single_c2p_ideal(const akmi_pack *p, double *u, const double *bcc, double *w) {
  const double gm1 = p->gamma - 1.0;
  const double efloor = p->pfloor/gm1;
  double dfl = p->dfloor, e_m = 0.0;
  if (bcc) {
    const double b2 = bcc[0]*bcc[0] + bcc[1]*bcc[1] + bcc[2]*bcc[2];
    dfl = fmax(p->dfloor, b2/p->sigma_max);
    e_m = 0.5*(bcc[0]*bcc[0] + bcc[1]*bcc[1] + bcc[2]*bcc[2]);
  }
  if (u[0] < dfl) u[0] = dfl;
  w[0] = u[0];
  const double di = 1.0/u[0];
  w[1] = di*u[1]; w[2] = di*u[2]; w[3] = di*u[3];
  const double e_k = 0.5*di*(u[1]*u[1] + u[2]*u[2] + u[3]*u[3]);
  w[4] = bcc ? (u[4] - e_k - e_m) : (u[4] - e_k);
  if (w[4] < efloor) w[4] = efloor;
  if (gm1*w[4]*di < p->tfloor) w[4] = w[0]*p->tfloor/gm1;
  const double spe_over_eps = gm1/pow(w[0], gm1);
  const double spe = spe_over_eps*w[4]*di;
  if (spe <= p->sfloor) w[4] = w[0]*p->sfloor/spe_over_eps;
}
#define FC1(a,m,k,j,i) (a)[((((size_t)(m)*s->cN3 + (k))*s->cN2 + (j))*(s->cN1 + 1) + (i))]
#define FC2(a,m,k,j,i) (a)[((((size_t)(m)*s->cN3 + (k))*(s->cN2 + 1) + (j))*s->cN1 + (i))]
#define FC3(a,m,k,j,i) (a)[((((size_t)(m)*(s->cN3 + 1) + (k))*s->cN2 + (j))*s->cN1 + (i))]
#define FF1(a,m,k,j,i) (a)[((((size_t)(m)*s->N3 + (k))*s->N2 + (j))*(s->N1 + 1) + (i))]
#define FF2(a,m,k,j,i) (a)[((((size_t)(m)*s->N3 + (k))*(s->N2 + 1) + (j))*s->N1 + (i))]
#define FF3(a,m,k,j,i) (a)[((((size_t)(m)*(s->N3 + 1) + (k))*s->N2 + (j))*s->N1 + (i))]
/* ConsToPrimCoarseBndry, prolong_prims.cpp:35-186 (hydro), 303-461 (MHD) */
int akref_smr_c2p_coarse(akref_smr *s, const akmi_pack *p, double *cu, const double *cb1, const double *cb2,
                         const double *cb3, double *cw) {
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bi_t *x = &s->cc.recvbuf[n].iprol[0];
    int il = x->bis - 1, iu = x->bie + 1, jl = x->bjs, ju = x->bje, kl = x->bks, ku = x->bke;
    if (s->multi_d) { jl -= 1; ju += 1; }
    if (s->three_d) { kl -= 1; ku += 1; }
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      double u[5], w[5], bcc[3];
      for (int v = 0; v < 5; ++v) u[v] = C5(cu,nvar,m,v,k,j,i);
      if (cb1) {
        bcc[0] = 0.5*(FC1(cb1,m,k,j,i) + FC1(cb1,m,k,j,i+1));
        bcc[1] = 0.5*(FC2(cb2,m,k,j,i) + FC2(cb2,m,k,j+1,i));
        bcc[2] = 0.5*(FC3(cb3,m,k,j,i) + FC3(cb3,m,k+1,j,i));
      }
      single_c2p_ideal(p, u, cb1 ? bcc : NULL, w);
      for (int v = 0; v < 5; ++v) C5(cw,nvar,m,v,k,j,i) = w[v];
      for (int v = 5; v < nvar; ++v) {
        if (C5(cu,nvar,m,v,k,j,i) < 0.0) C5(cu,nvar,m,v,k,j,i) = 0.0;
        C5(cw,nvar,m,v,k,j,i) = C5(cu,nvar,m,v,k,j,i)/u[0];
      }
    }
  }
  return 0;
}
/* PrimToConsFineBndry, prolong_prims.cpp:190-296 (hydro), 465-575 (MHD); SingleP2C_IdealHyd / _IdealMHD */
int akref_smr_p2c_fine(akref_smr *s, const double *w, const double *b1, const double *b2, const double *b3,
                       double *u) {
  const int nvar = s->cc.nvar;
  for (int m = 0; m < s->nmb; ++m) for (int n = 0; n < s->nnghbr; ++n) {
    if (!(s->gid[NG(m,n)] >= 0 && s->lev[NG(m,n)] < s->mblev[m])) continue;
    const bi_t *x = &s->cc.recvbuf[n].iprol[0];
    const int il = (x->bis - s->cis)*2 + s->is, iu = (x->bie - s->cis)*2 + s->is + 1;
    const int jl = (x->bjs - s->cjs)*2 + s->js, ju = (x->bje - s->cjs)*2 + s->js + (s->multi_d ? 1 : 0);
    const int kl = (x->bks - s->cks)*2 + s->ks, ku = (x->bke - s->cks)*2 + s->ks + (s->three_d ? 1 : 0);
    for (int k = kl; k <= ku; ++k) for (int j = jl; j <= ju; ++j) for (int i = il; i <= iu; ++i) {
      const double d = A5(w,nvar,m,0,k,j,i), vx = A5(w,nvar,m,1,k,j,i), vy = A5(w,nvar,m,2,k,j,i),
                   vz = A5(w,nvar,m,3,k,j,i), e = A5(w,nvar,m,4,k,j,i);
      A5(u,nvar,m,0,k,j,i) = d; A5(u,nvar,m,1,k,j,i) = d*vx; A5(u,nvar,m,2,k,j,i) = d*vy; A5(u,nvar,m,3,k,j,i) = d*vz;
      if (b1) {
        const double bx = 0.5*(FF1(b1,m,k,j,i) + FF1(b1,m,k,j,i+1));
        const double by = 0.5*(FF2(b2,m,k,j,i) + FF2(b2,m,k,j+1,i));
        const double bz = 0.5*(FF3(b3,m,k,j,i) + FF3(b3,m,k+1,j,i));
        A5(u,nvar,m,4,k,j,i) = e + 0.5*(d*(vx*vx + vy*vy + vz*vz) + (bx*bx + by*by + bz*bz));
      } else {
        A5(u,nvar,m,4,k,j,i) = e + 0.5*d*(vx*vx + vy*vy + vz*vz);
      }
      for (int v = 5; v < nvar; ++v) A5(u,nvar,m,v,k,j,i) = d*A5(w,nvar,m,v,k,j,i);
    }
  }
  return 0;
}
int akref_smr_c2p_coarse_t(const akmi_pack *p, const akmi_smr *t, int nvar, double *cu, const double *cb1,
                           const double *cb2, const double *cb3, double *cw) {
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_c2p_coarse(s, p, cu, cb1, cb2, cb3, cw);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_p2c_fine_t(const akmi_pack *p, const akmi_smr *t, int nvar, const double *w, const double *b1,
                         const double *b2, const double *b3, double *u) {
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_p2c_fine(s, w, b1, b2, b3, u);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_prolong_cc_t(const akmi_pack *p, const akmi_smr *t, int nvar, const double *cu, double *u) {
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_prolong_cc(s, u, cu);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_prolong_fc_t(const akmi_pack *p, const akmi_smr *t, const double *cb1, const double *cb2,
                           const double *cb3, double *b1, double *b2, double *b3) {
  akref_smr *s = from_desc(p, t, 1);
  akref_smr_prolong_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_flux_cc_t(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, double *flx1,
                        double *flx2, double *flx3, double *buf) {
  (void)buf;
  akref_smr *s = from_desc(p, t, nvar);
  akref_smr_flux_cc(s, flx1, flx2, flx3, face_shaped);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_emf_exchange(const akmi_pack *p, const akmi_smr *t, const int *nflx, double *e1, double *e2,
                           double *e3, double *buf) {
  (void)buf; (void)nflx;
  akref_smr *s = from_desc(p, t, 1);
  akref_smr_flux_fc(s, e1, e2, e3);
  akref_smr_destroy(s);
  return 0;
}

/* the pack / unpack halves (include/akmi.h): the segments live in the caller's buffer */
static akref_smr *from_desc_x(const akmi_pack *p, const akmi_smr *t, int nvar, const double *buf) {
  akref_smr *s = from_desc(p, t, nvar);
  s->xbuf = (double *)buf; s->xlay = t->layout; s->soff = t->soff; s->roff = t->roff;
  return s;
}
int akref_smr_pack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, const double *cu, double *buf) {
  akref_smr *s = from_desc_x(p, t, nvar, buf);
  akref_smr_send_cc(s, u, cu);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_unpack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *buf, double *u, double *cu) {
  akref_smr *s = from_desc_x(p, t, nvar, buf);
  akref_smr_recv_cc(s, u, cu);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_pack_fc(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2, const double *b3,
                      const double *cb1, const double *cb2, const double *cb3, double *buf) {
  akref_smr *s = from_desc_x(p, t, 1, buf);
  akref_smr_send_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_unpack_fc(const akmi_pack *p, const akmi_smr *t, const double *buf, double *b1, double *b2, double *b3,
                        double *cb1, double *cb2, double *cb3) {
  akref_smr *s = from_desc_x(p, t, 1, buf);
  akref_smr_recv_fc(s, b1, b2, b3, cb1, cb2, cb3);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_pack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *flx1,
                           const double *flx2, const double *flx3, double *buf) {
  akref_smr *s = from_desc_x(p, t, nvar, buf);
  flux_cc_phase(s, (double *)flx1, (double *)flx2, (double *)flx3, face_shaped, 1);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_unpack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *buf,
                             double *flx1, double *flx2, double *flx3) {
  akref_smr *s = from_desc_x(p, t, nvar, buf);
  flux_cc_phase(s, flx1, flx2, flx3, face_shaped, 2);
  akref_smr_destroy(s);
  return 0;
}

int akref_smr_pack_emf(const akmi_pack *p, const akmi_smr *t, const double *e1, const double *e2, const double *e3,
                       double *buf) {
  akref_smr *s = from_desc_x(p, t, 1, buf);
  flux_fc_phase(s, (double *)e1, (double *)e2, (double *)e3, 1);
  akref_smr_destroy(s);
  return 0;
}
int akref_smr_unpack_emf(const akmi_pack *p, const akmi_smr *t, const int *nflx, const double *buf, double *e1,
                         double *e2, double *e3) {
  (void)nflx;
  akref_smr *s = from_desc_x(p, t, 1, buf);
  flux_fc_phase(s, e1, e2, e3, 2);
  akref_smr_destroy(s);
  return 0;
}
