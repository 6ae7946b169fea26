// akmi_smr.hip -- boundary values of a statically refined MeshBlockPack: the
// level-aware exchange of cell- and face-centred variables through the 56 buffer slots of a
// MeshBlock, the coarse-buffer fill and ghost prolongation, and the flux / EMF correction at
// fine/coarse boundaries.  Replaces the bodies of
//   MeshBoundaryValuesCC::PackAndSendCC / RecvAndUnpackCC      src/bvals/bvals_cc.cpp:42-447
//   MeshBoundaryValuesFC::PackAndSendFC / RecvAndUnpackFC      src/bvals/bvals_fc.cpp:63-436
//   FillCoarseInBndryCC/FC, ProlongateCC/FC                     src/bvals/prolongation.cpp:366-785
//   PackAndSendFluxCC / RecvAndUnpackFluxCC                     src/bvals/flux_correct_cc.cpp:29-304
//   PackAndSendFluxFC / RecvAndUnpackFluxFC (+Sum/Zero/Average) src/bvals/flux_correct_fc.cpp:29-1034
// Everything is table driven: the host hands over the index boxes of every slot (the reference's
// MeshBoundaryBuffer::isame/icoar/ifine/iprol/iflux_*), the neighbour table with levels and the
// buffer layout; a workgroup owns one (block, slot[, variable]) and strides over the box.  Where
// the reference orders overlapping writes (face fields: slots of a block are unpacked one after
// the other), a workgroup owns (block, component) and walks the slots with a barrier in between.
#include <vector>
#include "akmi_common.hpp"

namespace akmi {

struct SGeo {
  int N1, N2, N3, cN1, cN2, cN3;
  int is, ie, js, je, ks, ke, cis, cjs, cks;
  int multi_d, three_d;
};
static SGeo make_sgeo(const akmi_pack *p) {
  Geo g = make_geo(p);
  SGeo s;
  s.N1 = g.N1; s.N2 = g.N2; s.N3 = g.N3;
  const int cnx1 = g.nx1/2, cnx2 = g.multi_d ? g.nx2/2 : 1, cnx3 = g.three_d ? g.nx3/2 : 1;
  s.cN1 = cnx1 + 2*g.ng; s.cN2 = g.multi_d ? cnx2 + 2*g.ng : 1; s.cN3 = g.three_d ? cnx3 + 2*g.ng : 1;
  s.is = g.is; s.ie = g.ie; s.js = g.js; s.je = g.je; s.ks = g.ks; s.ke = g.ke;
  s.cis = g.ng; s.cjs = g.multi_d ? g.ng : 0; s.cks = g.three_d ? g.ng : 0;
  s.multi_d = g.multi_d; s.three_d = g.three_d;
  return s;
}

struct Bx { int il, iu, jl, ju, kl, ku; };
enum { K_SAME = 0, K_COAR = 1, K_FINE = 2, K_PROL = 3, K_FLXS = 4, K_FLXC = 5 };
enum { T_SEND = 0, T_RECV = 1 };
__device__ __forceinline__ Bx box_of(const int *tab, int sr, int kind, int n, int v) {
  const int *q = tab + ((((size_t)sr*6 + kind)*56 + n)*3 + v)*6;
  return Bx{q[0], q[1], q[2], q[3], q[4], q[5]};
}
__device__ __forceinline__ int bcount(const Bx &b) {
  return (b.iu - b.il + 1)*(b.ju - b.jl + 1)*(b.ku - b.kl + 1);
}
// element e of a box -> (k,j,i), i fastest
__device__ __forceinline__ void bdecode(const Bx &b, int e, int &k, int &j, int &i) {
  const int ni = b.iu - b.il + 1, nj = b.ju - b.jl + 1;
  i = b.il + e%ni; e /= ni;
  j = b.jl + e%nj;
  k = b.kl + e/nj;
}

// addressing of fine / coarse cell and face arrays
__device__ __forceinline__ size_t c5(const SGeo &s, int coarse, int nv, int m, int v, int k, int j, int i) {
  return coarse ? ix5(nv, s.cN3, s.cN2, s.cN1, m, v, k, j, i) : ix5(nv, s.N3, s.N2, s.N1, m, v, k, j, i);
}
__device__ __forceinline__ size_t f4(const SGeo &s, int coarse, int v, int m, int k, int j, int i) {
  const int n1 = coarse ? s.cN1 : s.N1, n2 = coarse ? s.cN2 : s.N2, n3 = coarse ? s.cN3 : s.N3;
  return ix4(n3 + (v == 2), n2 + (v == 1), n1 + (v == 0), m, k, j, i);
}
struct F3 { double *b[3]; };
struct CF3 { const double *b[3]; };

struct Tab {                      // device view of akmi_smr
  int nnghbr, multilevel;
  const int *ng, *lev, *cc, *fc, *ndat;
  const long long *lay;
  const long long *soff, *roff;   // [4][nmb][56] or null: where a segment is written / read (ranks)
  int nmb;
  int direct;                     // akmi_smr::direct_same
  const unsigned char *needs;     // akmi_smr::needs_coarse
  const int *lists;               // akmi_smr::lists (work lists of (block, slot) pairs) or null
  int cnt[AKMI_SMR_NLISTS];       // akmi_smr::list_cnt
  int lstride;                    // ints between two lists: 2*nmb*56
};
static Tab make_tab(const akmi_pack *p, const akmi_smr *t) {
  Tab tb{t->nnghbr, t->multilevel, t->nghbr, t->mblev, t->cc_tab, t->fc_tab, t->ndat, t->layout,
         t->soff, t->roff, p->nmb, t->direct_same, t->needs_coarse, t->lists, {0}, 2*p->nmb*56};
  for (int l = 0; l < AKMI_SMR_NLISTS; ++l) tb.cnt[l] = t->list_cnt[l];
  return tb;
}
// Which (MeshBlock, slot) a workgroup works on.  The kernels below are launched over (block, slot[, component]); on a
// refined mesh most of those pairs have nothing to do -- 27 of 56 slots of a block exist, a few per cent of them cross a
// level -- and a launch of 161 000 workgroups that return at once still costs their dispatch (k_smr_average_flux_fc:
// 248 us for a few MB).  With akmi_smr::lists (akmi_smr_build_lists) a launch covers only the pairs of the list that
// is a superset of the kernel's own test; every kernel keeps that test, so a list can only cost time, never a result.
enum { L_VALID = 0, L_VALID_CC = 1, L_COARSER = 2, L_FINER = 3, L_SAME_NEEDS = 4, L_AVG = 5 };
__device__ __forceinline__ void wg_slot(const Tab &t, int L, int per, int &m, int &n, int &v) {
  const int w = blockIdx.x/per;
  v = blockIdx.x - w*per;
  if (t.lists) { const int *q = t.lists + (size_t)L*t.lstride + 2*(size_t)w; m = q[0]; n = q[1]; }
  else { n = w%t.nnghbr; m = w/t.nnghbr; }
}
static unsigned wg_count(const Tab &t, int L) { return t.lists ? (unsigned)t.cnt[L] : (unsigned)t.nmb*t.nnghbr; }
#define NGID(t, m, n) (t).ng[((size_t)(m)*56 + (n))*3]
#define NLEV(t, m, n) (t).ng[((size_t)(m)*56 + (n))*3 + 1]
#define NDST(t, m, n) (t).ng[((size_t)(m)*56 + (n))*3 + 2]
// buffer of (block m, slot n): cls 0 cc vars, 1 cc flux, 2 fc vars, 3 fc flux
__device__ __forceinline__ size_t buf_at(const Tab &t, int cls, int m, int n) {
  const long long *l = t.lay + ((size_t)cls*56 + n)*2;
  return (size_t)l[0] + (size_t)m*(size_t)l[1];
}
// where block m writes the segment of its slot n (into the receive buffer (dm, dn) of a neighbour in
// this pack, or into the message to another rank) and where it reads the segment it receives
__device__ __forceinline__ size_t seg_w(const Tab &t, int cls, int m, int n, int dm, int dn) {
  return t.soff ? (size_t)t.soff[((size_t)cls*t.nmb + m)*56 + n] : buf_at(t, cls, dm, dn);
}
__device__ __forceinline__ size_t seg_r(const Tab &t, int cls, int m, int n) {
  return t.roff ? (size_t)t.roff[((size_t)cls*t.nmb + m)*56 + n] : buf_at(t, cls, m, n);
}
// ndat[fc][slot][send|recv][same, coar, fine, flxs, flxc]
__device__ __forceinline__ int ndat_of(const Tab &t, int fc, int n, int sr, int q) {
  return t.ndat[(((size_t)fc*56 + n)*2 + sr)*5 + q];
}

// ---- PackAndSendCC: block m writes into the receive buffer (dm, dn) of its neighbour -------------
__global__ void __launch_bounds__(256)
k_smr_pack_cc(SGeo s, Tab t, int nvar, const double *__restrict__ a, const double *__restrict__ ca,
              double *__restrict__ buf) {
  // one workgroup per (block, slot), all variables (a fifth of the workgroups of the (block, slot, variable) grid:
  // most slots of a block are empty and the launch itself was a third of the kernel's time)
  int m, n, v; wg_slot(t, L_VALID_CC, 1, m, n, v); (void)v;
  const int dm = NGID(t, m, n);
  if (dm < 0) return;
  const int nl = NLEV(t, m, n), ml = t.lev[m];
  if (t.direct && nl == ml && dm < t.nmb) return;          // filled by the caller's direct gather
  const Bx b = box_of(t.cc, T_SEND, nl < ml ? K_COAR : (nl == ml ? K_SAME : K_FINE), n, 0);
  const int cnt = bcount(b);
  double *out = buf + seg_w(t, 0, m, n, dm, NDST(t, m, n));
  const int coarse = nl < ml;
  const double *src = coarse ? ca : a;
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    for (int v = 0; v < nvar; ++v) out[(size_t)cnt*v + e] = src[c5(s, coarse, nvar, m, v, k, j, i)];
  }
}

// ---- RecvAndUnpackCC ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_smr_unpack_cc(SGeo s, Tab t, int nvar, const double *__restrict__ buf, double *__restrict__ a,
                double *__restrict__ ca) {
  int m, n, v; wg_slot(t, L_VALID_CC, 1, m, n, v); (void)v;
  if (NGID(t, m, n) < 0) return;
  const int nl = NLEV(t, m, n), ml = t.lev[m];
  if (t.direct && nl == ml && NGID(t, m, n) < t.nmb) return;
  const Bx b = box_of(t.cc, T_RECV, nl < ml ? K_COAR : (nl == ml ? K_SAME : K_FINE), n, 0);
  const int cnt = bcount(b);
  const double *in = buf + seg_r(t, 0, m, n);
  const int coarse = nl < ml;
  double *dst = coarse ? ca : a;
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    for (int v = 0; v < nvar; ++v) dst[c5(s, coarse, nvar, m, v, k, j, i)] = in[(size_t)cnt*v + e];
  }
}

// ---- PackAndSendFC --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_smr_pack_fc(SGeo s, Tab t, CF3 b, CF3 cb, double *__restrict__ buf) {
  int m, n, v; wg_slot(t, L_VALID, 3, m, n, v); (void)v;
  const int dm = NGID(t, m, n);
  if (dm < 0) return;
  const int nl = NLEV(t, m, n), ml = t.lev[m];
  const int q = nl < ml ? 1 : (nl == ml ? 0 : 2);
  const Bx bx = box_of(t.fc, T_SEND, q == 1 ? K_COAR : (q == 0 ? K_SAME : K_FINE), n, v);
  const int cnt = bcount(bx);
  const int dn = NDST(t, m, n);
  // the receiver unpacks with ITS ndat of slot dn and the matching level relation
  const int rq = q == 1 ? 2 : (q == 2 ? 1 : 0);
  double *out = buf + seg_w(t, 2, m, n, dm, dn) + (size_t)ndat_of(t, 1, dn, T_RECV, rq)*v;
  const int coarse = nl < ml;
  const double *src = coarse ? cb.b[v] : b.b[v];
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(bx, e, k, j, i);
    out[e] = src[f4(s, coarse, v, m, k, j, i)];
  }
}

__device__ __forceinline__ bool active_face(const SGeo &s, int v, int k, int j, int i) {   // IsActiveFCFace
  return i >= s.is && i <= s.ie + (v == 0) && j >= s.js && j <= s.je + (v == 1) && k >= s.ks && k <= s.ke + (v == 2);
}

// ---- RecvAndUnpackFC: the slots of a (block, component) one after the other ----------------------
// The slots are unpacked in the reference's order (regions of different slots overlap on refined meshes and the
// later one wins), so a workgroup walks them with a barrier in between.  What a slot needs from the tables -- its box,
// where its segment starts, which array it goes to -- is fetched for all slots at once into LDS first: inside the
// walk those three dependent global look-ups per slot were most of the kernel's time (27 slots x ~2 us for a few KB).
struct SlotMeta { Bx b; long long in; int q; };     // q: -1 no neighbour, 0 same level, 1 coarser, 2 finer
__global__ void __launch_bounds__(256)
k_smr_unpack_fc(SGeo s, Tab t, const double *__restrict__ buf, F3 b, F3 cb) {
  const int v = blockIdx.x%3, m = blockIdx.x/3;
  const int ml = t.lev[m];
  __shared__ SlotMeta sm[56];
  if ((int)threadIdx.x < t.nnghbr) {
    const int n = threadIdx.x;
    SlotMeta x;
    x.q = -1; x.in = 0; x.b = Bx{0, -1, 0, -1, 0, -1};
    if (NGID(t, m, n) >= 0) {
      const int nl = NLEV(t, m, n);
      x.q = nl < ml ? 1 : (nl == ml ? 0 : 2);
      x.b = box_of(t.fc, T_RECV, x.q == 1 ? K_COAR : (x.q == 0 ? K_SAME : K_FINE), n, v);
      x.in = (long long)seg_r(t, 2, m, n) + (long long)ndat_of(t, 1, n, T_RECV, x.q)*v;
    }
    sm[n] = x;
  }
  __syncthreads();
  // (an order-preserving PARALLEL walk -- round numbers from box intersections, all slots of a round at once -- was
  //  built and measured: planning the rounds in the kernel costs more than the walk saves, 43 -> 112 us at deck
  //  size; profiles/r03_config5.txt)
  for (int n = 0; n < t.nnghbr; ++n) {
    const int q = sm[n].q;
    if (q < 0) continue;                                   // uniform over the workgroup
    const Bx bx = sm[n].b;
    const int cnt = bcount(bx);
    const double *in = buf + sm[n].in;
    const int coarse = q == 1;
    double *dst = coarse ? cb.b[v] : b.b[v];
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
      int k, j, i;
      bdecode(bx, e, k, j, i);
      if (!coarse && active_face(s, v, k, j, i)) continue;
      dst[f4(s, coarse, v, m, k, j, i)] = in[e];
    }
    __syncthreads();
  }
}

// ---- FillCoarseInBndryCC --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_smr_fill_coarse_cc(SGeo s, Tab t, int nvar, const double *__restrict__ a, double *__restrict__ ca) {
  int m, n, v; wg_slot(t, L_SAME_NEEDS, nvar, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || NLEV(t, m, n) != t.lev[m] || (t.needs && !t.needs[m])) return;
  const Bx r = box_of(t.cc, T_RECV, K_SAME, n, 0);
  const Bx b{(r.il + s.cis)/2, (r.iu + s.cis)/2, (r.jl + s.cjs)/2, (r.ju + s.cjs)/2, (r.kl + s.cks)/2,
             (r.ku + s.cks)/2};
  const int cnt = bcount(b);
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    const int fi = (i - s.cis)*2 + s.is, fj = (j - s.cjs)*2 + s.js, fk = (k - s.cks)*2 + s.ks;
    auto A = [&](int kk, int jj, int ii) { return a[c5(s, 0, nvar, m, v, kk, jj, ii)]; };
    if (!s.three_d)
      ca[c5(s, 1, nvar, m, v, b.kl, j, i)] = 0.25*(A(b.kl, fj, fi) + A(b.kl, fj, fi + 1) + A(b.kl, fj + 1, fi) + A(b.kl, fj + 1, fi + 1));
    else
      ca[c5(s, 1, nvar, m, v, k, j, i)] = 0.125*(A(fk, fj, fi) + A(fk, fj, fi + 1) + A(fk, fj + 1, fi) + A(fk, fj + 1, fi + 1)
                                               + A(fk + 1, fj, fi) + A(fk + 1, fj, fi + 1) + A(fk + 1, fj + 1, fi) + A(fk + 1, fj + 1, fi + 1));
  }
}

// ---- FillCoarseInBndryFC --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_smr_fill_coarse_fc(SGeo s, Tab t, CF3 b, F3 cb) {
  int m, n, v; wg_slot(t, L_SAME_NEEDS, 3, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || NLEV(t, m, n) != t.lev[m] || (t.needs && !t.needs[m])) return;
  const Bx r = box_of(t.fc, T_RECV, K_SAME, n, v);
  const Bx bx{(r.il + s.cis)/2, (r.iu + s.cis)/2, (r.jl + s.cjs)/2, (r.ju + s.cjs)/2, (r.kl + s.cks)/2,
              (r.ku + s.cks)/2};
  const int cnt = bcount(bx);
  auto B = [&](int kk, int jj, int ii) { return b.b[v][f4(s, 0, v, m, kk, jj, ii)]; };
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(bx, e, k, j, i);
    const int fi = (i - s.cis)*2 + s.is, fj = (j - s.cjs)*2 + s.js, fk = (k - s.cks)*2 + s.ks;
    const int kl = bx.kl;
    if (!s.three_d) {
      if (v == 0) cb.b[0][f4(s, 1, 0, m, kl, j, i)] = 0.5*(B(kl, fj, fi) + B(kl, fj + 1, fi));
      else if (v == 1) cb.b[1][f4(s, 1, 1, m, kl, j, i)] = 0.5*(B(kl, fj, fi) + B(kl, fj, fi + 1));
      else {
        const double b3c = 0.25*(B(kl, fj, fi) + B(kl, fj, fi + 1) + B(kl, fj + 1, fi) + B(kl, fj + 1, fi + 1));
        cb.b[2][f4(s, 1, 2, m, kl, j, i)] = b3c;
        cb.b[2][f4(s, 1, 2, m, kl + 1, j, i)] = b3c;
      }
    } else {
      double r4;
      if (v == 0) r4 = 0.25*(B(fk, fj, fi) + B(fk, fj + 1, fi) + B(fk + 1, fj, fi) + B(fk + 1, fj + 1, fi));
      else if (v == 1) r4 = 0.25*(B(fk, fj, fi) + B(fk, fj, fi + 1) + B(fk + 1, fj, fi) + B(fk + 1, fj, fi + 1));
      else r4 = 0.25*(B(fk, fj, fi) + B(fk, fj, fi + 1) + B(fk, fj + 1, fi) + B(fk, fj + 1, fi + 1));
      cb.b[v][f4(s, 1, v, m, k, j, i)] = r4;
    }
  }
}

__device__ __forceinline__ double s_sgn(double x) { return (x < 0.0) ? -1.0 : 1.0; }        // SIGN, athena.hpp:52
__device__ __forceinline__ double s_mm8(double dl, double dr) {
  return 0.125*(s_sgn(dl) + s_sgn(dr))*fmin(fabs(dl), fabs(dr));
}

// ---- ProlongateCC (ProlongCC, src/mesh/prolongation.hpp:19-63) -----------------------------------
__global__ void __launch_bounds__(256)
k_smr_prolong_cc(SGeo s, Tab t, int nvar, const double *__restrict__ ca, double *__restrict__ a) {
  int m, n, v; wg_slot(t, L_COARSER, 1, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  const Bx b = box_of(t.cc, T_RECV, K_PROL, n, 0);
  const int cnt = bcount(b);
  for (int ev = threadIdx.x; ev < cnt*nvar; ev += blockDim.x) {
    const int v = ev/cnt, e = ev - v*cnt;
    auto CA = [&](int kk, int jj, int ii) { return ca[c5(s, 1, nvar, m, v, kk, jj, ii)]; };
    auto A = [&](int kk, int jj, int ii) -> double & { return a[c5(s, 0, nvar, m, v, kk, jj, ii)]; };
    int k, j, i;
    bdecode(b, e, k, j, i);
    const int fi = (i - s.cis)*2 + s.is, fj = (j - s.cjs)*2 + s.js, fk = (k - s.cks)*2 + s.ks;
    const double q = CA(k, j, i);
    const double dvar1 = s_mm8(q - CA(k, j, i - 1), CA(k, j, i + 1) - q);
    double dvar2 = 0.0, dvar3 = 0.0;
    if (s.multi_d) dvar2 = s_mm8(q - CA(k, j - 1, i), CA(k, j + 1, i) - q);
    if (s.three_d) dvar3 = s_mm8(q - CA(k - 1, j, i), CA(k + 1, j, i) - q);
    A(fk, fj, fi) = q - dvar1 - dvar2 - dvar3;
    A(fk, fj, fi + 1) = q + dvar1 - dvar2 - dvar3;
    if (s.multi_d) {
      A(fk, fj + 1, fi) = q - dvar1 + dvar2 - dvar3;
      A(fk, fj + 1, fi + 1) = q + dvar1 + dvar2 - dvar3;
    }
    if (s.three_d) {
      A(fk + 1, fj, fi) = q - dvar1 - dvar2 + dvar3;
      A(fk + 1, fj, fi + 1) = q + dvar1 - dvar2 + dvar3;
      A(fk + 1, fj + 1, fi) = q - dvar1 + dvar2 + dvar3;
      A(fk + 1, fj + 1, fi + 1) = q + dvar1 + dvar2 + dvar3;
    }
  }
}

// ---- <mesh_refinement>/prolong_primitives = true (src/bvals/prolong_prims.cpp) -------------------------
// ConsToPrimCoarseBndry (:35-186 hydro, :303-461 MHD): the coarse cells the prolongation stencil of slot n
// reads (iprol widened by one cell in every active direction) from conserved to primitive variables; the
// cell-centred field is the average of the COARSE face fields; floors act on the primitives only (the coarse
// conserved array is scratch, :164-165), except that negative passive scalars are zeroed in place (:173-176).
// Ideal gas (the reference calls SingleC2P_IdealHyd / _IdealMHD unconditionally).
__global__ void __launch_bounds__(256)
k_smr_c2p_coarse(SGeo s, Tab t, Eos eos, int nvar, int mhd, double *__restrict__ cu, CF3 cb, double *__restrict__ cw) {
  int m, n, v; wg_slot(t, L_COARSER, 1, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  Bx b = box_of(t.cc, T_RECV, K_PROL, n, 0);
  // (widened by one cell, as the reference does: the boxes of different slots of a block then overlap, and several
  //  workgroups convert the same coarse cells concurrently -- every one of them writes the SAME values (cw, and cu
  //  where a floor or a negative scalar was reset: functions of that cell's cu alone), so the race is benign)
  b.il -= 1; b.iu += 1;
  if (s.multi_d) { b.jl -= 1; b.ju += 1; }
  if (s.three_d) { b.kl -= 1; b.ku += 1; }
  const int cnt = bcount(b);
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    double ud = cu[c5(s, 1, nvar, m, 0, k, j, i)], ue = cu[c5(s, 1, nvar, m, 4, k, j, i)];
    const double umx = cu[c5(s, 1, nvar, m, 1, k, j, i)], umy = cu[c5(s, 1, nvar, m, 2, k, j, i)],
                 umz = cu[c5(s, 1, nvar, m, 3, k, j, i)];
    double wd, wx, wy, wz, we;
    bool f1 = false, f2 = false, f3 = false;
    if (mhd) {
      const double bx = 0.5*(cb.b[0][f4(s, 1, 0, m, k, j, i)] + cb.b[0][f4(s, 1, 0, m, k, j, i + 1)]);
      const double by = 0.5*(cb.b[1][f4(s, 1, 1, m, k, j, i)] + cb.b[1][f4(s, 1, 1, m, k, j + 1, i)]);
      const double bz = 0.5*(cb.b[2][f4(s, 1, 2, m, k, j, i)] + cb.b[2][f4(s, 1, 2, m, k + 1, j, i)]);
      c2p_mhd(eos, ud, umx, umy, umz, ue, bx, by, bz, wd, wx, wy, wz, we, f1, f2, f3);
    } else {
      c2p_hyd(eos, ud, umx, umy, umz, ue, wd, wx, wy, wz, we, f1, f2, f3);
    }
    cw[c5(s, 1, nvar, m, 0, k, j, i)] = wd; cw[c5(s, 1, nvar, m, 1, k, j, i)] = wx;
    cw[c5(s, 1, nvar, m, 2, k, j, i)] = wy; cw[c5(s, 1, nvar, m, 3, k, j, i)] = wz;
    cw[c5(s, 1, nvar, m, 4, k, j, i)] = we;
    for (int v = 5; v < nvar; ++v) {
      double sc = cu[c5(s, 1, nvar, m, v, k, j, i)];
      if (sc < 0.0) { sc = 0.0; cu[c5(s, 1, nvar, m, v, k, j, i)] = 0.0; }
      cw[c5(s, 1, nvar, m, v, k, j, i)] = sc/ud;
    }
  }
}

// PrimToConsFineBndry (:190-296 hydro, :465-575 MHD): the fine ghost cells slot n prolongated, back to conserved
// variables (SingleP2C_IdealHyd / _IdealMHD, src/eos/ideal_c2p_hyd.hpp:76-83, ideal_c2p_mhd.hpp:76-84); the
// cell-centred field is the average of the FINE face fields, which ProlongateFC has filled before.
__global__ void __launch_bounds__(256)
k_smr_p2c_fine(SGeo s, Tab t, int nvar, int mhd, const double *__restrict__ w, CF3 fb, double *__restrict__ u) {
  int m, n, v; wg_slot(t, L_COARSER, 1, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  const Bx c = box_of(t.cc, T_RECV, K_PROL, n, 0);
  Bx b;
  b.il = (c.il - s.cis)*2 + s.is; b.iu = (c.iu - s.cis)*2 + s.is + 1;
  b.jl = (c.jl - s.cjs)*2 + s.js; b.ju = (c.ju - s.cjs)*2 + s.js + (s.multi_d ? 1 : 0);
  b.kl = (c.kl - s.cks)*2 + s.ks; b.ku = (c.ku - s.cks)*2 + s.ks + (s.three_d ? 1 : 0);
  const int cnt = bcount(b);
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    const double d = w[c5(s, 0, nvar, m, 0, k, j, i)], vx = w[c5(s, 0, nvar, m, 1, k, j, i)],
                 vy = w[c5(s, 0, nvar, m, 2, k, j, i)], vz = w[c5(s, 0, nvar, m, 3, k, j, i)],
                 ei = w[c5(s, 0, nvar, m, 4, k, j, i)];
    u[c5(s, 0, nvar, m, 0, k, j, i)] = d;
    u[c5(s, 0, nvar, m, 1, k, j, i)] = d*vx;
    u[c5(s, 0, nvar, m, 2, k, j, i)] = d*vy;
    u[c5(s, 0, nvar, m, 3, k, j, i)] = d*vz;
    if (mhd) {
      const double bx = 0.5*(fb.b[0][f4(s, 0, 0, m, k, j, i)] + fb.b[0][f4(s, 0, 0, m, k, j, i + 1)]);
      const double by = 0.5*(fb.b[1][f4(s, 0, 1, m, k, j, i)] + fb.b[1][f4(s, 0, 1, m, k, j + 1, i)]);
      const double bz = 0.5*(fb.b[2][f4(s, 0, 2, m, k, j, i)] + fb.b[2][f4(s, 0, 2, m, k + 1, j, i)]);
      u[c5(s, 0, nvar, m, 4, k, j, i)] = ei + 0.5*(d*(sqr(vx) + sqr(vy) + sqr(vz)) + (sqr(bx) + sqr(by) + sqr(bz)));
    } else {
      u[c5(s, 0, nvar, m, 4, k, j, i)] = ei + 0.5*d*(sqr(vx) + sqr(vy) + sqr(vz));
    }
    for (int v = 5; v < nvar; ++v) u[c5(s, 0, nvar, m, v, k, j, i)] = d*w[c5(s, 0, nvar, m, v, k, j, i)];
  }
}

// offsets of slot n (the inverse of NeighborIndex, prolongation.cpp:28-53), passed as a table by the host
struct Owned {
  const SGeo &s; const Tab &t; int m, ox1, ox2, ox3, mylev;
  // level of the finest existing neighbour across the face (ox) of the block, -1 when none
  __device__ int maxlev(int a1, int a2, int a3) const {
    int mx = -1;
    // NeighborIndex of a face: x1 faces 0-7 (ix<0: 0..3, ix>0: 4..7), x2 faces 8-15, x3 faces 24-31
    const int base = a1 != 0 ? (a1 < 0 ? 0 : 4) : (a2 != 0 ? (a2 < 0 ? 8 : 12) : (a3 < 0 ? 24 : 28));
    for (int q = 0; q < 4; ++q) {
      const int idx = base + q;
      if (idx < t.nnghbr && NGID(t, m, idx) >= 0) { const int l = NLEV(t, m, idx); mx = l > mx ? l : mx; }
    }
    return mx;
  }
  // CanProlongateFCFace, prolongation.cpp:90-133
  __device__ bool can(int v, int k, int j, int i) const {
    if (!active_face(s, v, k, j, i)) return true;
    int nox;
    if (v == 0) {
      if (i == s.is) nox = -1; else if (i == s.ie + 1) nox = 1; else return false;
      return ox1 == nox && ox2 == 0 && ox3 == 0 && maxlev(nox, 0, 0) < mylev;
    } else if (v == 1) {
      if (j == s.js) nox = -1; else if (j == s.je + 1) nox = 1; else return false;
      return ox1 == 0 && ox2 == nox && ox3 == 0 && maxlev(0, nox, 0) < mylev;
    }
    if (k == s.ks) nox = -1; else if (k == s.ke + 1) nox = 1; else return false;
    return ox1 == 0 && ox2 == 0 && ox3 == nox && maxlev(0, 0, nox) < mylev;
  }
};

// ---- ProlongateFC, shared faces (ProlongFCSharedX1/2/3FaceOwned, prolongation.cpp:149-258) -------
__global__ void __launch_bounds__(256)
k_smr_prolong_fc_shared(SGeo s, Tab t, const int *__restrict__ slot_ox, CF3 cb, F3 b) {
  int m, n, v; wg_slot(t, L_COARSER, 3, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  const Owned own{s, t, m, slot_ox[3*n], slot_ox[3*n + 1], slot_ox[3*n + 2], t.lev[m]};
  const Bx bx = box_of(t.fc, T_RECV, K_PROL, n, v);
  const int cnt = bcount(bx);
  auto C = [&](int kk, int jj, int ii) { return cb.b[v][f4(s, 1, v, m, kk, jj, ii)]; };
  auto ST = [&](int kk, int jj, int ii, double val) {
    if (own.can(v, kk, jj, ii)) b.b[v][f4(s, 0, v, m, kk, jj, ii)] = val;
  };
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(bx, e, k, j, i);
    const int fi = (i - s.cis)*2 + s.is;
    const int fj = s.multi_d ? (j - s.cjs)*2 + s.js : j;
    const int fk = s.three_d ? (k - s.cks)*2 + s.ks : k;
    const double q = C(k, j, i);
    if (v == 0) {
      double dvar2 = 0.0, dvar3 = 0.0;
      if (s.multi_d) dvar2 = s_mm8(q - C(k, j - 1, i), C(k, j + 1, i) - q);
      if (s.three_d) dvar3 = s_mm8(q - C(k - 1, j, i), C(k + 1, j, i) - q);
      ST(fk, fj, fi, q - dvar2 - dvar3);
      if (s.multi_d) ST(fk, fj + 1, fi, q + dvar2 - dvar3);
      if (s.three_d) { ST(fk + 1, fj, fi, q - dvar2 + dvar3); ST(fk + 1, fj + 1, fi, q + dvar2 + dvar3); }
    } else if (v == 1) {
      const double dvar1 = s_mm8(q - C(k, j, i - 1), C(k, j, i + 1) - q);
      double dvar3 = 0.0;
      if (s.three_d) dvar3 = s_mm8(q - C(k - 1, j, i), C(k + 1, j, i) - q);
      ST(fk, fj, fi, q - dvar1 - dvar3);
      ST(fk, fj, fi + 1, q + dvar1 - dvar3);
      if (s.three_d) { ST(fk + 1, fj, fi, q - dvar1 + dvar3); ST(fk + 1, fj, fi + 1, q + dvar1 + dvar3); }
    } else {
      const double dvar1 = s_mm8(q - C(k, j, i - 1), C(k, j, i + 1) - q);
      double dvar2 = 0.0;
      if (s.multi_d) dvar2 = s_mm8(q - C(k, j - 1, i), C(k, j + 1, i) - q);
      ST(fk, fj, fi, q - dvar1 - dvar2);
      ST(fk, fj, fi + 1, q + dvar1 - dvar2);
      if (s.multi_d) { ST(fk, fj + 1, fi, q - dvar1 + dvar2); ST(fk, fj + 1, fi + 1, q + dvar1 + dvar2); }
    }
  }
}

// ---- ProlongateFC, faces inside the coarse cells (ProlongFCInternalOwned, :260-359) ---------------
__global__ void __launch_bounds__(256)
k_smr_prolong_fc_internal(SGeo s, Tab t, const int *__restrict__ slot_ox, F3 b) {
  int m, n, v; wg_slot(t, L_COARSER, 1, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  const Owned own{s, t, m, slot_ox[3*n], slot_ox[3*n + 1], slot_ox[3*n + 2], t.lev[m]};
  const Bx p0 = box_of(t.fc, T_RECV, K_PROL, n, 0), p1 = box_of(t.fc, T_RECV, K_PROL, n, 1),
           p2 = box_of(t.fc, T_RECV, K_PROL, n, 2);
  const Bx bx{p2.il, p2.iu, p0.jl, p0.ju, p1.kl, p1.ku};
  const int cnt = bcount(bx);
  auto B1 = [&](int kk, int jj, int ii) { return b.b[0][f4(s, 0, 0, m, kk, jj, ii)]; };
  auto B2 = [&](int kk, int jj, int ii) { return b.b[1][f4(s, 0, 1, m, kk, jj, ii)]; };
  auto B3 = [&](int kk, int jj, int ii) { return b.b[2][f4(s, 0, 2, m, kk, jj, ii)]; };
  auto ST = [&](int v, int kk, int jj, int ii, double val) {
    if (own.can(v, kk, jj, ii)) b.b[v][f4(s, 0, v, m, kk, jj, ii)] = val;
  };
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(bx, e, k, j, i);
    const int fi = (i - s.cis)*2 + s.is, fj = (j - s.cjs)*2 + s.js, fk = (k - s.cks)*2 + s.ks;
    if (!s.multi_d) {
      ST(0, fk, fj, fi + 1, 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2)));
    } else if (s.three_d) {
      double Uxx = 0.0, Vyy = 0.0, Wzz = 0.0, Uxyz = 0.0, Vxyz = 0.0, Wxyz = 0.0;
      for (int jj = 0; jj < 2; jj++) {
        const int jsgn = 2*jj - 1, fjj = fj + jj, fjp = fj + 2*jj;
        for (int ii = 0; ii < 2; ii++) {
          const int isgn = 2*ii - 1, fii = fi + ii, fip = fi + 2*ii;
          Uxx += isgn*(jsgn*(B2(fk, fjp, fii) + B2(fk + 1, fjp, fii)) + (B3(fk + 2, fjj, fii) - B3(fk, fjj, fii)));
          Vyy += jsgn*((B3(fk + 2, fjj, fii) - B3(fk, fjj, fii)) + isgn*(B1(fk, fjj, fip) + B1(fk + 1, fjj, fip)));
          Wzz += isgn*(B1(fk + 1, fjj, fip) - B1(fk, fjj, fip)) + jsgn*(B2(fk + 1, fjp, fii) - B2(fk, fjp, fii));
          Uxyz += isgn*jsgn*(B1(fk + 1, fjj, fip) - B1(fk, fjj, fip));
          Vxyz += isgn*jsgn*(B2(fk + 1, fjp, fii) - B2(fk, fjp, fii));
          Wxyz += isgn*jsgn*(B3(fk + 2, fjj, fii) - B3(fk, fjj, fii));
        }
      }
      Uxx *= 0.125; Vyy *= 0.125; Wzz *= 0.125;
      Uxyz *= 0.0625; Vxyz *= 0.0625; Wxyz *= 0.0625;
      // all operands are read before any store of this cell (the stores touch only faces strictly
      // inside the coarse cell, which no other cell reads)
      const double a00 = 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2)), a01 = 0.5*(B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2)),
                   a10 = 0.5*(B1(fk + 1, fj, fi) + B1(fk + 1, fj, fi + 2)), a11 = 0.5*(B1(fk + 1, fj + 1, fi) + B1(fk + 1, fj + 1, fi + 2));
      const double c00 = 0.5*(B2(fk, fj, fi) + B2(fk, fj + 2, fi)), c01 = 0.5*(B2(fk, fj, fi + 1) + B2(fk, fj + 2, fi + 1)),
                   c10 = 0.5*(B2(fk + 1, fj, fi) + B2(fk + 1, fj + 2, fi)), c11 = 0.5*(B2(fk + 1, fj, fi + 1) + B2(fk + 1, fj + 2, fi + 1));
      const double d00 = 0.5*(B3(fk + 2, fj, fi) + B3(fk, fj, fi)), d01 = 0.5*(B3(fk + 2, fj, fi + 1) + B3(fk, fj, fi + 1)),
                   d10 = 0.5*(B3(fk + 2, fj + 1, fi) + B3(fk, fj + 1, fi)), d11 = 0.5*(B3(fk + 2, fj + 1, fi + 1) + B3(fk, fj + 1, fi + 1));
      ST(0, fk, fj, fi + 1, a00 + Uxx - Vxyz - Wxyz);
      ST(0, fk, fj + 1, fi + 1, a01 + Uxx - Vxyz + Wxyz);
      ST(0, fk + 1, fj, fi + 1, a10 + Uxx + Vxyz - Wxyz);
      ST(0, fk + 1, fj + 1, fi + 1, a11 + Uxx + Vxyz + Wxyz);
      ST(1, fk, fj + 1, fi, c00 + Vyy - Uxyz - Wxyz);
      ST(1, fk, fj + 1, fi + 1, c01 + Vyy - Uxyz + Wxyz);
      ST(1, fk + 1, fj + 1, fi, c10 + Vyy + Uxyz - Wxyz);
      ST(1, fk + 1, fj + 1, fi + 1, c11 + Vyy + Uxyz + Wxyz);
      ST(2, fk + 1, fj, fi, d00 + Wzz - Uxyz - Vxyz);
      ST(2, fk + 1, fj, fi + 1, d01 + Wzz - Uxyz + Vxyz);
      ST(2, fk + 1, fj + 1, fi, d10 + Wzz + Uxyz - Vxyz);
      ST(2, fk + 1, fj + 1, fi + 1, d11 + Wzz + Uxyz + Vxyz);
    } else {
      const double tmp1 = 0.25*(B2(fk, fj + 2, fi + 1) - B2(fk, fj, fi + 1) - B2(fk, fj + 2, fi) + B2(fk, fj, fi));
      const double tmp2 = 0.25*(B1(fk, fj, fi) - B1(fk, fj, fi + 2) - B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2));
      const double a0 = 0.5*(B1(fk, fj, fi) + B1(fk, fj, fi + 2)), a1 = 0.5*(B1(fk, fj + 1, fi) + B1(fk, fj + 1, fi + 2));
      const double c0 = 0.5*(B2(fk, fj, fi) + B2(fk, fj + 2, fi)), c1 = 0.5*(B2(fk, fj, fi + 1) + B2(fk, fj + 2, fi + 1));
      ST(0, fk, fj, fi + 1, a0 + tmp1);
      ST(0, fk, fj + 1, fi + 1, a1 + tmp1);
      ST(1, fk, fj + 1, fi, c0 + tmp2);
      ST(1, fk, fj + 1, fi + 1, c1 + tmp2);
    }
  }
}

// ---- PackAndSendFluxCC: restricted fluxes of a finer block into the coarser neighbour's buffer ----
struct Flx3 { double *f[3]; };
__global__ void __launch_bounds__(256)
k_smr_pack_flux_cc(SGeo s, Tab t, int nvar, int fs, Flx3 flx, double *__restrict__ buf) {
  int m, n, v; wg_slot(t, L_COARSER, nvar, m, n, v); (void)v;
  const int dm = NGID(t, m, n);
  if (dm < 0 || !(NLEV(t, m, n) < t.lev[m])) return;
  int dir;
  if (n < 8) dir = 0; else if (n < 16) dir = 1; else if (n >= 24 && n < 32) dir = 2; else return;
  const Bx b = box_of(t.cc, T_SEND, K_FLXC, n, 0);
  const int cnt = bcount(b);
  double *out = buf + seg_w(t, 1, m, n, dm, NDST(t, m, n)) + (size_t)cnt*v;
  const double *f = flx.f[dir];
  auto X = [&](int kk, int jj, int ii) {
    return f[ix5(nvar, s.N3 + (dir == 2 ? fs : 0), s.N2 + (dir == 1 ? fs : 0), s.N1 + (dir == 0 ? fs : 0), m, v, kk, jj, ii)];
  };
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    const int fi = 2*i - s.cis, fj = 2*j - s.cjs, fk = 2*k - s.cks;
    double r;
    if (dir == 0) {
      if (!s.multi_d) r = X(0, 0, fi);
      else if (!s.three_d) r = 0.5*(X(0, fj, fi) + X(0, fj + 1, fi));
      else r = 0.25*(X(fk, fj, fi) + X(fk, fj + 1, fi) + X(fk + 1, fj, fi) + X(fk + 1, fj + 1, fi));
    } else if (dir == 1) {
      if (!s.three_d) r = 0.5*(X(0, fj, fi) + X(0, fj, fi + 1));
      else r = 0.25*(X(fk, fj, fi) + X(fk, fj, fi + 1) + X(fk + 1, fj, fi) + X(fk + 1, fj, fi + 1));
    } else {
      r = 0.25*(X(fk, fj, fi) + X(fk, fj, fi + 1) + X(fk, fj + 1, fi) + X(fk, fj + 1, fi + 1));
    }
    out[e] = r;       // box order (i fastest) == the reference's (j,k)/(i,k)/(i,j) order: one extent is 1
  }
}

__global__ void __launch_bounds__(256)
k_smr_unpack_flux_cc(SGeo s, Tab t, int nvar, int fs, const double *__restrict__ buf, Flx3 flx) {
  int m, n, v; wg_slot(t, L_FINER, nvar, m, n, v); (void)v;
  if (NGID(t, m, n) < 0 || !(NLEV(t, m, n) > t.lev[m])) return;
  int dir;
  if (n < 8) dir = 0; else if (n < 16) dir = 1; else if (n >= 24 && n < 32) dir = 2; else return;
  const Bx b = box_of(t.cc, T_RECV, K_FLXC, n, 0);
  const int cnt = bcount(b);
  const double *in = buf + seg_r(t, 1, m, n) + (size_t)cnt*v;
  double *f = flx.f[dir];
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    int k, j, i;
    bdecode(b, e, k, j, i);
    f[ix5(nvar, s.N3 + (dir == 2 ? fs : 0), s.N2 + (dir == 1 ? fs : 0), s.N1 + (dir == 0 ? fs : 0), m, v, k, j, i)] = in[e];
  }
}

// ---- edge EMFs ------------------------------------------------------------------------------------
struct E3 { double *e[3]; };
__device__ __forceinline__ size_t e4(const SGeo &s, int v, int m, int k, int j, int i) {
  return ix4(s.N3 + (v != 2), s.N2 + (v != 1), s.N1 + (v != 0), m, k, j, i);
}
// which EMF components a slot carries (flux_correct_fc.cpp:78-284): faces the two tangential ones,
// edges the one along the edge, corners none
__device__ __forceinline__ bool slot_has(int n, int v) {
  if (n < 8) return v != 0;
  if (n < 16) return v != 1;
  if (n < 24) return v == 2;
  if (n < 32) return v != 2;
  if (n < 40) return v == 1;
  if (n < 48) return v == 0;
  return false;
}

// PackAndSendFluxFC: same level -> the values themselves, coarser neighbour -> restricted pairs
__global__ void __launch_bounds__(256)
k_smr_pack_flux_fc(SGeo s, Tab t, E3 ef, double *__restrict__ buf) {
  int m, n, v; wg_slot(t, L_VALID, 3, m, n, v); (void)v;
  const int dm = NGID(t, m, n);
  if (dm < 0 || !(NLEV(t, m, n) <= t.lev[m]) || !slot_has(n, v)) return;
  const bool same = NLEV(t, m, n) == t.lev[m];
  const Bx b = box_of(t.fc, T_SEND, same ? K_FLXS : K_FLXC, n, v);
  const int cnt = bcount(b);
  const int dn = NDST(t, m, n);
  double *out = buf + seg_w(t, 3, m, n, dm, dn) + (size_t)ndat_of(t, 1, dn, T_RECV, same ? 3 : 4)*v;
  const double *e = ef.e[v];
  auto E = [&](int kk, int jj, int ii) { return e[e4(s, v, m, kk, jj, ii)]; };
  for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
    int k, j, i;
    bdecode(b, q, k, j, i);
    double r;
    if (same) {
      r = E(k, j, i);
    } else {
      const int fi = 2*i - s.cis, fj = s.multi_d ? 2*j - s.cjs : 0, fk = s.three_d ? 2*k - s.cks : 0;
      // a coarse edge of component v is the average of the two fine edges along direction v; in
      // collapsed directions there is one (flux_correct_fc.cpp:88-101,136-148,...)
      if (v == 0) r = 0.5*(E(fk, fj, fi) + E(fk, fj, fi + 1));
      else if (v == 1) r = s.multi_d ? 0.5*(E(fk, fj, fi) + E(fk, fj + 1, fi)) : E(0, 0, fi);
      else r = s.three_d ? 0.5*(E(fk, fj, fi) + E(fk + 1, fj, fi)) : E(0, fj, fi);
    }
    out[q] = r;
  }
}

// SumBoundaryFluxes: slots of a (block, component) one after the other
__global__ void __launch_bounds__(256)
k_smr_sum_flux_fc(SGeo s, Tab t, int same_level, const double *__restrict__ buf, E3 ef) {
  const int v = blockIdx.x%3, m = blockIdx.x/3;
  const int ml = t.lev[m];
  double *e = ef.e[v];
  __shared__ SlotMeta sm[48];                 // table look-ups of all slots first (see k_smr_unpack_fc)
  if ((int)threadIdx.x < 48) {
    const int n = threadIdx.x;
    SlotMeta x;
    x.q = -1; x.in = 0; x.b = Bx{0, -1, 0, -1, 0, -1};
    if (n < t.nnghbr && NGID(t, m, n) >= 0) {
      const int nl = NLEV(t, m, n);
      if (((same_level && nl == ml) || (!same_level && nl > ml)) && slot_has(n, v)) {
        x.q = 0;
        x.b = box_of(t.fc, T_RECV, same_level ? K_FLXS : K_FLXC, n, v);
        x.in = (long long)seg_r(t, 3, m, n) + (long long)ndat_of(t, 1, n, T_RECV, same_level ? 3 : 4)*v;
      }
    }
    sm[n] = x;
  }
  __syncthreads();
  for (int n = 0; n < t.nnghbr && n < 48; ++n) {
    if (sm[n].q < 0) continue;
    const Bx b = sm[n].b;
    const int cnt = bcount(b);
    const double *in = buf + sm[n].in;
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      int k, j, i;
      bdecode(b, q, k, j, i);
      e[e4(s, v, m, k, j, i)] += in[q];
    }
    __syncthreads();
  }
}

// ZeroFluxesAtBoundaryWithFiner
__global__ void __launch_bounds__(256)
k_smr_zero_flux_fc(SGeo s, Tab t, E3 ef) {
  int m, n, v; wg_slot(t, L_FINER, 3, m, n, v); (void)v;
  if (n >= 48 || NGID(t, m, n) < 0 || !(NLEV(t, m, n) > t.lev[m]) || !slot_has(n, v)) return;
  const Bx b = box_of(t.fc, T_RECV, K_FLXC, n, v);
  const int cnt = bcount(b);
  for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
    int k, j, i;
    bdecode(b, q, k, j, i);
    ef.e[v][e4(s, v, m, k, j, i)] = 0.0;
  }
}

// AverageBoundaryFluxes (flux_correct_fc.cpp:801-1034); nflx[nmb][48] from the host (the counting
// of :470-570 and :676-760 depends on the neighbour table only)
__global__ void __launch_bounds__(256)
k_smr_average_flux_fc(SGeo s, Tab t, const int *__restrict__ nflx, E3 ef) {
  int m, n, v; wg_slot(t, L_AVG, 3, m, n, v); (void)v;
  if (n >= 48 || !slot_has(n, v)) return;
  // only the first sub-block slot of a face / edge carries the averaging (n = 0,4,8,12,16,18,...)
  const bool face = (n == 0 || n == 4 || n == 8 || n == 12 || n == 24 || n == 28);
  const bool edge = (n >= 16 && n < 24 && n%2 == 0) || (n >= 32 && n < 48 && n%2 == 0);
  if (!face && !edge) return;
  Bx b = box_of(t.fc, T_RECV, K_FLXS, n, v);
  if (b.iu < b.il || b.ju < b.jl || b.ku < b.kl) return;
  double *e = ef.e[v];
  if (edge) {
    const double d = (double)nflx[(size_t)m*48 + n];
    const int cnt = bcount(b);
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      int k, j, i;
      bdecode(b, q, k, j, i);
      e[e4(s, v, m, k, j, i)] /= d;
    }
    return;
  }
  const int nl = NLEV(t, m, n), ml = t.lev[m];       // nl = -1 when there is no neighbour
  // direction along which the face box is trimmed (same level) or halved (finer): the extended one,
  // i.e. the tangential direction other than v
  const int fdir = n < 8 ? 0 : (n < 16 ? 1 : 2);
  const int tdir = 3 - fdir - v;
  const bool tact = tdir == 0 ? true : (tdir == 1 ? s.multi_d != 0 : s.three_d != 0);
  int *lo = tdir == 0 ? &b.il : (tdir == 1 ? &b.jl : &b.kl);
  int *hi = tdir == 0 ? &b.iu : (tdir == 1 ? &b.ju : &b.ku);
  if (nl == ml) {
    if (tact) { *lo += 1; *hi -= 1; }
  } else if (nl >= ml) {
    if (!tact) return;
    const int mid = *lo + (*hi - *lo + 1)/2;
    *lo = mid; *hi = mid;
  } else {
    return;
  }
  if (*hi < *lo) return;
  const int cnt = bcount(b);
  for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
    int k, j, i;
    bdecode(b, q, k, j, i);
    e[e4(s, v, m, k, j, i)] *= 0.5;
  }
}

// RecvAndUnpackFluxFC after the messages are in (flux_correct_fc.cpp:374-1034) for one (MeshBlock, component) per workgroup:
// the four steps above -- sum of the same-level contributions, zero where finer neighbours contribute, sum of theirs,
// average -- touch the edges of block m only and read the buffer, so one workgroup can run them back to back with barriers
// in between: one launch instead of four (120 blocks of 16^3: 13 + 5 + 13 + 11 us of mostly launch latency).  Same
// operations on the same operands in the same order per edge as the four kernels.
__global__ void __launch_bounds__(256)
k_smr_emf_finish(SGeo s, Tab t, const int *__restrict__ nflx, const double *__restrict__ buf, E3 ef) {
  const int v = blockIdx.x%3, m = blockIdx.x/3;
  const int ml = t.lev[m];
  double *e = ef.e[v];
  __shared__ SlotMeta same[48], finer[48];    // receive boxes of the slots (q < 0: the slot contributes nothing)
  if ((int)threadIdx.x < 48) {
    const int n = threadIdx.x;
    SlotMeta x, y;
    x.q = -1; x.in = 0; x.b = Bx{0, -1, 0, -1, 0, -1};
    y = x;
    if (n < t.nnghbr && NGID(t, m, n) >= 0 && slot_has(n, v)) {
      const int nl = NLEV(t, m, n);
      if (nl == ml) {
        x.q = 0;
        x.b = box_of(t.fc, T_RECV, K_FLXS, n, v);
        x.in = (long long)seg_r(t, 3, m, n) + (long long)ndat_of(t, 1, n, T_RECV, 3)*v;
      } else if (nl > ml) {
        y.q = 0;
        y.b = box_of(t.fc, T_RECV, K_FLXC, n, v);
        y.in = (long long)seg_r(t, 3, m, n) + (long long)ndat_of(t, 1, n, T_RECV, 4)*v;
      }
    }
    same[n] = x; finer[n] = y;
  }
  __syncthreads();
  auto add = [&](const SlotMeta &sm) {         // SumBoundaryFluxes: the slots one after the other
    const Bx b = sm.b;
    const int cnt = bcount(b);
    const double *in = buf + sm.in;
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      int k, j, i;
      bdecode(b, q, k, j, i);
      e[e4(s, v, m, k, j, i)] += in[q];
    }
  };
  for (int n = 0; n < t.nnghbr && n < 48; ++n) {
    if (same[n].q < 0) continue;               // (uniform over the workgroup)
    add(same[n]);
    __syncthreads();
  }
  if (t.multilevel) {
    for (int n = 0; n < t.nnghbr && n < 48; ++n) {   // ZeroFluxesAtBoundaryWithFiner
      if (finer[n].q < 0) continue;
      const Bx b = finer[n].b;
      const int cnt = bcount(b);
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
        int k, j, i;
        bdecode(b, q, k, j, i);
        e[e4(s, v, m, k, j, i)] = 0.0;
      }
    }
    __syncthreads();
    for (int n = 0; n < t.nnghbr && n < 48; ++n) {
      if (finer[n].q < 0) continue;
      add(finer[n]);
      __syncthreads();
    }
  }
  // AverageBoundaryFluxes (see k_smr_average_flux_fc): the boxes of different faces / edges are disjoint
  for (int n = 0; n < t.nnghbr && n < 48; ++n) {               // (1-D: 8 slots, 2-D: 24)
    if (!slot_has(n, v)) continue;
    const bool face = (n == 0 || n == 4 || n == 8 || n == 12 || n == 24 || n == 28);
    const bool edge = (n >= 16 && n < 24 && n%2 == 0) || (n >= 32 && n < 48 && n%2 == 0);
    if (!face && !edge) continue;
    Bx b = box_of(t.fc, T_RECV, K_FLXS, n, v);
    if (b.iu < b.il || b.ju < b.jl || b.ku < b.kl) continue;
    if (edge) {
      const double d = (double)nflx[(size_t)m*48 + n];
      const int cnt = bcount(b);
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
        int k, j, i;
        bdecode(b, q, k, j, i);
        e[e4(s, v, m, k, j, i)] /= d;
      }
      continue;
    }
    const int nl = NLEV(t, m, n);
    const int fdir = n < 8 ? 0 : (n < 16 ? 1 : 2);
    const int tdir = 3 - fdir - v;
    const bool tact = tdir == 0 ? true : (tdir == 1 ? s.multi_d != 0 : s.three_d != 0);
    int *lo = tdir == 0 ? &b.il : (tdir == 1 ? &b.jl : &b.kl);
    int *hi = tdir == 0 ? &b.iu : (tdir == 1 ? &b.ju : &b.ku);
    if (nl == ml) {
      if (tact) { *lo += 1; *hi -= 1; }
    } else if (nl >= ml) {
      if (!tact) continue;
      const int mid = *lo + (*hi - *lo + 1)/2;
      *lo = mid; *hi = mid;
    } else {
      continue;
    }
    if (*hi < *lo) continue;
    const int cnt = bcount(b);
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      int k, j, i;
      bdecode(b, q, k, j, i);
      e[e4(s, v, m, k, j, i)] *= 0.5;
    }
  }
}

// launch over the (block, slot) pairs of work list L (or the full grid without lists), `per` workgroups per pair
#define SMR_LAUNCH(kern, L, per, stream, ...) \
  do { const unsigned nwg_ = wg_count(tb, L)*(unsigned)(per); if (nwg_) kern<<<nwg_, 256, 0, stream>>>(__VA_ARGS__); } while (0)

static int check_smr(const akmi_pack *p, const akmi_smr *t, const char *who) {
  if (!p || !t || !t->nghbr || !t->mblev || !t->cc_tab || !t->fc_tab || !t->layout || !t->ndat) {
    set_error("%s: incomplete akmi_smr descriptor", who);
    return AKMI_FAIL;
  }
  if (p->nx1 % 2 || (p->nx2 > 1 && p->nx2 % 2) || (p->nx3 > 1 && p->nx3 % 2) || p->ng % 2) {
    set_error("%s: MeshBlock cells and ghost cells must be even with mesh refinement", who);
    return AKMI_FAIL;
  }
  return AKMI_COMPLETE;
}

// ---------------------------------------------------------------------------------------------------------------
// The face-field exchange as ONE list of element copies (akmi_smr_fc_map / akmi_smr_fc_copy).
//
// PackAndSendFC + RecvAndUnpackFC (bvals_fc.cpp:40-560) move values and nothing else: every ghost face of the fine or
// coarse array of a MeshBlock ends up holding the value of ONE face of another array (the RestrictFC'ed coarse array of
// a finer neighbour, the fine array of a same-level or coarser one) -- the one the LAST slot of the reference's
// sequential unpack that covers it delivers, unless the block owns the face itself.  Which element that is depends on
// the mesh only.  So the walk is run ONCE, at set-up, on arrays whose elements hold their own index: k_smr_pack_fc and
// k_smr_unpack_fc, unchanged, leave in every element they write the index of the element the value came from, and the
// pairs (destination, source) that differ from the identity ARE the exchange, in the reference's order of
// precedence.  Per stage one launch copies them (k_smr_fc_copy): no slot walk (k_smr_unpack_fc: 27 dependent rounds
// per (block, component)), no buffer round trip for neighbours in the pack.
// Index space: [b1 | b2 | b3 | cb1 | cb2 | cb3 | buf], 32-bit.  With ranks the message to another rank is the part
// [send_lo, send_hi) of buf: `which` = 1 lists (buffer element <- array element) for it, `which` = 0 lists what the
// unpack does, with sources in the arrays of this pack or in the received part of buf.
struct FcIdx { long long base[8]; };             // starts of the seven arrays in the index space, base[7] = end
struct FcArr { double *a[7]; };
__global__ void k_fc_codes(double *__restrict__ a, long long n, long long first) {
  const long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) a[i] = (double)(first + i + 1);
}
// elements [lo, lo + n) of one array of the index space that do not hold their own code: per workgroup of 256 the
// count (map == null) or the pairs, written in order at off[workgroup].  A pair whose SOURCE is itself overwritten by the
// exchange (the restricted surface faces of a fine block's coarse array: sent to the coarser neighbour and received
// from it) has to read the old value -- the reference packs every message before it unpacks any: such pairs are
// counted in *ndep, replaced by a no-op in the ordered list and appended behind it (akmi_smr_fc_copy runs that tail
// first, in a launch of its own).
// VarSel (cell-centred variables): the copies of the nvar variables of a cell are the same copy nvar times -- keep the
// pairs of variable 0 only; index space [u | cu], nu = elements of u, cs / ccs = cells per variable of a fine / coarse block
struct VarSel { long long nu, cs, ccs; int nvar; };
__device__ __forceinline__ bool var0(const VarSel &f, long long g) {
  return f.nvar == 0 || (g < f.nu ? (g/f.cs)%f.nvar : ((g - f.nu)/f.ccs)%f.nvar) == 0;
}
static_assert(AKMI_WAVE == 64, "k_fc_pairs: 64-lane ballots, four waves per 256-thread workgroup");
__global__ void __launch_bounds__(256)
k_fc_pairs(const double *__restrict__ a, long long n, long long first, const double *__restrict__ tmp, long long ntmp,
           int *__restrict__ cnt, const long long *__restrict__ off, int *__restrict__ map, long long np,
           int *__restrict__ ndep, int *__restrict__ bad, VarSel sel) {
  const long long i = (long long)blockIdx.x*256 + threadIdx.x;
  const bool in = i < n;
  const double v = in ? a[i] : 0.0;
  const bool ch = in && v != (double)(first + i + 1) && var0(sel, first + i);
  __shared__ int wsum[4];
  const unsigned long long bal = __ballot(ch);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  if (!map && threadIdx.x == 0) cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (!ch) return;
  const long long src = (long long)v - 1;
  if (!(v >= 1.0) || v != (double)(src + 1)) { atomicAdd(bad, 1); return; }      // not a code: a value nobody wrote
  if (sel.nvar && (src >= ntmp || !var0(sel, src))) { atomicAdd(bad, 1); return; }   // variable 0 comes from variable 0
  bool dep = false;
  if (src < ntmp && tmp[src] != (double)(src + 1)) {          // the source is a destination of another copy
    dep = true;
    const long long s2 = (long long)tmp[src] - 1;             // ... whose own source must then be an untouched one
    if (s2 >= 0 && s2 < ntmp && tmp[s2] != (double)(s2 + 1)) atomicAdd(bad, 1);
  }
  if (!map) { if (dep) atomicAdd(ndep, 1); return; }
  int r = __popcll(bal & ((1ull << lane) - 1ull));
  for (int q = 0; q < wv; ++q) r += wsum[q];
  const long long o = off[blockIdx.x] + r;
  map[2*o] = (int)(first + i);
  map[2*o + 1] = dep ? (int)(first + i) : (int)src;
  if (dep) {
    const long long t = np + atomicAdd(ndep, 1);
    map[2*t] = (int)(first + i);
    map[2*t + 1] = (int)src;
  }
}
__device__ __forceinline__ double *fc_at(const FcIdx &ix, const FcArr &ar, int g) {
  int q = 0;
#pragma unroll
  for (int t = 1; t < 7; ++t) q += ((long long)g >= ix.base[t]) ? 1 : 0;
  return ar.a[q] + ((long long)g - ix.base[q]);
}
__global__ void __launch_bounds__(256)
k_smr_fc_copy(FcIdx ix, FcArr ar, const int2 *__restrict__ map, long long np) {
  const long long e = (long long)blockIdx.x*256 + threadIdx.x;
  if (e >= np) return;
  const int2 ds = map[e];
  if (ds.x != ds.y) *fc_at(ix, ar, ds.x) = *fc_at(ix, ar, ds.y);
}
// the cell-centred list: pairs of variable 0 in the index space [u | cu]; variable v of a pair lies v*cs (fine array) or
// v*ccs (coarse array) further on.  A thread moves all variables of its pair (the list is read once): loads of up to
// eight variables first, then their stores.
__global__ void __launch_bounds__(256)
k_smr_cc_copy(VarSel f, double *__restrict__ u, double *__restrict__ cu, const int2 *__restrict__ map, long long np) {
  const long long e = (long long)blockIdx.x*256 + threadIdx.x;
  if (e >= np) return;
  const int2 ds = map[e];
  if (ds.x == ds.y) return;
  const long long d = ds.x, s = ds.y;
  const double *src = s < f.nu ? u + s : cu + (s - f.nu);
  double *dst = d < f.nu ? u + d : cu + (d - f.nu);
  const long long ss = s < f.nu ? f.cs : f.ccs, sd = d < f.nu ? f.cs : f.ccs;
  for (int v0 = 0; v0 < f.nvar; v0 += 8) {
    double val[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) if (v0 + q < f.nvar) val[q] = src[(v0 + q)*ss];
#pragma unroll
    for (int q = 0; q < 8; ++q) if (v0 + q < f.nvar) dst[(v0 + q)*sd] = val[q];
  }
}
static VarSel cc_sel(const akmi_pack *p, int nvar) {
  const SGeo s = make_sgeo(p);
  VarSel f;
  f.cs = (long long)s.N3*s.N2*s.N1; f.ccs = (long long)s.cN3*s.cN2*s.cN1;
  f.nu = (long long)p->nmb*nvar*f.cs; f.nvar = nvar;
  return f;
}
static FcIdx fc_index(const akmi_pack *p, long long buf_doubles) {
  const SGeo s = make_sgeo(p);
  FcIdx ix;
  long long o = 0;
  for (int c = 0; c < 2; ++c)
    for (int v = 0; v < 3; ++v) {
      const long long n1 = c ? s.cN1 : s.N1, n2 = c ? s.cN2 : s.N2, n3 = c ? s.cN3 : s.N3;
      ix.base[c*3 + v] = o;
      o += (long long)p->nmb*(n3 + (v == 2))*(n2 + (v == 1))*(n1 + (v == 0));
    }
  ix.base[6] = o;
  ix.base[7] = o + buf_doubles;
  return ix;
}

}  // namespace akmi

using namespace akmi;

extern "C" {

// phase: 1 = PackAndSend*, 2 = RecvAndUnpack*, 3 = both (neighbours in the same pack only)
static int smr_exchange_cc(const akmi_pack *p, const akmi_smr *t, int nvar, double *u, double *cu,
                           double *buf, void *stream, int phase) {
  if (check_smr(p, t, "smr_exchange_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  const SGeo s = make_sgeo(p);
  const Tab tb = make_tab(p, t);
  hipStream_t st = (hipStream_t)stream;
  if (phase & 1) SMR_LAUNCH(k_smr_pack_cc, L_VALID_CC, 1, st, s, tb, nvar, u, cu, buf);
  if (phase & 2) SMR_LAUNCH(k_smr_unpack_cc, L_VALID_CC, 1, st, s, tb, nvar, buf, u, cu);
  AKMI_CHECK_LAUNCH("smr_exchange_cc");
  return AKMI_COMPLETE;
}

static int smr_exchange_fc(const akmi_pack *p, const akmi_smr *t, double *b1, double *b2, double *b3,
                           double *cb1, double *cb2, double *cb3, double *buf, void *stream, int phase) {
  if (check_smr(p, t, "smr_exchange_fc") != AKMI_COMPLETE) return AKMI_FAIL;
  const SGeo s = make_sgeo(p);
  const Tab tb = make_tab(p, t);
  hipStream_t st = (hipStream_t)stream;
  if (phase & 1) SMR_LAUNCH(k_smr_pack_fc, L_VALID, 3, st, s, tb, CF3{{b1, b2, b3}}, CF3{{cb1, cb2, cb3}}, buf);
  if (phase & 2) k_smr_unpack_fc<<<(unsigned)p->nmb*3, 256, 0, st>>>(s, tb, buf, F3{{b1, b2, b3}}, F3{{cb1, cb2, cb3}});
  AKMI_CHECK_LAUNCH("smr_exchange_fc");
  return AKMI_COMPLETE;
}

int akmi_smr_fill_coarse_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, double *cu,
                            void *stream) {
  if (check_smr(p, t, "smr_fill_coarse_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  if (p->nx2 <= 1) return AKMI_COMPLETE;
  const Tab tb = make_tab(p, t);
  SMR_LAUNCH(k_smr_fill_coarse_cc, L_SAME_NEEDS, nvar, (hipStream_t)stream, make_sgeo(p), tb, nvar, u, cu);
  AKMI_CHECK_LAUNCH("smr_fill_coarse_cc");
  return AKMI_COMPLETE;
}

int akmi_smr_fill_coarse_fc(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2,
                            const double *b3, double *cb1, double *cb2, double *cb3, void *stream) {
  if (check_smr(p, t, "smr_fill_coarse_fc") != AKMI_COMPLETE) return AKMI_FAIL;
  if (p->nx2 <= 1) return AKMI_COMPLETE;
  const Tab tb = make_tab(p, t);
  SMR_LAUNCH(k_smr_fill_coarse_fc, L_SAME_NEEDS, 3, (hipStream_t)stream, make_sgeo(p), tb, CF3{{b1, b2, b3}}, F3{{cb1, cb2, cb3}});
  AKMI_CHECK_LAUNCH("smr_fill_coarse_fc");
  return AKMI_COMPLETE;
}

int akmi_smr_prolong_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *cu, double *u,
                        void *stream) {
  if (check_smr(p, t, "smr_prolong_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  const Tab tb = make_tab(p, t);
  SMR_LAUNCH(k_smr_prolong_cc, L_COARSER, 1, (hipStream_t)stream, make_sgeo(p), tb, nvar, cu, u);
  AKMI_CHECK_LAUNCH("smr_prolong_cc");
  return AKMI_COMPLETE;
}

static int check_prims(const akmi_pack *p, int nvar, const char *who) {
  if (!p->is_ideal || nvar < 5) {
    // prolong_prims.cpp converts with SingleC2P_IdealHyd / _IdealMHD whatever the EOS of the run is
    set_error("%s: prolong_primitives needs the ideal-gas EOS (5 fluid variables)", who);
    return AKMI_FAIL;
  }
  return AKMI_COMPLETE;
}

int akmi_smr_c2p_coarse(const akmi_pack *p, const akmi_smr *t, int nvar, double *cu, const double *cb1,
                        const double *cb2, const double *cb3, double *cw, void *stream) {
  if (check_smr(p, t, "smr_c2p_coarse") != AKMI_COMPLETE || check_prims(p, nvar, "smr_c2p_coarse") != AKMI_COMPLETE)
    return AKMI_FAIL;
  const Tab tb = make_tab(p, t);
  SMR_LAUNCH(k_smr_c2p_coarse, L_COARSER, 1, (hipStream_t)stream,
      make_sgeo(p), tb, make_eos(p), nvar, cb1 != nullptr, cu, CF3{{cb1, cb2, cb3}}, cw);
  AKMI_CHECK_LAUNCH("smr_c2p_coarse");
  return AKMI_COMPLETE;
}

int akmi_smr_p2c_fine(const akmi_pack *p, const akmi_smr *t, int nvar, const double *w, const double *b1,
                      const double *b2, const double *b3, double *u, void *stream) {
  if (check_smr(p, t, "smr_p2c_fine") != AKMI_COMPLETE || check_prims(p, nvar, "smr_p2c_fine") != AKMI_COMPLETE)
    return AKMI_FAIL;
  const Tab tb = make_tab(p, t);
  SMR_LAUNCH(k_smr_p2c_fine, L_COARSER, 1, (hipStream_t)stream,
      make_sgeo(p), tb, nvar, b1 != nullptr, w, CF3{{b1, b2, b3}}, u);
  AKMI_CHECK_LAUNCH("smr_p2c_fine");
  return AKMI_COMPLETE;
}

int akmi_smr_prolong_fc(const akmi_pack *p, const akmi_smr *t, const double *cb1, const double *cb2,
                        const double *cb3, double *b1, double *b2, double *b3, void *stream) {
  if (check_smr(p, t, "smr_prolong_fc") != AKMI_COMPLETE) return AKMI_FAIL;
  if (!t->slot_ox) { set_error("smr_prolong_fc: slot offsets missing"); return AKMI_FAIL; }
  const SGeo s = make_sgeo(p);
  const Tab tb = make_tab(p, t);
  hipStream_t st = (hipStream_t)stream;
  SMR_LAUNCH(k_smr_prolong_fc_shared, L_COARSER, 3, st, s, tb, t->slot_ox, CF3{{cb1, cb2, cb3}}, F3{{b1, b2, b3}});
  SMR_LAUNCH(k_smr_prolong_fc_internal, L_COARSER, 1, st, s, tb, t->slot_ox, F3{{b1, b2, b3}});
  AKMI_CHECK_LAUNCH("smr_prolong_fc");
  return AKMI_COMPLETE;
}

static int smr_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, double *flx1,
                       double *flx2, double *flx3, double *buf, void *stream, int phase) {
  if (check_smr(p, t, "smr_flux_cc") != AKMI_COMPLETE) return AKMI_FAIL;
  const SGeo s = make_sgeo(p);
  const Tab tb = make_tab(p, t);
  hipStream_t st = (hipStream_t)stream;
  const int fs = face_shaped ? 1 : 0;
  if (phase & 1) SMR_LAUNCH(k_smr_pack_flux_cc, L_COARSER, nvar, st, s, tb, nvar, fs, Flx3{{flx1, flx2, flx3}}, buf);
  if (phase & 2) SMR_LAUNCH(k_smr_unpack_flux_cc, L_FINER, nvar, st, s, tb, nvar, fs, buf, Flx3{{flx1, flx2, flx3}});
  AKMI_CHECK_LAUNCH("smr_flux_cc");
  return AKMI_COMPLETE;
}

static int smr_emf_exchange(const akmi_pack *p, const akmi_smr *t, const int *nflx, double *e1, double *e2,
                            double *e3, double *buf, void *stream, int phase) {
  if (check_smr(p, t, "smr_emf_exchange") != AKMI_COMPLETE) return AKMI_FAIL;
  const SGeo s = make_sgeo(p);
  const Tab tb = make_tab(p, t);
  hipStream_t st = (hipStream_t)stream;
  const E3 ef{{e1, e2, e3}};
  if ((phase & 2) && !nflx) { set_error("smr_emf_exchange: nflx missing"); return AKMI_FAIL; }
  if (phase & 1) SMR_LAUNCH(k_smr_pack_flux_fc, L_VALID, 3, st, s, tb, ef, buf);
  if (phase & 2) {
    static const bool split4 = getenv("AKMI_SMR_EMF_SPLIT") && atoi(getenv("AKMI_SMR_EMF_SPLIT")) != 0;   // A/B: the four kernels
    if (!split4) {
      k_smr_emf_finish<<<(unsigned)p->nmb*3, 256, 0, st>>>(s, tb, nflx, buf, ef);
    } else {
    k_smr_sum_flux_fc<<<(unsigned)p->nmb*3, 256, 0, st>>>(s, tb, 1, buf, ef);
    if (tb.multilevel) {
      SMR_LAUNCH(k_smr_zero_flux_fc, L_FINER, 3, st, s, tb, ef);
      k_smr_sum_flux_fc<<<(unsigned)p->nmb*3, 256, 0, st>>>(s, tb, 0, buf, ef);
    }
    SMR_LAUNCH(k_smr_average_flux_fc, L_AVG, 3, st, s, tb, nflx, ef);
    }
  }
  AKMI_CHECK_LAUNCH("smr_emf_exchange");
  return AKMI_COMPLETE;
}

// Work lists (include/akmi.h).  A set-up call: the neighbour table comes back to the host once, the lists are built
// there with the tests of the kernels above (every list is a superset of its kernels' tests) and go to `lists`.
int akmi_smr_build_lists(const akmi_pack *p, const akmi_smr *t, int *lists, int *counts, void *stream) {
  if (check_smr(p, t, "smr_build_lists") != AKMI_COMPLETE) return AKMI_FAIL;
  if (!lists || !counts) { set_error("smr_build_lists: lists / counts missing"); return AKMI_FAIL; }
  hipStream_t st = (hipStream_t)stream;
  const int nmb = p->nmb, nn = t->nnghbr;
  std::vector<int> ng((size_t)nmb*56*3), lev(nmb);
  std::vector<unsigned char> needs(nmb, 1);
  if (hipMemcpyAsync(ng.data(), t->nghbr, ng.size()*sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(lev.data(), t->mblev, lev.size()*sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      (t->needs_coarse && hipMemcpyAsync(needs.data(), t->needs_coarse, (size_t)nmb, hipMemcpyDeviceToHost, st) != hipSuccess) ||
      hipStreamSynchronize(st) != hipSuccess) {
    set_error("smr_build_lists: cannot read the tables"); return AKMI_FAIL;
  }
  const size_t stride = (size_t)2*nmb*56;
  std::vector<int> out(stride*AKMI_SMR_NLISTS, 0);
  int cnt[AKMI_SMR_NLISTS] = {0};
  auto push = [&](int L, int m, int n) { out[L*stride + 2*(size_t)cnt[L]] = m; out[L*stride + 2*(size_t)cnt[L] + 1] = n; ++cnt[L]; };
  for (int m = 0; m < nmb; ++m) for (int n = 0; n < nn; ++n) {
    const int gid = ng[((size_t)m*56 + n)*3], nl = ng[((size_t)m*56 + n)*3 + 1], ml = lev[m];
    if (gid >= 0) {
      push(L_VALID, m, n);
      if (!(t->direct_same && nl == ml && gid < nmb)) push(L_VALID_CC, m, n);
      if (nl < ml) push(L_COARSER, m, n);
      if (nl > ml) push(L_FINER, m, n);
      if (nl == ml && needs[m]) push(L_SAME_NEEDS, m, n);
    }
    // AverageBoundaryFluxes: the first sub-block slot of every face and edge, neighbour or not
    const bool face = (n == 0 || n == 4 || n == 8 || n == 12 || n == 24 || n == 28);
    const bool edge = (n >= 16 && n < 24 && n%2 == 0) || (n >= 32 && n < 48 && n%2 == 0);
    if (face || edge) push(L_AVG, m, n);
  }
  if (hipMemcpyAsync(lists, out.data(), out.size()*sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    set_error("smr_build_lists: cannot write the lists"); return AKMI_FAIL;
  }
  for (int l = 0; l < AKMI_SMR_NLISTS; ++l) counts[l] = cnt[l];
  return AKMI_COMPLETE;
}


int akmi_smr_exchange_cc(const akmi_pack *p, const akmi_smr *t, int nvar, double *u, double *cu,
                         double *buf, void *stream) {
  return smr_exchange_cc(p, t, nvar, u, cu, buf, stream, 3);
}
int akmi_smr_pack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, const double *cu,
                     double *buf, void *stream) {
  return smr_exchange_cc(p, t, nvar, const_cast<double *>(u), const_cast<double *>(cu), buf, stream, 1);
}
int akmi_smr_unpack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *buf, double *u,
                       double *cu, void *stream) {
  return smr_exchange_cc(p, t, nvar, u, cu, const_cast<double *>(buf), stream, 2);
}
int akmi_smr_exchange_fc(const akmi_pack *p, const akmi_smr *t, double *b1, double *b2, double *b3,
                         double *cb1, double *cb2, double *cb3, double *buf, void *stream) {
  return smr_exchange_fc(p, t, b1, b2, b3, cb1, cb2, cb3, buf, stream, 3);
}
int akmi_smr_pack_fc(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2,
                     const double *b3, const double *cb1, const double *cb2, const double *cb3, double *buf,
                     void *stream) {
  return smr_exchange_fc(p, t, const_cast<double *>(b1), const_cast<double *>(b2), const_cast<double *>(b3),
                         const_cast<double *>(cb1), const_cast<double *>(cb2), const_cast<double *>(cb3), buf,
                         stream, 1);
}
int akmi_smr_unpack_fc(const akmi_pack *p, const akmi_smr *t, const double *buf, double *b1, double *b2,
                       double *b3, double *cb1, double *cb2, double *cb3, void *stream) {
  return smr_exchange_fc(p, t, b1, b2, b3, cb1, cb2, cb3, const_cast<double *>(buf), stream, 2);
}
int akmi_smr_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, double *flx1,
                     double *flx2, double *flx3, double *buf, void *stream) {
  return smr_flux_cc(p, t, nvar, face_shaped, flx1, flx2, flx3, buf, stream, 3);
}
int akmi_smr_pack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *flx1,
                          const double *flx2, const double *flx3, double *buf, void *stream) {
  return smr_flux_cc(p, t, nvar, face_shaped, const_cast<double *>(flx1), const_cast<double *>(flx2),
                     const_cast<double *>(flx3), buf, stream, 1);
}
int akmi_smr_unpack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *buf,
                            double *flx1, double *flx2, double *flx3, void *stream) {
  return smr_flux_cc(p, t, nvar, face_shaped, flx1, flx2, flx3, const_cast<double *>(buf), stream, 2);
}
int akmi_smr_emf_exchange(const akmi_pack *p, const akmi_smr *t, const int *nflx, double *e1, double *e2,
                          double *e3, double *buf, void *stream) {
  return smr_emf_exchange(p, t, nflx, e1, e2, e3, buf, stream, 3);
}
int akmi_smr_pack_emf(const akmi_pack *p, const akmi_smr *t, const double *e1, const double *e2,
                      const double *e3, double *buf, void *stream) {
  return smr_emf_exchange(p, t, nullptr, const_cast<double *>(e1), const_cast<double *>(e2),
                          const_cast<double *>(e3), buf, stream, 1);
}
int akmi_smr_unpack_emf(const akmi_pack *p, const akmi_smr *t, const int *nflx, const double *buf, double *e1,
                        double *e2, double *e3, void *stream) {
  return smr_emf_exchange(p, t, nflx, e1, e2, e3, const_cast<double *>(buf), stream, 2);
}

long long akmi_smr_fc_map(const akmi_pack *p, const akmi_smr *t, double *buf, long long buf_doubles, long long send_lo,
                          long long send_hi, int which, int *map, long long cap, long long *ntail, void *stream) {
  using namespace akmi;
  if (check_smr(p, t, "smr_fc_map") != AKMI_COMPLETE) return -1;
  hipStream_t st = (hipStream_t)stream;
  const FcIdx ix = fc_index(p, buf_doubles);
  if (ix.base[7] >= (1ll << 31)) { set_error("smr_fc_map: more than 2^31 face elements in a pack"); return -1; }
  if (which && !(0 <= send_lo && send_lo <= send_hi && send_hi <= buf_doubles)) {
    set_error("smr_fc_map: send range outside the buffer"); return -1;
  }
  const long long ntmp = ix.base[6];
  double *tmp = nullptr;
  int *d_cnt = nullptr, *d_flag = nullptr;          // d_flag[0]: pairs that read an overwritten source, [1]: errors
  long long *d_off = nullptr;
  long long result = -1;
  // the region that is scanned: the six arrays (which = 0) or the outgoing part of the buffer (which = 1)
  const double *scan = nullptr;
  const long long first = which ? ntmp + send_lo : 0, n = which ? send_hi - send_lo : ntmp;
  const unsigned nwg = (unsigned)((n + 255)/256);
  std::vector<int> h_cnt(nwg);
  std::vector<long long> h_off(nwg);
  long long np = 0;
  int h_flag[2] = {0, 0};
#define FCM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("smr_fc_map: %s", hipGetErrorString(e_)); goto done; } } while (0)
  FCM_HIP(hipMalloc(&tmp, sizeof(double)*(size_t)(ntmp > 0 ? ntmp : 1)));
  FCM_HIP(hipMalloc(&d_cnt, sizeof(int)*(size_t)(nwg + 1)));
  FCM_HIP(hipMalloc(&d_off, sizeof(long long)*(size_t)(nwg + 1)));
  FCM_HIP(hipMalloc(&d_flag, 2*sizeof(int)));
  FCM_HIP(hipMemsetAsync(d_flag, 0, 2*sizeof(int), st));
  k_fc_codes<<<(unsigned)((ntmp + 255)/256), 256, 0, st>>>(tmp, ntmp, 0);
  if (buf_doubles > 0) k_fc_codes<<<(unsigned)((buf_doubles + 255)/256), 256, 0, st>>>(buf, buf_doubles, ntmp);
  if (smr_exchange_fc(p, t, tmp + ix.base[0], tmp + ix.base[1], tmp + ix.base[2], tmp + ix.base[3], tmp + ix.base[4],
                      tmp + ix.base[5], buf, stream, which ? 1 : 3) != AKMI_COMPLETE) goto done;
  scan = which ? buf + send_lo : tmp;
  if (n > 0) {
    k_fc_pairs<<<nwg, 256, 0, st>>>(scan, n, first, tmp, ntmp, d_cnt, nullptr, nullptr, 0, d_flag, d_flag + 1, VarSel{0, 1, 1, 0});
    FCM_HIP(hipMemcpyAsync(h_cnt.data(), d_cnt, sizeof(int)*nwg, hipMemcpyDeviceToHost, st));
    FCM_HIP(hipMemcpyAsync(h_flag, d_flag, 2*sizeof(int), hipMemcpyDeviceToHost, st));
    FCM_HIP(hipStreamSynchronize(st));
    for (unsigned w = 0; w < nwg; ++w) { h_off[w] = np; np += h_cnt[w]; }
  }
  if (h_flag[1]) { set_error("smr_fc_map: %d copies are not independent element copies", h_flag[1]); goto done; }
  if (map && np > 0) {
    if (np + h_flag[0] > cap) { set_error("smr_fc_map: %lld pairs, room for %lld", np + h_flag[0], cap); goto done; }
    FCM_HIP(hipMemcpyAsync(d_off, h_off.data(), sizeof(long long)*nwg, hipMemcpyHostToDevice, st));
    FCM_HIP(hipMemsetAsync(d_flag, 0, 2*sizeof(int), st));
    k_fc_pairs<<<nwg, 256, 0, st>>>(scan, n, first, tmp, ntmp, d_cnt, d_off, map, np, d_flag, d_flag + 1, VarSel{0, 1, 1, 0});
  }
  if (buf_doubles > 0) FCM_HIP(hipMemsetAsync(buf, 0, sizeof(double)*(size_t)buf_doubles, st));
  FCM_HIP(hipStreamSynchronize(st));
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_error("smr_fc_map: %s", hipGetErrorString(e_)); goto done; } }
  if (ntail) *ntail = h_flag[0];
  result = np + h_flag[0];
done:
#undef FCM_HIP
  (void)hipFree(tmp); (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_flag);
  return result;
}

int akmi_smr_fc_copy(const akmi_pack *p, const int *map, long long npairs, long long ntail, long long buf_doubles,
                     double *b1, double *b2, double *b3, double *cb1, double *cb2, double *cb3, double *buf,
                     void *stream) {
  using namespace akmi;
  if (npairs <= 0) return AKMI_COMPLETE;
  const FcIdx ix = fc_index(p, buf_doubles);
  const FcArr ar{{b1, b2, b3, cb1, cb2, cb3, buf}};
  const int2 *m2 = reinterpret_cast<const int2 *>(map);
  const long long head = npairs - ntail;
  if (ntail > 0)         // the copies that read a face another copy overwrites: first
    k_smr_fc_copy<<<(unsigned)((ntail + 255)/256), 256, 0, (hipStream_t)stream>>>(ix, ar, m2 + head, ntail);
  if (head > 0)
    k_smr_fc_copy<<<(unsigned)((head + 255)/256), 256, 0, (hipStream_t)stream>>>(ix, ar, m2, head);
  AKMI_CHECK_LAUNCH("smr_fc_copy");
  return AKMI_COMPLETE;
}

// The cell-centred list keeps the pairs of variable 0 and akmi_smr_cc_copy applies them to every variable: check, once
// at set-up, that the exchange really moved variable v > 0 exactly as variable 0 (same destination cell, same source
// cell, variable v of the source's own register).
static __global__ void __launch_bounds__(256)
k_cc_var_check(const double *__restrict__ tmp, long long ntmp, akmi::VarSel f, int *__restrict__ bad) {
  const long long g = (long long)blockIdx.x*256 + threadIdx.x;
  if (g >= ntmp) return;
  const bool in_u = g < f.nu;
  const long long st = in_u ? f.cs : f.ccs;
  const int v = (int)((in_u ? g/f.cs : (g - f.nu)/f.ccs)%f.nvar);
  if (v == 0) return;
  const long long g0 = g - (long long)v*st;
  const double c0 = tmp[g0], c = tmp[g];
  if (c0 == (double)(g0 + 1)) { if (c != (double)(g + 1)) atomicAdd(bad, 1); return; }     // untouched together
  const long long s0 = (long long)c0 - 1;
  const long long want = s0 + (long long)v*(s0 < f.nu ? f.cs : f.ccs);
  if (c != (double)(want + 1)) atomicAdd(bad, 1);
}

long long akmi_smr_cc_map(const akmi_pack *p, const akmi_smr *t, int nvar, const int *same27, double *buf,
                          long long buf_doubles, int *map, long long cap, long long *ntail, void *stream) {
  using namespace akmi;
  if (check_smr(p, t, "smr_cc_map") != AKMI_COMPLETE) return -1;
  if (t->soff || t->roff) { set_error("smr_cc_map: one rank only (neighbours on other ranks go through the buffers)"); return -1; }
  if (nvar < 1) { set_error("smr_cc_map: nvar"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const VarSel sel = cc_sel(p, nvar);
  const long long ncu = (long long)p->nmb*nvar*sel.ccs, ntmp = sel.nu + ncu;
  if (ntmp + buf_doubles >= (1ll << 31)) { set_error("smr_cc_map: more than 2^31 elements in a pack"); return -1; }
  double *tmp = nullptr;
  int *d_cnt = nullptr, *d_flag = nullptr;
  long long *d_off = nullptr;
  long long result = -1, np = 0;
  const unsigned nwg = (unsigned)((ntmp + 255)/256);
  std::vector<int> h_cnt(nwg);
  std::vector<long long> h_off(nwg);
  int h_flag[2] = {0, 0};
#define CCM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("smr_cc_map: %s", hipGetErrorString(e_)); goto done; } } while (0)
  CCM_HIP(hipMalloc(&tmp, sizeof(double)*(size_t)ntmp));
  CCM_HIP(hipMalloc(&d_cnt, sizeof(int)*(size_t)(nwg + 1)));
  CCM_HIP(hipMalloc(&d_off, sizeof(long long)*(size_t)(nwg + 1)));
  CCM_HIP(hipMalloc(&d_flag, 2*sizeof(int)));
  CCM_HIP(hipMemsetAsync(d_flag, 0, 2*sizeof(int), st));
  k_fc_codes<<<(unsigned)((ntmp + 255)/256), 256, 0, st>>>(tmp, ntmp, 0);
  if (buf_doubles > 0) k_fc_codes<<<(unsigned)((buf_doubles + 255)/256), 256, 0, st>>>(buf, buf_doubles, ntmp);
  // the exchange as the hosts run it: pack + unpack across levels, then the direct same-level gather
  if (smr_exchange_cc(p, t, nvar, tmp, tmp + sel.nu, buf, stream, 3) != AKMI_COMPLETE) goto done;
  if (t->direct_same) {
    if (!same27) { set_error("smr_cc_map: direct_same without the same-level table"); goto done; }
    if (akmi_bvals_cc_local(p, nvar, same27, tmp, stream) != AKMI_COMPLETE) goto done;
  }
  k_fc_pairs<<<nwg, 256, 0, st>>>(tmp, ntmp, 0, tmp, ntmp, d_cnt, nullptr, nullptr, 0, d_flag, d_flag + 1, sel);
  if (nvar > 1) k_cc_var_check<<<nwg, 256, 0, st>>>(tmp, ntmp, sel, d_flag + 1);
  CCM_HIP(hipMemcpyAsync(h_cnt.data(), d_cnt, sizeof(int)*nwg, hipMemcpyDeviceToHost, st));
  CCM_HIP(hipMemcpyAsync(h_flag, d_flag, 2*sizeof(int), hipMemcpyDeviceToHost, st));
  CCM_HIP(hipStreamSynchronize(st));
  for (unsigned w = 0; w < nwg; ++w) { h_off[w] = np; np += h_cnt[w]; }
  if (h_flag[1]) { set_error("smr_cc_map: %d copies are not independent element copies of all variables alike", h_flag[1]); goto done; }
  if (map && np > 0) {
    if (np + h_flag[0] > cap) { set_error("smr_cc_map: %lld pairs, room for %lld", np + h_flag[0], cap); goto done; }
    CCM_HIP(hipMemcpyAsync(d_off, h_off.data(), sizeof(long long)*nwg, hipMemcpyHostToDevice, st));
    CCM_HIP(hipMemsetAsync(d_flag, 0, 2*sizeof(int), st));
    k_fc_pairs<<<nwg, 256, 0, st>>>(tmp, ntmp, 0, tmp, ntmp, d_cnt, d_off, map, np, d_flag, d_flag + 1, sel);
  }
  if (buf_doubles > 0) CCM_HIP(hipMemsetAsync(buf, 0, sizeof(double)*(size_t)buf_doubles, st));
  CCM_HIP(hipStreamSynchronize(st));
  { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_error("smr_cc_map: %s", hipGetErrorString(e_)); goto done; } }
  if (ntail) *ntail = h_flag[0];
  result = np + h_flag[0];
done:
#undef CCM_HIP
  (void)hipFree(tmp); (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_flag);
  return result;
}

int akmi_smr_cc_copy(const akmi_pack *p, int nvar, const int *map, long long npairs, long long ntail, double *u,
                     double *cu, void *stream) {
  using namespace akmi;
  if (npairs <= 0) return AKMI_COMPLETE;
  const VarSel sel = cc_sel(p, nvar);
  const int2 *m2 = reinterpret_cast<const int2 *>(map);
  const long long head = npairs - ntail;
  if (ntail > 0)
    k_smr_cc_copy<<<(unsigned)((ntail + 255)/256), 256, 0, (hipStream_t)stream>>>(sel, u, cu, m2 + head, ntail);
  if (head > 0)
    k_smr_cc_copy<<<(unsigned)((head + 255)/256), 256, 0, (hipStream_t)stream>>>(sel, u, cu, m2, head);
  AKMI_CHECK_LAUNCH("smr_cc_copy");
  return AKMI_COMPLETE;
}

}  // extern "C"
