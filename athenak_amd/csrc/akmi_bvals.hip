// akmi_bvals.hip -- same-level ghost-zone fill, pack/unpack for off-rank neighbours, and
// physical boundary conditions.
//
// The reference packs the ng innermost active layers of every block into per-neighbour
// buffers (src/bvals/buffs_cc.cpp:36-46, buffs_fc.cpp:39-76), copies them straight into the
// neighbour's receive buffer when it lives on the same rank (src/bvals/bvals_cc.cpp:122-135)
// and unpacks into the ghost layers (buffs_cc.cpp:176-203, buffs_fc.cpp:396-431).  Here the
// same-rank path is a single gather kernel over the ghost shell (no staging buffer at all):
// ghost element (k,j,i) of block m in direction o := element (k-o3*nx3, j-o2*nx2, i-o1*nx1)
// of nghbr[m][o].  Off-rank segments use the same index maps on both sides, so a segment
// packed by the sender for direction d is exactly the receiver's ghost region for -d.
#include "akmi_common.hpp"
#include <cfloat>

namespace akmi {

// one dimension of a (cell- or face-centred) array: owned range [s,eo], ng ghosts each side
struct Dim {
  int s, eo, ng, nx;
  __host__ __device__ int lo(int o) const { return o < 0 ? s - ng : (o > 0 ? eo + 1 : s); }
  __host__ __device__ int hi(int o) const { return o < 0 ? s - 1 : (o > 0 ? eo + ng : eo); }
  __host__ __device__ int cnt(int o) const { return hi(o) - lo(o) + 1; }
  __host__ __device__ int total() const { return eo + ng + 1; }
  // classify index -> o, (-2 if outside)
  __host__ __device__ int side(int idx) const { return idx < s ? -1 : (idx > eo ? 1 : 0); }
};

// component c: 0 = cell-centred, 1/2/3 = x1f/x2f/x3f
struct Comp {
  Dim d1, d2, d3;
  int n3, n2, n1;
};

__host__ __device__ inline Comp make_comp(const Geo &g, int c) {
  Comp q;
  q.d1 = Dim{g.is, g.ie + (c == 1), g.ng, g.nx1};
  q.d2 = Dim{g.js, g.je + (c == 2), g.multi_d ? g.ng : 0, g.multi_d ? g.nx2 : 0};
  q.d3 = Dim{g.ks, g.ke + (c == 3), g.three_d ? g.ng : 0, g.three_d ? g.nx3 : 0};
  q.n1 = g.N1 + (c == 1); q.n2 = g.N2 + (c == 2); q.n3 = g.N3 + (c == 3);
  return q;
}

__host__ __device__ inline bool dir_valid(const Geo &g, int d, int &o1, int &o2, int &o3) {
  o1 = d%3 - 1; o2 = (d/3)%3 - 1; o3 = d/9 - 1;
  if (d == 13) return false;
  if (!g.multi_d && o2 != 0) return false;
  if (!g.three_d && o3 != 0) return false;
  return true;
}

__host__ __device__ inline long long seg_count(const Comp &q, int o1, int o2, int o3) {
  return (long long)q.d1.cnt(o1)*q.d2.cnt(o2)*q.d3.cnt(o3);
}

// ---- ghost fill: thread per element of the ghost shell -----------------------------------
// ONE launch covers the whole shell of every component: blockIdx.y = component of the set,
// blockIdx.z = slab of the shell (0: k-ghost slabs incl. all j,i; 1: j-ghost slabs of the owned
// k range; 2: i-ghost slabs of the owned k,j range), blockIdx.x grid-strides over the slab.
//   KIND 0  gather from same-rank neighbours (nghbr >= 0)
//   KIND 1  cell-centred unpack of off-rank segments (nghbr <= -2)
//   KIND 2  face-centred unpack (segment = x1f, x2f, x3f parts back to back)
struct GhostSet {
  Comp q[3];
  double *a[3];
  int comp[3];       // 0 cell-centred, 1/2/3 x1f/x2f/x3f
  int ncomp;
};

// Index arithmetic: the block index carries (MeshBlock, variable, chunk of the slab) -- decoded once per thread from
// wave-uniform values -- and the element inside the slab is a 32-bit number (a slab of one variable of one
// MeshBlock is far below 2^31 elements; checked at launch).  The first version decoded a 64-bit flat index over
// (block, variable, element) with four 64-bit divisions per element behind a 65535-workgroup grid-stride loop;
// measured, that arithmetic was NOT what bounds the fill of many small blocks (960 blocks of 32^3, ng = 4:
// 919 -> 895 us for 2.4 GB): the x1 slabs are rows of 2 ng doubles, 64-byte pieces of 128-byte lines on both sides.
// BC = true (KIND 0 only): the physical boundary conditions of the pack folded into the gather.  The reference fills the
// ghost zones from the neighbours and then applies the boundary functions direction by direction (x1, x2, x3), each over
// ALL transverse indices (bvals/physics/hydro_bcs.cpp:69-230, bfield_bcs.cpp:66-300), so the value that ends up in a ghost
// element is  T3(T2(T1(fill value at c')))  where c' is the element with every physically bounded ghost coordinate mapped
// to its source coordinate (x3 first, then x2, then x1: the inverse order) and T_d the value rule of direction d (identity
// for outflow, the sign of the normal component for reflect, the clamp for diode, constants for inflow / vacuum).  One
// launch instead of the gather plus one kernel per bounded direction; every source element is an owned element of some
// MeshBlock (or a ghost element nothing fills), never one this launch writes.
struct GhostBC {
  const int *bcs;        // [nmb][6] AKMI_BC_*
  const double *in;      // inflow constants: u_in[nvar][6] (cell-centred set) / b_in[3][6] (face-centred set), may be null
  double *dt3;           // when non-null: the three CFL minima are reset here (saves the k_init_dt3 launch of the last stage)
};
__host__ __device__ inline bool bc_is_physical(int f) {
  return f == AKMI_BC_REFLECT || f == AKMI_BC_OUTFLOW || f == AKMI_BC_INFLOW || f == AKMI_BC_DIODE || f == AKMI_BC_VACUUM;
}
// source coordinate of ghost coordinate x on side o (-1 / +1) of a dimension under boundary type f; fcd: the array is
// face-centred along this dimension (owned faces s..eo = e+1)
__device__ __forceinline__ int bc_source(const Dim &dm, int x, int o, int f, bool fcd) {
  if (f == AKMI_BC_REFLECT) return o < 0 ? 2*dm.s - (fcd ? 0 : 1) - x : 2*dm.eo + (fcd ? 0 : 1) - x;
  return o < 0 ? dm.s : dm.eo;
}
// value rule of direction D (face = 2 D + side) for variable n of a cell-centred set (comp 0) or for face component comp
__device__ __forceinline__ double bc_value(double v, int f, int D, int side, int comp, int n, const double *in) {
  const bool normal = comp == 0 ? (n == 1 + D) : (comp == 1 + D);
  if (f == AKMI_BC_REFLECT) {
    if (comp == 0) { const double sgn = normal ? -1.0 : 1.0; return sgn*v; }
    return normal ? -1.0*v : v;
  }
  if (f == AKMI_BC_INFLOW) return in ? in[6*(comp == 0 ? n : comp - 1) + 2*D + side] : 0.0;   // no table (akmi.h allows NULL): zeros
  if (comp == 0) {
    if (f == AKMI_BC_DIODE) return normal ? (side ? fmax(0.0, v) : fmin(0.0, v)) : v;
    if (f == AKMI_BC_VACUUM) return 0.0;
  }
  return v;              // outflow; the field under diode / vacuum (bfield_bcs.cpp:88-97)
}

template <int KIND, bool BC = false>
__global__ void __launch_bounds__(256)
k_ghost_fill(Geo g, GhostSet gs, int nv, unsigned chunks, const int *__restrict__ nghbr,
             const long long *__restrict__ seg_off, const double *__restrict__ recvbuf, GhostBC bc) {
  const Comp q = gs.q[blockIdx.y];
  double *__restrict__ a = gs.a[blockIdx.y];
  const int comp = gs.comp[blockIdx.y];
  const int mode = blockIdx.z;
  unsigned e1, e2, e3;
  if (mode == 0) { e1 = q.n1; e2 = q.n2; e3 = 2*q.d3.ng; }
  else if (mode == 1) { e1 = q.n1; e2 = 2*q.d2.ng; e3 = q.d3.eo - q.d3.s + 1; }
  else { e1 = 2*q.d1.ng; e2 = q.d2.eo - q.d2.s + 1; e3 = q.d3.eo - q.d3.s + 1; }
  const unsigned per = e1*e2*e3, e12 = e1*e2;
  const unsigned mn = blockIdx.x/chunks, ch = blockIdx.x - mn*chunks;
  const int m = (int)(mn/(unsigned)nv), n = (int)(mn - (unsigned)m*(unsigned)nv);
  const size_t vbase = ((size_t)m*nv + n)*q.n3;
  // the 27 neighbour entries of this MeshBlock once per workgroup (LDS), not one dependent global load per element;
  // GU elements per thread and pass, all loads of a pass before its stores: the kernel has next to no arithmetic,
  // its rate is the number of bytes in flight (960 blocks of 32^3, ng = 4: 895 -> 748 us, profiles/r03_config5.txt)
  __shared__ int s_src[27];
  __shared__ int s_bc[6];
  if (threadIdx.x < 27) s_src[threadIdx.x] = nghbr[m*27 + threadIdx.x];
  if constexpr (BC) {
    if (threadIdx.x >= 32 && threadIdx.x < 38) s_bc[threadIdx.x - 32] = bc.bcs[6*m + threadIdx.x - 32];
    if (bc.dt3 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x >= 64 && threadIdx.x < 67)
      bc.dt3[threadIdx.x - 64] = (double)FLT_MAX;
  }
  __syncthreads();
  constexpr int GU = 4;
  const unsigned stride = chunks*256u;
  for (unsigned r0 = ch*256u + threadIdx.x; r0 < per; r0 += stride*GU) {
    double v[GU];
    size_t dst[GU];
    bool ok[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const unsigned r = r0 + (unsigned)u*stride;
      ok[u] = r < per;
      v[u] = 0.0; dst[u] = 0;
      if (!ok[u]) continue;
      const unsigned kk = r/e12, r2 = r - kk*e12;
      const unsigned jj = r2/e1;
      const int ii = (int)(r2 - jj*e1);
      int i, j, k;
      if (mode == 0) { i = ii; j = (int)jj; k = (int)kk < q.d3.ng ? (int)kk : q.d3.eo + 1 + ((int)kk - q.d3.ng); }
      else if (mode == 1) { i = ii; j = (int)jj < q.d2.ng ? (int)jj : q.d2.eo + 1 + ((int)jj - q.d2.ng); k = q.d3.s + (int)kk; }
      else { i = ii < q.d1.ng ? ii : q.d1.eo + 1 + (ii - q.d1.ng); j = q.d2.s + (int)jj; k = q.d3.s + (int)kk; }
      int o1 = q.d1.side(i), o2 = q.d2.side(j), o3 = q.d3.side(k);
      dst[u] = ((vbase + k)*q.n2 + j)*q.n1 + i;
      if constexpr (BC) {
        int f1 = -1, f2 = -1, f3 = -1, sd1 = 0, sd2 = 0, sd3 = 0;     // boundary type applied per direction (-1: none)
        int kk2 = k, jj2 = j, ii2 = i;
        if (o3 != 0) { sd3 = o3 > 0; const int f = s_bc[4 + sd3]; if (bc_is_physical(f)) { f3 = f; kk2 = bc_source(q.d3, k, o3, f, comp == 3); o3 = 0; } }
        if (o2 != 0) { sd2 = o2 > 0; const int f = s_bc[2 + sd2]; if (bc_is_physical(f)) { f2 = f; jj2 = bc_source(q.d2, j, o2, f, comp == 2); o2 = 0; } }
        if (o1 != 0) { sd1 = o1 > 0; const int f = s_bc[sd1]; if (bc_is_physical(f)) { f1 = f; ii2 = bc_source(q.d1, i, o1, f, comp == 1); o1 = 0; } }
        const bool mapped = (f1 >= 0) || (f2 >= 0) || (f3 >= 0);
        const int d = (o3 + 1)*9 + (o2 + 1)*3 + (o1 + 1);
        int src = (d == 13) ? m : s_src[d];
        int oo1 = o1, oo2 = o2, oo3 = o3;
        if (src < 0) {
          if (!mapped) { ok[u] = false; continue; }
          src = m; oo1 = oo2 = oo3 = 0;           // nothing fills c': the boundary functions copy what it holds
        }
        double val = a[((((size_t)src*nv + n)*q.n3 + (kk2 - oo3*q.d3.nx))*q.n2 + (jj2 - oo2*q.d2.nx))*q.n1 + (ii2 - oo1*q.d1.nx)];
        if (f1 >= 0) val = bc_value(val, f1, 0, sd1, comp, n, bc.in);
        if (f2 >= 0) val = bc_value(val, f2, 1, sd2, comp, n, bc.in);
        if (f3 >= 0) val = bc_value(val, f3, 2, sd3, comp, n, bc.in);
        v[u] = val;
        continue;
      }
      const int d = (o3 + 1)*9 + (o2 + 1)*3 + (o1 + 1);
      const int src = s_src[d];
      if constexpr (KIND == 0) {
        if (src < 0) { ok[u] = false; continue; }
        v[u] = a[((((size_t)src*nv + n)*q.n3 + (k - o3*q.d3.nx))*q.n2 + (j - o2*q.d2.nx))*q.n1 + (i - o1*q.d1.nx)];
      } else {
        if (src > -2) { ok[u] = false; continue; }
        const int c1 = q.d1.cnt(o1), c2 = q.d2.cnt(o2), c3 = q.d3.cnt(o3);
        long long base = seg_off[-(src + 2)];
        if constexpr (KIND == 2)
          for (int c = 1; c < comp; ++c) base += seg_count(make_comp(g, c), o1, o2, o3);
        const long long off = base + (((long long)n*c3 + (k - q.d3.lo(o3)))*c2 + (j - q.d2.lo(o2)))*c1 +
                              (i - q.d1.lo(o1));
        v[u] = recvbuf[off];
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (ok[u]) a[dst[u]] = v[u];
  }
}

template <int KIND, bool BC = false>
static int launch_ghost(const Geo &g, const GhostSet &gs, int nv, const int *nghbr,
                        const long long *seg_off, const double *recvbuf, hipStream_t st, GhostBC bc = GhostBC{}) {
  long long nmax = 1;
  for (int c = 0; c < gs.ncomp; ++c) {
    const Comp &q = gs.q[c];
    long long n0 = (long long)q.n1*q.n2*2*q.d3.ng;
    long long n1 = (long long)q.n1*2*q.d2.ng*(q.d3.eo - q.d3.s + 1);
    long long n2 = (long long)2*q.d1.ng*(q.d2.eo - q.d2.s + 1)*(q.d3.eo - q.d3.s + 1);
    long long n = (n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2));
    if (n > nmax) nmax = n;
  }
  if (nmax >= (1ll << 31)) { set_error("bvals ghost fill: a ghost slab of one variable has 2^31 elements or more"); return AKMI_FAIL; }
  // chunks of 256 elements per (MeshBlock, variable) slab; few enough of them that the whole grid stays below
  // 2^31 - 1 workgroups and above a few thousand (a thread then strides over its slab)
  const long long mnv = (long long)g.nmb*nv;
  long long chunks = (nmax + 256*4 - 1)/(256*4);          // GU = 4 elements per thread and pass (k_ghost_fill)
  const long long cap = ((1ll << 31) - 1)/mnv;
  if (chunks > cap) chunks = cap;
  if (chunks > 64 && mnv*chunks > (1ll << 20)) { chunks = (1ll << 20)/mnv; if (chunks < 64) chunks = 64; if (chunks > cap) chunks = cap; }
  if (chunks < 1) { set_error("bvals ghost fill: too many MeshBlocks x variables for one launch"); return AKMI_FAIL; }
  dim3 grid((unsigned)(mnv*chunks), gs.ncomp, 3);
  k_ghost_fill<KIND, BC><<<grid, 256, 0, st>>>(g, gs, nv, (unsigned)chunks, nghbr, seg_off, recvbuf, bc);
  AKMI_CHECK_LAUNCH("bvals ghost fill");
  return AKMI_COMPLETE;
}

// ---- hydro, after akmi_hydro_stage_w: ghost zones of u AND w, one thread per ghost CELL -------------------------------
// (akmi_hydro_ghost_uw.)  The thread finds the cell's one source cell with the index maps of k_ghost_fill<0, true>, copies
// its five conserved and five primitive variables under the value rules of the boundary (copy / reflect only: the caller
// has refused everything else) and counts the floor flags of the source.  The generic kernel run over both arrays does the same
// with one thread per (variable, element) and the index arithmetic ten times over: 69 against 40 us at 256^3, 21 against 13 at 128^3.
__global__ void __launch_bounds__(256)
k_ghost_fill_uw(Geo g, Comp q, unsigned chunks, const int *__restrict__ nghbr, const int *__restrict__ bcs,
                double *__restrict__ u, double *__restrict__ w, const unsigned char *__restrict__ flags,
                int *__restrict__ counters) {
  const int mode = blockIdx.z;
  unsigned e1, e2, e3;
  if (mode == 0) { e1 = q.n1; e2 = q.n2; e3 = 2*q.d3.ng; }
  else if (mode == 1) { e1 = q.n1; e2 = 2*q.d2.ng; e3 = q.d3.eo - q.d3.s + 1; }
  else { e1 = 2*q.d1.ng; e2 = q.d2.eo - q.d2.s + 1; e3 = q.d3.eo - q.d3.s + 1; }
  const unsigned per = e1*e2*e3, e12 = e1*e2;
  const unsigned m = blockIdx.x/chunks, ch = blockIdx.x - m*chunks;
  __shared__ int s_src[27];
  __shared__ int s_bc[6];
  if (threadIdx.x < 27) s_src[threadIdx.x] = nghbr[m*27 + threadIdx.x];
  if (threadIdx.x >= 32 && threadIdx.x < 38) s_bc[threadIdx.x - 32] = bcs[6*m + threadIdx.x - 32];
  __syncthreads();
  const size_t cs = (size_t)q.n3*q.n2*q.n1;
  for (unsigned r = ch*256u + threadIdx.x; r < per; r += chunks*256u) {
    const unsigned kk = r/e12, r2 = r - kk*e12;
    const unsigned jj = r2/e1;
    const int ii = (int)(r2 - jj*e1);
    int i, j, k;
    if (mode == 0) { i = ii; j = (int)jj; k = (int)kk < q.d3.ng ? (int)kk : q.d3.eo + 1 + ((int)kk - q.d3.ng); }
    else if (mode == 1) { i = ii; j = (int)jj < q.d2.ng ? (int)jj : q.d2.eo + 1 + ((int)jj - q.d2.ng); k = q.d3.s + (int)kk; }
    else { i = ii < q.d1.ng ? ii : q.d1.eo + 1 + (ii - q.d1.ng); j = q.d2.s + (int)jj; k = q.d3.s + (int)kk; }
    int o1 = q.d1.side(i), o2 = q.d2.side(j), o3 = q.d3.side(k);
    int f1 = -1, f2 = -1, f3 = -1;                                   // boundary type applied per direction (-1: none)
    int kk2 = k, jj2 = j, ii2 = i;
    if (o3 != 0) { const int f = s_bc[4 + (o3 > 0)]; if (bc_is_physical(f)) { f3 = f; kk2 = bc_source(q.d3, k, o3, f, false); o3 = 0; } }
    if (o2 != 0) { const int f = s_bc[2 + (o2 > 0)]; if (bc_is_physical(f)) { f2 = f; jj2 = bc_source(q.d2, j, o2, f, false); o2 = 0; } }
    if (o1 != 0) { const int f = s_bc[(o1 > 0)]; if (bc_is_physical(f)) { f1 = f; ii2 = bc_source(q.d1, i, o1, f, false); o1 = 0; } }
    const bool mapped = (f1 >= 0) || (f2 >= 0) || (f3 >= 0);
    const int d = (o3 + 1)*9 + (o2 + 1)*3 + (o1 + 1);
    int src = (d == 13) ? (int)m : s_src[d];
    if (src < 0) {
      if (!mapped) continue;                      // nothing fills this cell (cannot happen with the boundary types admitted)
      src = (int)m; o1 = o2 = o3 = 0;
    }
    const size_t sc = (((size_t)src*q.n3 + (kk2 - o3*q.d3.nx))*q.n2 + (jj2 - o2*q.d2.nx))*q.n1 + (ii2 - o1*q.d1.nx);
    const size_t so = sc + (size_t)src*4*cs, dst = (((size_t)m*5*q.n3 + k)*q.n2 + j)*q.n1 + i;
    double a[5], b[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) { a[n] = u[so + n*cs]; b[n] = w[so + n*cs]; }
    const unsigned fl = flags[sc];
    // reflect: the sign of the normal momentum and of the normal velocity (outflow and copies: identity)
    if (f1 == AKMI_BC_REFLECT) { a[1] = -1.0*a[1]; b[1] = -1.0*b[1]; }
    if (f2 == AKMI_BC_REFLECT) { a[2] = -1.0*a[2]; b[2] = -1.0*b[2]; }
    if (f3 == AKMI_BC_REFLECT) { a[3] = -1.0*a[3]; b[3] = -1.0*b[3]; }
#pragma unroll
    for (int n = 0; n < 5; ++n) { u[dst + n*cs] = a[n]; w[dst + n*cs] = b[n]; }
    if (fl) {
      if (fl & 1u) atomicAdd(&counters[0], 1);
      if (fl & 2u) atomicAdd(&counters[1], 1);
      if (fl & 4u) atomicAdd(&counters[2], 1);
    }
  }
}

static GhostSet cc_set(const Geo &g, double *u) {
  GhostSet gs{};
  gs.q[0] = make_comp(g, 0); gs.a[0] = u; gs.comp[0] = 0; gs.ncomp = 1;
  return gs;
}
static GhostSet fc_set(const Geo &g, double *b1, double *b2, double *b3) {
  GhostSet gs{};
  double *b[3] = {b1, b2, b3};
  for (int c = 0; c < 3; ++c) { gs.q[c] = make_comp(g, c + 1); gs.a[c] = b[c]; gs.comp[c] = c + 1; }
  gs.ncomp = 3;
  return gs;
}

// ---- pack: one grid.y per segment, one grid.z per component of the set (cell-centred: one; face field: x1f, x2f, x3f) ---
struct PackSet { const double *a[3]; int comp0, ncomp; };      // comp0: component of a[0] (0 cell-centred, 1 x1f)
__global__ void __launch_bounds__(256)
k_pack(Geo g, PackSet ps, int nv, const int *__restrict__ send_tab, const long long *__restrict__ send_off,
       double *__restrict__ sendbuf) {
  const int s = blockIdx.y;
  const int m = send_tab[2*s], d = send_tab[2*s + 1];
  int o1, o2, o3;
  if (!dir_valid(g, 26 - d, o1, o2, o3)) return;
  const int fc_comp = ps.comp0 + (int)blockIdx.z;
  const Comp q = make_comp(g, fc_comp);
  const double *__restrict__ a = ps.a[blockIdx.z];
  const int c1 = q.d1.cnt(o1), c2 = q.d2.cnt(o2), c3 = q.d3.cnt(o3);
  long long base = send_off[s];
  if (fc_comp > 1) {  // FC segments hold x1f, x2f, x3f parts back to back
    for (int c = 1; c < fc_comp; ++c) base += seg_count(make_comp(g, c), o1, o2, o3);
  }
  // 32-bit index arithmetic (a segment of one MeshBlock is far below 2^31 elements; checked at launch): the first form
  // decoded a 64-bit flat index with three 64-bit divisions per element on 64 workgroups per segment and packed the 25 MB of a
  // 256^3 block's 26 segments in 57 us (0.44 TB/s; `roofline.halo.pack_ms`)
  const unsigned c12 = (unsigned)c2*(unsigned)c1, per = (unsigned)c3*c12, tot = (unsigned)nv*per;
  const size_t vb = (size_t)m*nv;
  for (unsigned t = blockIdx.x*256u + threadIdx.x; t < tot; t += gridDim.x*256u) {
    const unsigned n = t/per, r = t - n*per;
    const unsigned kk = r/c12, r2 = r - kk*c12;
    const unsigned jj = r2/(unsigned)c1, ii = r2 - jj*(unsigned)c1;
    const int i = q.d1.lo(o1) + (int)ii - o1*q.d1.nx;
    const int j = q.d2.lo(o2) + (int)jj - o2*q.d2.nx;
    const int k = q.d3.lo(o3) + (int)kk - o3*q.d3.nx;
    sendbuf[base + t] = a[(((vb + n)*q.n3 + k)*q.n2 + j)*q.n1 + i];
  }
}

// ---- physical BCs ------------------------------------------------------------------------
// HydroBCs (src/bvals/physics/hydro_bcs.cpp:69-...): DIR-normal ghost layers, all transverse
// indices including ghosts; outflow and reflect.
template <int DIR>
__global__ void __launch_bounds__(256)
k_hydro_bc(Geo g, int nv, const int *__restrict__ bcs, const double *__restrict__ u_in,
           double *__restrict__ u) {
  const int t1 = (DIR == 0) ? g.N2 : g.N1;            // fastest transverse extent
  const int t2 = (DIR == 2) ? g.N2 : g.N3;            // slowest transverse extent
  const long long tot = (long long)g.nmb*nv*t2*t1;
  long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= tot) return;
  int a = (int)(t%t1); t /= t1;
  int b = (int)(t%t2); t /= t2;
  int n = (int)(t%nv);
  int m = (int)(t/nv);
  const int bi = bcs[6*m + 2*DIR], bo = bcs[6*m + 2*DIR + 1];
  const int s = (DIR == 0) ? g.is : (DIR == 1 ? g.js : g.ks);
  const int e = (DIR == 0) ? g.ie : (DIR == 1 ? g.je : g.ke);
  auto at = [&](int x) -> double & {
    int i = (DIR == 0) ? x : a;
    int j = (DIR == 0) ? a : (DIR == 1 ? x : b);
    int k = (DIR == 2) ? x : b;
    return u[ix5(nv, g.N3, g.N2, g.N1, m, n, k, j, i)];
  };
  const double sgn = (n == 1 + DIR) ? -1.0 : 1.0;
  const bool normal = (n == 1 + DIR);
  for (int q = 0; q < g.ng; ++q) {
    if (bi == AKMI_BC_REFLECT) at(s - q - 1) = sgn*at(s + q);
    else if (bi == AKMI_BC_OUTFLOW) at(s - q - 1) = at(s);
    else if (bi == AKMI_BC_INFLOW) at(s - q - 1) = u_in[6*n + 2*DIR];
    else if (bi == AKMI_BC_DIODE) at(s - q - 1) = normal ? fmin(0.0, at(s)) : at(s);
    else if (bi == AKMI_BC_VACUUM) at(s - q - 1) = 0.0;
  }
  for (int q = 0; q < g.ng; ++q) {
    if (bo == AKMI_BC_REFLECT) at(e + q + 1) = sgn*at(e - q);
    else if (bo == AKMI_BC_OUTFLOW) at(e + q + 1) = at(e);
    else if (bo == AKMI_BC_INFLOW) at(e + q + 1) = u_in[6*n + 2*DIR + 1];
    else if (bo == AKMI_BC_DIODE) at(e + q + 1) = normal ? fmax(0.0, at(e)) : at(e);
    else if (bo == AKMI_BC_VACUUM) at(e + q + 1) = 0.0;
  }
}

// BFieldBCs (src/bvals/physics/bfield_bcs.cpp:66-...).  Thread per transverse cell; the
// extra "+1" faces of the transverse components are handled by the last thread in that
// direction exactly as in the reference.
template <int DIR>
__global__ void __launch_bounds__(256)
k_bfield_bc(Geo g, const int *__restrict__ bcs, const double *__restrict__ b_in,
            double *__restrict__ b1, double *__restrict__ b2, double *__restrict__ b3) {
  const int t1 = (DIR == 0) ? g.N2 : g.N1;
  const int t2 = (DIR == 2) ? g.N2 : g.N3;
  const long long tot = (long long)g.nmb*t2*t1;
  long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= tot) return;
  int a = (int)(t%t1); t /= t1;
  int b = (int)(t%t2);
  int m = (int)(t/t2);
  const int bi = bcs[6*m + 2*DIR], bo = bcs[6*m + 2*DIR + 1];
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  // component accessors taking (normal index x, transverse a, b) -> (k,j,i)
  auto kji = [&](int x, int aa, int bb, int &k, int &j, int &i) {
    i = (DIR == 0) ? x : aa;
    j = (DIR == 0) ? aa : (DIR == 1 ? x : bb);
    k = (DIR == 2) ? x : bb;
  };
  auto B1 = [&](int x, int aa, int bb) -> double & { int k, j, i; kji(x, aa, bb, k, j, i); return b1[ix4(N3, N2, N1 + 1, m, k, j, i)]; };
  auto B2 = [&](int x, int aa, int bb) -> double & { int k, j, i; kji(x, aa, bb, k, j, i); return b2[ix4(N3, N2 + 1, N1, m, k, j, i)]; };
  auto B3 = [&](int x, int aa, int bb) -> double & { int k, j, i; kji(x, aa, bb, k, j, i); return b3[ix4(N3 + 1, N2, N1, m, k, j, i)]; };
  const int s = (DIR == 0) ? g.is : (DIR == 1 ? g.js : g.ks);
  const int e = (DIR == 0) ? g.ie : (DIR == 1 ? g.je : g.ke);
  // which transverse coordinate carries the "+1" face of each transverse component:
  //  DIR=0: a=j (x2f extra at j==N2-1), b=k (x3f extra at k==N3-1)
  //  DIR=1: a=i (x1f extra at i==N1-1), b=k (x3f extra at k==N3-1)
  //  DIR=2: a=i (x1f extra at i==N1-1), b=j (x2f extra at j==N2-1)
  for (int q = 0; q < g.ng; ++q) {
    for (int side = 0; side < 2; ++side) {
      const int bc = side ? bo : bi;
      // diode and vacuum treat the field like outflow (bfield_bcs.cpp:88-97); inflow: b_in
      if (bc != AKMI_BC_REFLECT && bc != AKMI_BC_OUTFLOW && bc != AKMI_BC_DIODE &&
          bc != AKMI_BC_VACUUM && bc != AKMI_BC_INFLOW) continue;
      const bool refl = (bc == AKMI_BC_REFLECT);
      const bool infl = (bc == AKMI_BC_INFLOW);
      const int face = 2*DIR + side;
      // ghost index / source index for the normal component (face-centred along DIR)
      const int gn = side ? e + q + 2 : s - q - 1;
      const int sn = side ? (refl ? e - q : e + 1) : (refl ? s + q + 1 : s);
      // for transverse components (cell-centred along DIR)
      const int gt = side ? e + q + 1 : s - q - 1;
      const int stt = side ? (refl ? e - q : e) : (refl ? s + q : s);
      const double sg = refl ? -1.0 : 1.0;
      // value stored in a ghost face of component c: the inflow constant or the (reflected) source
#define BV(c, expr) (infl ? b_in[6*(c) + face] : (expr))
      if (DIR == 0) {
        B1(gn, a, b) = BV(0, sg*B1(sn, a, b));
        B2(gt, a, b) = BV(1, B2(stt, a, b));
        if (a == N2 - 1) B2(gt, a + 1, b) = BV(1, B2(stt, a + 1, b));
        B3(gt, a, b) = BV(2, B3(stt, a, b));
        if (b == N3 - 1) B3(gt, a, b + 1) = BV(2, B3(stt, a, b + 1));
      } else if (DIR == 1) {
        B1(gt, a, b) = BV(0, B1(stt, a, b));
        if (a == N1 - 1) B1(gt, a + 1, b) = BV(0, B1(stt, a + 1, b));
        B2(gn, a, b) = BV(1, sg*B2(sn, a, b));
        B3(gt, a, b) = BV(2, B3(stt, a, b));
        if (b == N3 - 1) B3(gt, a, b + 1) = BV(2, B3(stt, a, b + 1));
      } else {
        B1(gt, a, b) = BV(0, B1(stt, a, b));
        if (a == N1 - 1) B1(gt, a + 1, b) = BV(0, B1(stt, a + 1, b));
        B2(gt, a, b) = BV(1, B2(stt, a, b));
        if (b == N2 - 1) B2(gt, a, b + 1) = BV(1, B2(stt, a, b + 1));
        B3(gn, a, b) = BV(2, sg*B3(sn, a, b));
      }
#undef BV
    }
  }
}

}  // namespace akmi

using namespace akmi;

extern "C" {

long long akmi_bvals_cc_segsize(const akmi_pack *p, int d) {
  Geo g = make_geo(p);
  int o1, o2, o3;
  if (!dir_valid(g, d, o1, o2, o3)) return 0;
  return seg_count(make_comp(g, 0), o1, o2, o3);
}

long long akmi_bvals_fc_segsize(const akmi_pack *p, int d) {
  Geo g = make_geo(p);
  int o1, o2, o3;
  if (!dir_valid(g, d, o1, o2, o3)) return 0;
  long long t = 0;
  for (int c = 1; c <= 3; ++c) t += seg_count(make_comp(g, c), o1, o2, o3);
  return t;
}

int akmi_bvals_cc_local(const akmi_pack *p, int nvar, const int *nghbr, double *u, void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<0>(g, cc_set(g, u), nvar, nghbr, nullptr, nullptr, (hipStream_t)stream);
}

int akmi_bvals_cc_local_bcs(const akmi_pack *p, int nvar, const int *nghbr, const int *bcs, const double *u_in, double *u,
                            double *dt3_reset, void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<0, true>(g, cc_set(g, u), nvar, nghbr, nullptr, nullptr, (hipStream_t)stream,
                               GhostBC{bcs, u_in, dt3_reset});
}

// Hydro, after akmi_hydro_stage_w: ghost zones of the conserved AND of the primitive variables in one launch.
// The reference fills the ghost zones of u0 (neighbour copies, boundary functions) and then converts EVERY cell, so a ghost
// cell's (u, w) is the conversion of the same numbers its source cell was converted from -- including the floors: the result
// is the source cell's (u, w) under the value rule of the boundary (identity for a neighbour / periodic / outflow copy, the
// sign of the normal momentum AND of the normal velocity for reflect; e_kin is even in both).  Converting the ghost copy of
// an already floored u again is NOT that: (efloor + e_kin) - e_kin need not give efloor back.  So both arrays are gathered
// with the index maps and value rules of k_ghost_fill<0, true>, and nothing is converted here.  Boundary types whose value
// rule does not commute with the conversion (diode, vacuum, inflow, user) are refused: the caller keeps the separate
// ConsToPrim pass for such packs.  The floor counters of the reference count every cell its ConsToPrim converts, ghost cells
// included: akmi_hydro_stage_w leaves one flag byte per active cell at the start of its workspace, and every ghost image of
// a flagged cell is counted here.
int akmi_hydro_ghost_uw(const akmi_pack *p, const int *nghbr, const int *bcs, const int *bcs_host, double *u, double *w,
                        const void *ws, int *counters, void *stream) {
  Geo g = make_geo(p);
  if (!bcs || !bcs_host || !ws || !counters) { set_error("hydro_ghost_uw: bcs (device and host), ws and counters are required"); return AKMI_FAIL; }
  if (bcs_host)
    for (int q = 0; q < 6*p->nmb; ++q) {
      const int f = bcs_host[q];
      if (f != AKMI_BC_BLOCK && f != AKMI_BC_PERIODIC && f != AKMI_BC_OUTFLOW && f != AKMI_BC_REFLECT) {
        set_error("hydro_ghost_uw: boundary type %d does not commute with ConsToPrim (periodic, outflow, reflect only)", f);
        return AKMI_FAIL;
      }
    }
  if (!p->is_ideal || p->nvar != 5) { set_error("hydro_ghost_uw: ideal gas without passive scalars"); return AKMI_FAIL; }
  const Comp q = make_comp(g, 0);
  const long long n0 = (long long)q.n1*q.n2*2*q.d3.ng, n1 = (long long)q.n1*2*q.d2.ng*(q.d3.eo - q.d3.s + 1),
                  n2 = (long long)2*q.d1.ng*(q.d2.eo - q.d2.s + 1)*(q.d3.eo - q.d3.s + 1);
  const long long nmax = n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2);
  if (nmax >= (1ll << 31)) { set_error("hydro_ghost_uw: a ghost slab has 2^31 cells or more"); return AKMI_FAIL; }
  long long chunks = (nmax + 255)/256;
  const long long cap = ((1ll << 31) - 1)/g.nmb;
  if (chunks > cap) chunks = cap;
  if (chunks > 64 && (long long)g.nmb*chunks > (1ll << 20)) { chunks = (1ll << 20)/g.nmb; if (chunks < 64) chunks = 64; if (chunks > cap) chunks = cap; }
  if (chunks < 1) { set_error("hydro_ghost_uw: too many MeshBlocks for one launch"); return AKMI_FAIL; }
  dim3 grid((unsigned)(g.nmb*chunks), 1, 3);
  k_ghost_fill_uw<<<grid, 256, 0, (hipStream_t)stream>>>(g, q, (unsigned)chunks, nghbr, bcs, u, w,
                                                        static_cast<const unsigned char *>(ws), counters);
  AKMI_CHECK_LAUNCH("hydro_ghost_uw");
  return AKMI_COMPLETE;
}

int akmi_bvals_fc_local_bcs(const akmi_pack *p, const int *nghbr, const int *bcs, const double *b_in, double *bx1f,
                            double *bx2f, double *bx3f, void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<0, true>(g, fc_set(g, bx1f, bx2f, bx3f), 1, nghbr, nullptr, nullptr, (hipStream_t)stream,
                               GhostBC{bcs, b_in, nullptr});
}

int akmi_bvals_cc_unpack(const akmi_pack *p, int nvar, const int *nghbr, const long long *seg_off,
                         const double *recvbuf, double *u, void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<1>(g, cc_set(g, u), nvar, nghbr, seg_off, recvbuf, (hipStream_t)stream);
}

// workgroups per segment: the largest segment of the pack (a face slab of nv variables) in pieces of 1024 elements, at most 512
// (a thread then strides); small segments leave most of them idle at once
static unsigned pack_chunks(const Geo &g, int nv) {
  const long long f1 = (long long)(g.N2 + 1)*(g.N3 + 1)*g.ng, f2 = (long long)(g.N1 + 1)*(g.N3 + 1)*g.ng,
                  f3 = (long long)(g.N1 + 1)*(g.N2 + 1)*g.ng;
  long long mx = f1 > f2 ? (f1 > f3 ? f1 : f3) : (f2 > f3 ? f2 : f3);
  long long ch = (mx*nv + 1023)/1024;
  return (unsigned)(ch < 1 ? 1 : (ch > 512 ? 512 : ch));
}

int akmi_bvals_cc_pack(const akmi_pack *p, int nvar, int nsend, const int *send_tab,
                       const long long *send_off, const double *u, double *sendbuf, void *stream) {
  if (nsend <= 0) return AKMI_COMPLETE;
  Geo g = make_geo(p);
  dim3 grid(pack_chunks(g, nvar), nsend, 1);
  k_pack<<<grid, 256, 0, (hipStream_t)stream>>>(g, PackSet{{u, nullptr, nullptr}, 0, 1}, nvar, send_tab, send_off, sendbuf);
  AKMI_CHECK_LAUNCH("cc_pack");
  return AKMI_COMPLETE;
}

int akmi_bvals_fc_local(const akmi_pack *p, const int *nghbr, double *bx1f, double *bx2f,
                        double *bx3f, void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<0>(g, fc_set(g, bx1f, bx2f, bx3f), 1, nghbr, nullptr, nullptr, (hipStream_t)stream);
}

int akmi_bvals_fc_pack(const akmi_pack *p, int nsend, const int *send_tab, const long long *send_off,
                       const double *bx1f, const double *bx2f, const double *bx3f, double *sendbuf,
                       void *stream) {
  if (nsend <= 0) return AKMI_COMPLETE;
  Geo g = make_geo(p);
  dim3 grid(pack_chunks(g, 1), nsend, 3);            // the three face components in one launch
  k_pack<<<grid, 256, 0, (hipStream_t)stream>>>(g, PackSet{{bx1f, bx2f, bx3f}, 1, 3}, 1, send_tab, send_off, sendbuf);
  AKMI_CHECK_LAUNCH("fc_pack");
  return AKMI_COMPLETE;
}

}  // extern "C"

extern "C" {

int akmi_bvals_fc_unpack(const akmi_pack *p, const int *nghbr, const long long *seg_off,
                         const double *recvbuf, double *bx1f, double *bx2f, double *bx3f,
                         void *stream) {
  Geo g = make_geo(p);
  return launch_ghost<2>(g, fc_set(g, bx1f, bx2f, bx3f), 1, nghbr, seg_off, recvbuf,
                         (hipStream_t)stream);
}

// dirs: bit d set = some MeshBlock of the pack has a physical boundary across direction d (the caller knows its flags;
// a direction without one would launch a kernel whose every thread returns at once: 4-5 us each, 13 % of a stage of
// the 128^3 Sod deck when both x2 and x3 are periodic).  7 = all directions.
static int hydro_bcs(const akmi_pack *p, int nvar, const int *bcs, const double *u_in, double *u,
                     void *stream, int dirs = 7) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  if (dirs & 1) {
    long long n = (long long)g.nmb*nvar*g.N3*g.N2;
    k_hydro_bc<0><<<(int)((n + 255)/256), 256, 0, st>>>(g, nvar, bcs, u_in, u);
  }
  if (g.multi_d && (dirs & 2)) {
    long long n = (long long)g.nmb*nvar*g.N3*g.N1;
    k_hydro_bc<1><<<(int)((n + 255)/256), 256, 0, st>>>(g, nvar, bcs, u_in, u);
  }
  if (g.three_d && (dirs & 4)) {
    long long n = (long long)g.nmb*nvar*g.N2*g.N1;
    k_hydro_bc<2><<<(int)((n + 255)/256), 256, 0, st>>>(g, nvar, bcs, u_in, u);
  }
  AKMI_CHECK_LAUNCH("hydro_bcs");
  return AKMI_COMPLETE;
}

int akmi_hydro_bcs(const akmi_pack *p, int nvar, const int *bcs, double *u, void *stream) {
  return hydro_bcs(p, nvar, bcs, nullptr, u, stream);
}
int akmi_hydro_bcs_inflow(const akmi_pack *p, int nvar, const int *bcs, const double *u_in, double *u,
                          void *stream) {
  return hydro_bcs(p, nvar, bcs, u_in, u, stream);
}
int akmi_hydro_bcs_dirs(const akmi_pack *p, int nvar, const int *bcs, int dirs, const double *u_in, double *u,
                        void *stream) {
  return hydro_bcs(p, nvar, bcs, u_in, u, stream, dirs);
}

static int bfield_bcs(const akmi_pack *p, const int *bcs, const double *b_in, double *bx1f,
                      double *bx2f, double *bx3f, void *stream, int dirs = 7) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  if (dirs & 1) {
    long long n = (long long)g.nmb*g.N3*g.N2;
    k_bfield_bc<0><<<(int)((n + 255)/256), 256, 0, st>>>(g, bcs, b_in, bx1f, bx2f, bx3f);
  }
  if (g.multi_d && (dirs & 2)) {
    long long n = (long long)g.nmb*g.N3*g.N1;
    k_bfield_bc<1><<<(int)((n + 255)/256), 256, 0, st>>>(g, bcs, b_in, bx1f, bx2f, bx3f);
  }
  if (g.three_d && (dirs & 4)) {
    long long n = (long long)g.nmb*g.N2*g.N1;
    k_bfield_bc<2><<<(int)((n + 255)/256), 256, 0, st>>>(g, bcs, b_in, bx1f, bx2f, bx3f);
  }
  AKMI_CHECK_LAUNCH("bfield_bcs");
  return AKMI_COMPLETE;
}

int akmi_bfield_bcs(const akmi_pack *p, const int *bcs, double *bx1f, double *bx2f, double *bx3f,
                    void *stream) {
  return bfield_bcs(p, bcs, nullptr, bx1f, bx2f, bx3f, stream);
}
int akmi_bfield_bcs_inflow(const akmi_pack *p, const int *bcs, const double *b_in, double *bx1f,
                           double *bx2f, double *bx3f, void *stream) {
  return bfield_bcs(p, bcs, b_in, bx1f, bx2f, bx3f, stream);
}
int akmi_bfield_bcs_dirs(const akmi_pack *p, const int *bcs, int dirs, const double *b_in, double *bx1f, double *bx2f,
                         double *bx3f, void *stream) {
  return bfield_bcs(p, bcs, b_in, bx1f, bx2f, bx3f, stream, dirs);
}

int akmi_calib_copy(double *dst, const double *src, long long n, void *stream);

}  // extern "C"

namespace akmi {
__global__ void __launch_bounds__(256) k_calib_copy(double *__restrict__ dst,
                                                    const double *__restrict__ src, long long n) {
  for (long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x; t < n;
       t += (long long)gridDim.x*blockDim.x)
    dst[t] = src[t];
}
}  // namespace akmi

extern "C" int akmi_calib_copy(double *dst, const double *src, long long n, void *stream) {
  long long nb = (n + 255)/256;
  if (nb > 16384) nb = 16384;
  akmi::k_calib_copy<<<(int)nb, 256, 0, (hipStream_t)stream>>>(dst, src, n);
  AKMI_CHECK_LAUNCH("calib_copy");
  return AKMI_COMPLETE;
}
